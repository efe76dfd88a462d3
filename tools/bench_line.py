"""Headline fields and the slowest kernels of a bench.py JSON line: python tools/bench_line.py <file with the line>."""
import json,sys
r=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value",r["value"],"ms",r["ms_per_step"],"p50",r.get("ms_per_step_p50"),"frac",r["roofline"]["frac"],"traffic",r["roofline"].get("traffic"))
ks=r.get("kernels") or r.get("per_kernel") or []
for k in ks[:14]: print("  %-60s %.4f"%(k["name"],k["ms_per_step"]))
