#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a compact per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db [--pmc] > profiles/<name>.md

Kernels are grouped by (short name, grid) so that one template instantiation used by several layers (e.g. the stem
and the layer-1 convs both run conv_mfma_kernel<4,1,false>) is reported per launch shape; durations are in us.
With --pmc the per-dispatch counter values are averaged per group as well.
"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("pnvo::", "")
    return name if len(name) < 70 else name[:67] + "..."


def main():
    path = sys.argv[1]
    pmc = "--pmc" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, duration, vgpr_count, accum_vgpr_count, lds_size, "
                       "dispatch_id from kernels").fetchall()
    groups = {}
    for name, gx, gy, wx, dur, vg, ag, lds, did in rows:
        k = (short(name), gx // max(wx, 1), gy)
        g = groups.setdefault(k, dict(n=0, tot=0.0, mn=1e30, mx=0.0, vgpr=vg, agpr=ag, lds=lds, ids=[]))
        g["n"] += 1
        g["tot"] += dur / 1e3
        g["mn"] = min(g["mn"], dur / 1e3)
        g["mx"] = max(g["mx"], dur / 1e3)
        g["ids"].append(did)
    total = sum(g["tot"] for g in groups.values())
    print(f"# rocprofv3 kernel-trace summary: {path}\n")
    print(f"total kernel time {total / 1e3:.3f} ms over {sum(g['n'] for g in groups.values())} dispatches\n")
    print("| kernel | grid (WGs x,y) | calls | avg us | min us | max us | total ms | % | VGPR | AGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["tot"]):
        if g["tot"] / total < 0.002:
            continue
        print(f"| {k[0]} | {k[1]},{k[2]} | {g['n']} | {g['tot'] / g['n']:.1f} | {g['mn']:.1f} | {g['mx']:.1f} | "
              f"{g['tot'] / 1e3:.3f} | {100 * g['tot'] / total:.1f} | {g['vgpr']} | {g['agpr']} | {g['lds']} |")
    if pmc:
        try:
            ev = cur.execute("select dispatch_id, counter_name, value from counters_collection").fetchall()
        except sqlite3.Error as e:
            print("\n(no counters:", e, ")")
            return
        by = {}
        for did, cname, val in ev:
            by.setdefault(did, {}).setdefault(cname, 0.0)
            by[did][cname] += val
        names = sorted({c for d in by.values() for c in d})
        print("\n## counters (average per dispatch)\n")
        print("| kernel | grid | " + " | ".join(names) + " |")
        print("|---|---|" + "---|" * len(names))
        for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["tot"]):
            vals = [by[i] for i in g["ids"] if i in by]
            if not vals or g["tot"] / total < 0.002:
                continue
            print(f"| {k[0]} | {k[1]},{k[2]} | " + " | ".join(
                f"{sum(v.get(n, 0.0) for v in vals) / len(vals):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main()
