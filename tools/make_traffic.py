#!/usr/bin/env python
"""profiles/traffic.json from the rocprofv3 PMC passes of bench.py (run by tools/round_profile.sh on the GPU box).

    python tools/make_traffic.py <config> <batch> <fetch_db> <write_db> [<out.json> [<kernel_trace.md>]]

HBM bytes per launch of the dominant kernel = FETCH_SIZE [KiB] x 1024 x 2 + WRITE_SIZE [KiB] x 1024: on gfx950 FETCH_SIZE
tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, section HBM), so a wide streaming read is doubled before it is compared with
a byte count; WRITE_SIZE is taken as is.  The record carries the hash of the kernel sources it was measured on: bench.py
reports it as stale (traffic = null) when the sources have changed since."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# rocprof kernel-name prefix -> the name bench.py gives that launch
KERNELS = {"fwd_fp32": ("stem_rs_kernel<2, true, false", "conv:visual_encoder.backbone.conv1.0"),
           "dual_bf16": ("stem_rs_kernel<1, false, false", "bf16:stem")}


def avg_counter(db, prefix, counter):
    cur = sqlite3.connect(db).cursor()
    ids = [r[0] for r in cur.execute("select dispatch_id, name from kernels").fetchall()
           if prefix in r[1].replace("void ", "").replace("pnvo::", "")]
    if not ids:
        return None
    vals = {}
    for did, cname, val in cur.execute("select dispatch_id, counter_name, value from counters_collection").fetchall():
        if cname == counter and did in set(ids):
            vals[did] = vals.get(did, 0.0) + val
    return sum(vals.values()) / max(len(vals), 1) if vals else None


def trace_row(md, prefix):
    """(calls, avg us, min us) of the kernel in a tools/rocprof_summary.py table of the plain --kernel-trace --stats pass (the PMC
    passes serialise the launches and stretch them: durations are never taken from those)."""
    try:
        for ln in open(md):
            c = [x.strip() for x in ln.split("|")]
            if len(c) > 6 and prefix in c[1]:
                return int(c[3]), float(c[4]), float(c[5])
    except (OSError, ValueError):
        pass
    return None


def main():
    import bench
    config, batch, fdb, wdb = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "traffic.json")
    row = trace_row(sys.argv[6], KERNELS[config][0]) if len(sys.argv) > 6 else None
    prefix, name = KERNELS[config]
    fetch, write = avg_counter(fdb, prefix, "FETCH_SIZE"), avg_counter(wdb, prefix, "WRITE_SIZE")
    if fetch is None or write is None:
        print("make_traffic: kernel or counters not found", prefix, fetch, write, file=sys.stderr)
        return 1
    try:
        rec = json.load(open(out))
    except (OSError, ValueError):
        rec = {}
    rec.setdefault(config, {})[name] = {
        "batch": batch, "bytes_per_launch": fetch * 1024 * 2 + write * 1024, "fetch_size_kib": fetch, "write_size_kib": write,
        "source_hash": bench.source_hash(), "rocprof_kernel": prefix,
        "file": f"profiles/ (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc, {os.path.basename(os.path.dirname(fdb))})"}
    if row is not None:            # the same command's rocprofv3 --kernel-trace --stats pass: launches, average / minimum duration
        rec[config][name].update({"rocprof_calls": row[0], "rocprof_avg_us": row[1], "rocprof_min_us": row[2],
                                  "rocprof_file": os.path.basename(sys.argv[6])})
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print("make_traffic:", config, name, rec[config][name])
    return 0


if __name__ == "__main__":
    sys.exit(main())
