"""MFMA-pipe utilisation per kernel from rocprofv3 counters (the method of profiles/r1g_mfma_util.md).

  python tools/mfma_util.py <kernel_trace.md> <pmc.md>  > profiles/<tag>_mfma_util.md

clock        = SQ_BUSY_CYCLES / 32 shader engines / average duration (kernel trace)
utilisation  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 x 1024 SIMDs)
Both files are the markdown tables tools/rocprof_summary.py writes (tools/round_profile.sh)."""
import sys


def rows(path, need):
    out, cols = {}, None
    for ln in open(path):
        if not ln.startswith("|"):
            continue
        cells = [c.strip() for c in ln.strip().strip("|").split("|")]
        if cells[0] == "kernel":
            cols = cells if all(n in cells for n in need) else None
            continue
        if cols is None or set(cells[0]) <= {"-"}:
            continue
        out.setdefault((cells[0], cells[1]), dict(zip(cols, cells)))
    return out


def main():
    trace = rows(sys.argv[1], ["avg us"])
    pmc = rows(sys.argv[2], ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"])
    print("# MFMA-pipe utilisation per kernel (counters: %s, durations: %s)\n" % (sys.argv[2], sys.argv[1]))
    print("clock = SQ_BUSY_CYCLES / 32 shader engines / average duration; utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)\n")
    print("| kernel | grid | avg us | clock GHz | MFMA utilisation |\n|---|---|---|---|---|")
    for key, t in sorted(trace.items(), key=lambda kv: -float(kv[1]["avg us"]) * float(kv[1]["calls"])):
        c = pmc.get(key)
        if c is None or float(c["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
            continue
        cyc = float(c["SQ_BUSY_CYCLES"]) / 32.0
        us = float(t["avg us"])
        print("| %s | %s | %.1f | %.2f | %.2f |" % (key[0], key[1], us, cyc / us / 1e3, float(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * 1024.0)))


if __name__ == "__main__":
    main()
