#!/usr/bin/env python
"""Instruction histogram of a kernel's hot loop straight from the code object (no GPU needed):

    python tools/isa_histogram.py pointnav-vo_amd/csrc/stem_rs.o "stem_rs_kernel<2, true, false, true>"

Unbundles the gfx950 code object of the .o (llvm-objdump --offloading), disassembles it, finds the kernel whose demangled name
contains the given text and prints, for every backward branch that spans more than --min-span instructions (the tile loops), the
instruction classes inside the loop.  What the numbers are for: with one wave per SIMD about 3.6 issue slots hide behind a 32x32x16
MFMA (profiles/r4_stem_rs_ablations.txt); every further instruction of the loop costs its ~4 cycles."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def cls(op):
    for pre, name in (("v_mfma", "MFMA"), ("s_waitcnt", "s_waitcnt"), ("s_nop", "s_nop"), ("s_barrier", "s_barrier"),
                      ("s_cbranch", "branch"), ("s_branch", "branch"), ("s_", "SALU"), ("ds_read", "LDS read"), ("ds_write", "LDS write"),
                      ("ds_", "LDS other"), ("buffer_load", "VMEM load"), ("global_load", "VMEM load"), ("buffer_store", "VMEM store"),
                      ("global_store", "VMEM store"), ("global_atomic", "VMEM atomic"), ("buffer_atomic", "VMEM atomic"),
                      ("scratch", "scratch"), ("v_accvgpr", "v_accvgpr"), ("v_cvt", "v_cvt"), ("v_readlane", "lane ops (SGPR spills)"),
                      ("v_writelane", "lane ops (SGPR spills)"), ("v_readfirstlane", "lane ops (SGPR spills)"), ("v_", "VALU other")):
        if op.startswith(pre):
            return name
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("obj")
    ap.add_argument("kernel")
    ap.add_argument("--min-span", type=int, default=400)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as t:
        base = os.path.basename(a.obj)
        subprocess.check_call(["cp", a.obj, t])
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", base], cwd=t, capture_output=True)
        co = [f for f in os.listdir(t) if "gfx950" in f][0]
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], cwd=t, capture_output=True, text=True).stdout
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
        m = re.match(r"[0-9a-f]{16} <(.*)>:", f)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if a.kernel not in name:
            continue
        ins = []
        for ln in f.split("\n")[1:]:
            code = ln.split("//")[0].strip()
            ad = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            if code and ad:
                ins.append((code.split()[0], code, int(ad.group(1), 16)))
        a2i = {x[2]: i for i, x in enumerate(ins)}
        print(f"## {name.split('(')[0]}: {len(ins)} instructions, {sum(1 for x in ins if x[0].startswith('v_mfma'))} MFMAs in the kernel")
        for i, (op, code, ad) in enumerate(ins):
            if op not in ("s_branch", "s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccnz", "s_cbranch_vccz", "s_cbranch_execnz"):
                continue
            off = int(code.split()[-1])
            off = off - 65536 if off > 32767 else off
            tgt = a2i.get(ad + 4 + 4 * off)
            if tgt is None or i - tgt < a.min_span:
                continue
            c = collections.Counter(cls(x[0]) for x in ins[tgt:i + 1])
            n = sum(c.values())
            print(f"\nloop of {n} instructions (#{tgt}..#{i}), {c['MFMA']} MFMAs -> {n - c['MFMA']} others = "
                  f"{(n - c['MFMA']) / max(c['MFMA'], 1):.2f} per MFMA")
            print("| class | count |\n|---|---|")
            for k, v in c.most_common():
                print(f"| {k} | {v} |")
            for kind in ("SALU", "VALU other"):
                cs = collections.Counter(x[0] for x in ins[tgt:i + 1] if cls(x[0]) == kind)
                print(f"{kind}: " + ", ".join(f"{k} {v}" for k, v in cs.most_common(10)))
        return 0
    print("kernel not found", file=sys.stderr)
    return 1


if __name__ == "__main__":
    sys.exit(main())
