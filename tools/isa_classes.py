#!/usr/bin/env python3
"""Static histogram of one kernel's ISA by source-line range and VALU issue class.

Input: the `.s` of a `hipcc --save-temps -gline-tables-only` build (tools/isa_dump.sh) and the kernel's mangled name.
Every instruction is attributed to the last `.loc` in front of it (file 'tamp_compress_kernel.hpp' only; inlined
callees keep their own lines) and classed by what profiles/r5_valu_issue_probe.txt measured on gfx950:

  fast   2.2 cycles per wave64 instruction when other wavefronts share the SIMD: v_and / v_or / v_xor / v_add_u32 /
         v_sub(rev)_u32 / v_lshrrev_b32 / v_mov_b32 / v_not in the VOP1 / VOP2 encodings with VGPR, inline-constant
         or literal operands (no SGPR operand, no carry, no DPP / SDWA)
  slow   4.1 cycles: everything else on the vector ALU (shifts left, bfe, perm, alignbyte, compares, cndmask, min / max,
         multiplies, three-operand forms, anything that reads an SGPR, v_readlane / v_writelane / v_readfirstlane)
  spill  v_writelane / v_readlane whose VGPR is one of the kernel's SGPR-spill registers (listed by the caller or
         detected: a v_writelane into a register that is later only v_readlane'd)

Output: one row per requested line range: instructions by class (static), and the loop heads found.
"""
import re
import sys
from collections import Counter, defaultdict

FAST_OPS = {
    "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_mov_b32",
    "v_not_b32", "v_fma_f32", "v_add_f32", "v_mul_f32",
}


def classify(op, operands):
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        return "other"
    if op in ("v_readlane_b32", "v_writelane_b32"):
        return "lane"
    base = op
    for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
        if base.endswith(suf):
            base = base[: -len(suf)]
    if op.endswith(("_dpp", "_sdwa")):
        return "slow"
    if base in FAST_OPS:
        # an SGPR / VCC / EXEC / M0 source makes it the slow form
        srcs = operands.split(",")[1:]
        for s_ in srcs:
            s_ = s_.strip()
            if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc)$", s_):
                return "slow"
        return "fast"
    return "slow"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    ranges = []
    for a in sys.argv[3:]:
        name, lo, hi = a.split(":")
        ranges.append((name, int(lo), int(hi)))
    lines = open(path).read().split("\n")
    # file number of tamp_compress_kernel.hpp
    fileno = None
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"tamp_compress_kernel\.hpp"', ln)
        if m:
            fileno = int(m.group(1))
    start = next(i for i, ln in enumerate(lines) if ln.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel " + kernel in lines[i])
    cur = None
    per_line = defaultdict(Counter)
    total = Counter()
    for ln in lines[start:end]:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = int(m.group(2)) if int(m.group(1)) == fileno else -int(m.group(1))
            continue
        if not s or s.startswith((".", ";", "_Z")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        operands = parts[1].split(";")[0] if len(parts) > 1 else ""
        c = classify(op, operands)
        per_line[cur][c] += 1
        total[c] += 1
    classes = ["fast", "slow", "lane", "salu", "lds", "vmem", "other"]
    print("# %s" % kernel)
    print("# static instruction counts; 'fast' / 'slow' = VALU issue classes (2.2 / 4.1 cycles), 'lane' = v_readlane / v_writelane")
    print("%-28s %s" % ("range", " ".join("%7s" % c for c in classes)))
    print("%-28s %s" % ("whole kernel", " ".join("%7d" % total[c] for c in classes)))
    for name, lo, hi in ranges:
        t = Counter()
        for l_, cnt in per_line.items():
            if l_ is not None and lo <= l_ <= hi:
                t.update(cnt)
        print("%-28s %s" % ("%s [%d-%d]" % (name, lo, hi), " ".join("%7d" % t[c] for c in classes)))


if __name__ == "__main__":
    main()
