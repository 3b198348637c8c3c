#!/usr/bin/env python3
"""Static instruction counts of a kernel between `; MARK x` comments of its assembly (hipcc -S of a source with
asm volatile("; MARK x") lines): VALU / SALU / LDS / VMEM / waitcnt per section, in program order.  A planning tool for
instruction-count work on the block coders (the compiler may move instructions across a marker; counts are approximate).
Usage: isa_sections.py file.s kernel-name-substring"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    lines = open(sys.argv[1]).read().splitlines()
    want = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if want in l and not l.startswith(("\t", ".")) and l.split(";")[0].strip().endswith(":"))
    sect, order, counts = "(before)", ["(before)"], {"(before)": {}}
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".section") or t.startswith(".rodata") or ".amdhsa_kernel" in t:
            break
        m = re.match(r"; MARK (\S+)", t)
        if m:
            sect = m.group(1)
            if sect not in counts:
                counts[sect] = {}; order.append(sect)
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c = classify(op)
        counts[sect][c] = counts[sect].get(c, 0) + 1
    cols = ["valu", "salu", "lds", "vmem", "wait", "branch", "other"]
    print("%-16s" % "section" + "".join("%8s" % c for c in cols))
    for s in order:
        print("%-16s" % s + "".join("%8d" % counts[s].get(c, 0) for c in cols))
    print("%-16s" % "total" + "".join("%8d" % sum(counts[s].get(c, 0) for s in order) for c in cols))


if __name__ == "__main__":
    main()
