#!/usr/bin/env python3
"""Scan a `hipcc -S` listing for inline-asm MFMAs whose A / B / C source registers are written by a VECTOR-ALU instruction in the
few issue slots before them -- hipcc's hazard recogniser does not look inside inline asm, so such a pair gets no wait states
(stfront.hip, ffn.hip pin their operands by hand; this is the check that they did).  usage: mfma_hazard_check.py file.s [slots=4]"""
import re
import sys


def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def main():
    path, slots = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
    kern, hist, bad = None, [], 0
    for ln, line in enumerate(open(path), 1):
        t = line.strip()
        if t.endswith(":") and t.startswith("_Z"):
            kern, hist = t[:-1], []
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        args = [a.strip() for a in rest.split(",")]
        if op.startswith("v_mfma"):
            srcs = set().union(*(regs(a) for a in args[1:4]))
            for d, (pln, pop, pdst) in enumerate(reversed(hist[-slots:]), 1):
                if pdst & srcs:
                    bad += 1
                    print(f"{path}:{ln}: {op} reads {sorted(pdst & srcs)[:4]}.. written {d} slot(s) earlier by `{pop}` (line {pln}) in {kern[:60]}")
            hist.append((ln, op, set()))
            continue
        if op.startswith("s_nop"):
            n = int(args[0]) + 1 if args and args[0].isdigit() else 1
            hist += [(ln, "s_nop", set())] * n
            continue
        dst = regs(args[0]) if (op.startswith("v_") and not op.startswith("v_cmp")) else set()
        hist.append((ln, op, dst))
    print(f"{bad} suspicious MFMA operand write(s) within {slots} slots")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
