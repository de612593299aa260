#!/usr/bin/env python3
"""Instruction census of the hot loop of one kernel in a `hipcc -S` listing, block by block in program order, priced with the
issue costs of MI355X_MICROARCH.md ("per-instruction cycle constants"): what the one-wave issue bound of a loop iteration is.
usage: loop_census.py listing.s <kernel-symbol-substring> [min_exp_per_loop]"""
import collections
import re
import sys

COST = {"mfma16": 16, "mfma32": 32, "trans": 8, "valu": 4, "pk32": 8, "lds": 4, "salu": 1, "vmem": 4, "wait": 0}
MFMA_HOLD = 8      # vector-issue cycles an MFMA holds (guide: 8 of its 16 / 32)


def cat(op):
    if op.startswith("v_mfma"):
        return "mfma32" if "32x32" in op else "mfma16"
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_rsq_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32"):
        return "trans"
    if op.startswith("v_pk_") and op.endswith("f32"):
        return "pk32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "valu"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(("Params:", ")")) or (l.startswith("_Z") and sym in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    # the hot loop: the backward branch whose body holds the most transcendentals / MFMAs
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            score = sum(("v_exp" in x) or ("v_mfma" in x) for x in body[a:i])
            if best is None or score > best[0]:
                best = (score, a, i)
    _, a, b = best
    print(f"kernel {body[0][:100]}")
    print(f"hot loop: listing lines {start + a + 1}..{start + b + 1}")
    tot = collections.Counter()
    cur, blk = "(loop head)", collections.Counter()
    order = []
    for l in body[a:b + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            order.append((cur, blk)); cur, blk = m.group(1), collections.Counter()
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        blk[t.split()[0]] += 1
    order.append((cur, blk))
    for name, blk in order:
        if not blk:
            continue
        cc = collections.Counter()
        for op, n in blk.items():
            cc[cat(op)] += n
        print(f"  block {name:14s} {sum(blk.values()):4d} instr  " + "  ".join(f"{k} {v}" for k, v in sorted(cc.items())))
        detail = {op: n for op, n in blk.items() if cat(op) in ("valu", "trans", "pk32")}
        print("      vector: " + ", ".join(f"{op} {n}" for op, n in sorted(detail.items(), key=lambda kv: -kv[1])))
        tot.update(cc)
    print("loop totals (every block once -- rare-branch blocks included, see the block list):")
    vec = tot["valu"] * COST["valu"] + tot["trans"] * COST["trans"] + tot["pk32"] * COST["pk32"]
    mf = tot["mfma16"] * COST["mfma16"] + tot["mfma32"] * COST["mfma32"]
    hold = (tot["mfma16"] + tot["mfma32"]) * MFMA_HOLD
    print(f"  MFMA {tot['mfma16']} x16 + {tot['mfma32']} x32 = {mf} matrix-pipe cycles; holds the vector issue {hold} cycles")
    print(f"  vector ALU {tot['valu']} plain x4 + {tot['trans']} transcendental x8 + {tot['pk32']} packed-fp32 x8 = {vec} cycles")
    print(f"  LDS {tot['lds']} x4 = {tot['lds'] * 4}; scalar {tot['salu']}; VMEM {tot['vmem']}")
    print(f"  one-wave vector-issue bound: {vec + hold + tot['lds'] * 4} cycles per iteration; matrix-pipe bound: {mf}")


if __name__ == "__main__":
    main()
