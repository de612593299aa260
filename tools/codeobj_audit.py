#!/usr/bin/env python3
"""Static audit of the gfx950 code objects in ``vface_amd/csrc/build/*.o`` (no GPU needed).

For every kernel: the code-object notes (unified VGPR count, SGPRs, scratch, static LDS, workgroup size) -> how many waves of it fit
on a SIMD and how many registers of the 512 per lane a resident wave leaves for ANOTHER kernel's wave; and a scan of the
disassembly for the instruction classes round 5's co-residency misread involved (HISTORY.md R5, profiles/r05_f_*):

  vcc_select   ``v_cndmask_b32`` whose lane mask is VCC (the instruction that returned the wrong branch in lanes 48-63 of the flow warp)
  sgpr_select  the same select on an SGPR pair (the form that was right in the same kernel)
  vcc_carry    ``v_addc / v_subb / v_subbrev`` (``_co_``) consuming VCC as carry-in
  div_fmas     ``v_div_fmas`` (implicit VCC read)
  setprio      ``s_setprio`` (a wave that raises its issue priority over co-resident waves)
  trans        ``v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos`` (quarter-rate transcendental unit)
  mfma         matrix-core instructions
  dpp          DPP / ``v_readlane`` / ``v_permlane`` cross-lane forms
  lds_dma      ``buffer_load ... lds`` / ``global_load_lds``
  pk32         packed-fp32 instructions (``v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_mov_b32``: 64-bit register pairs)
  pk32_e32     ... of those, the consumer is a 32-bit-encoded (VOP1 / VOP2 / VOPC, ``_e32`` / SDWA / DPP) instruction -- the failing pair
               was ``v_pk_mul_f32 v[40:41]`` -> ``v_cndmask_b32_e32 v40, 0, v40, vcc``; the VOP3 / VOP3P consumers two slots behind a
               packed product in the same kernel (``v_cvt_pk_f16_f32``, ``v_pk_fma_f32``) never produced a wrong value
  pk32_d1..d3  ... whose result (either half) a VECTOR-ALU instruction reads 1 / 2 / 3 issue slots later (d1 = the next instruction).
               Round 6 (tools/select_hazard_probe.py, profiles/r06_b_*): THIS is what made the old flow warp the victim -- its select read
               the low half of a v_pk_mul_f32 two slots after it (d2); the same kernel built without packed fp32, VCC-masked select
               and all, is clean beside every aggressor.

usage:  tools/codeobj_audit.py [--csv out.csv] [--victims] [--match REGEX]
  --victims   only kernels with at least one VCC-masked select / carry (the table committed as profiles/r06_*_vcc_select_audit.txt)
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BUNDLE = "hipv4-amdgcn-amd-amdhsa--gfx950"
VGPRS_PER_LANE = 512        # unified vector register file of a gfx950 SIMD, per lane (MI355X_MICROARCH.md)
GRANULE = 8


def extract(obj, outdir):
    """The gfx950 code object embedded in a host object (llvm-objdump --offloading writes the bundles beside its input)."""
    tmp = os.path.join(outdir, os.path.basename(obj))
    with open(obj, "rb") as f, open(tmp, "wb") as g:
        g.write(f.read())
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = f"{tmp}.0.{BUNDLE}"
    return co if os.path.exists(co) else None


def notes(co):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    a, b = txt.index("---"), txt.rindex("...")
    return {k[".name"]: k for k in yaml.safe_load(txt[a:b])["amdhsa.kernels"]}


PATTERNS = {
    "vcc_select": re.compile(r"^v_cndmask_b32\S*\s.*\bvcc\b"),
    "sgpr_select": re.compile(r"^v_cndmask_b32\S*\s.*\bs\[\d+:\d+\]"),
    "vcc_carry": re.compile(r"^v_(addc|subb|subbrev)_co_u32\S*\s.*,\s*vcc\s*$"),
    "div_fmas": re.compile(r"^v_div_fmas"),
    "setprio": re.compile(r"^s_setprio"),
    "trans": re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    "mfma": re.compile(r"^v_(mfma|smfmac)"),
    "dpp": re.compile(r"(\bdpp\b|row_shr|row_shl|row_bcast|quad_perm|^v_readlane|^v_readfirstlane|^v_permlane)"),
    "lds_dma": re.compile(r"^(buffer_load\S*\s.*\blds\b|global_load_lds)"),
}


PK32 = re.compile(r"^v_pk_(mul|add|fma)_f32|^v_pk_mov_b32")
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def vregs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return out


def scan(co, sites=None):
    """{kernel symbol: {class: count, 'insts': n}} from the disassembly.  ``sites``: a list that receives (kernel, distance, producer,
    consumer) for every packed-fp32 result read by a vector-ALU instruction within three issue slots (straight-line order; a label or a
    branch between the two resets the window)."""
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    res, cur, name, window = {}, None, None, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            if not name.startswith("_Z") and cur is not None and not name.endswith(".kd"):
                window = []          # a basic-block label inside the current kernel
                continue
            cur = res.setdefault(name, {k: 0 for k in PATTERNS} | {"insts": 0, "pk32": 0, "pk32_d1": 0, "pk32_d2": 0, "pk32_d3": 0, "pk32_e32": 0})
            window = []
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith("//"):
            continue
        t = t.split("//")[0].strip()
        cur["insts"] += 1
        for k, rx in PATTERNS.items():
            if rx.search(t):
                cur[k] += 1
        op, _, rest = t.partition(" ")
        if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_barrier" or op.startswith("s_endpgm"):
            window = []
            continue
        if op == "s_nop":
            n = int(rest.strip() or 0) + 1
            window = (window + [None] * n)[-3:]
            continue
        args = [a.strip() for a in rest.split(",")]
        if op.startswith("v_"):
            srcs = set().union(*(vregs(a) for a in args[1:])) if len(args) > 1 else set()
            if op.startswith("v_mfma") or op.startswith("v_smfmac") or op.startswith("v_fmac") or op.startswith("v_pk_fma") or "_fmac_" in op:
                srcs |= vregs(args[0])
            for dist, prod in enumerate(reversed(window), 1):
                if prod is not None and prod[1] & srcs:
                    cur[f"pk32_d{dist}"] += 1
                    if op.endswith("_e32") or op.endswith("_sdwa") or op.endswith("_dpp"):
                        cur["pk32_e32"] += 1          # the consumer is a 32-bit-encoded (VOP1 / VOP2 / VOPC) instruction, as the one that failed
                    if sites is not None:
                        sites.append((name, dist, prod[0], t))
                    break
        if PK32.match(op):
            cur["pk32"] += 1
            window = (window + [(t, vregs(args[0]))])[-3:]
        else:
            window = (window + [None])[-3:]
    return res


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, p.stdout.splitlines()))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv")
    ap.add_argument("--victims", action="store_true")
    ap.add_argument("--match", default="")
    ap.add_argument("--pk32", action="store_true", help="only kernels with a packed-fp32 result read within three issue slots")
    ap.add_argument("--sites", action="store_true", help="also list every such producer / consumer pair")
    ap.add_argument("--objs", default=os.path.join(ROOT, "vface_amd", "csrc", "build", "*.o"))
    a = ap.parse_args()
    rows, sites = [], []
    with tempfile.TemporaryDirectory() as td:
        for obj in sorted(glob.glob(a.objs)):
            co = extract(obj, td)
            if co is None:
                continue
            nt, sc = notes(co), scan(co, sites)
            dm = demangle(list(nt))
            for sym, k in nt.items():
                s = sc.get(sym, {})
                total = int(k[".vgpr_count"])           # gfx90a+: the unified count (arch VGPRs + AGPRs)
                alloc = max(GRANULE, -(-total // GRANULE) * GRANULE)
                waves = min(8, VGPRS_PER_LANE // alloc)
                wg_waves = -(-int(k[".max_flat_workgroup_size"]) // 64)
                rows.append({"file": os.path.basename(obj).split(".")[0], "kernel": short(dm[sym]), "vgpr": total, "agpr": int(k[".agpr_count"]),
                             "sgpr": int(k[".sgpr_count"]), "scratch": int(k[".private_segment_fixed_size"]),
                             "lds_static": int(k[".group_segment_fixed_size"]), "wg_waves": wg_waves, "waves_per_simd": waves,
                             # registers per lane a SIMD still has free with ONE / with the maximum number of this kernel's waves resident
                             "free_with_1": VGPRS_PER_LANE - alloc, "free_when_full": VGPRS_PER_LANE - waves * alloc, **s})
    if a.match:
        rx = re.compile(a.match)
        rows = [r for r in rows if rx.search(r["kernel"])]
    if a.victims:
        rows = [r for r in rows if r.get("vcc_select", 0) + r.get("vcc_carry", 0) + r.get("div_fmas", 0) > 0]
    cols = ["file", "kernel", "vgpr", "agpr", "sgpr", "scratch", "lds_static", "wg_waves", "waves_per_simd", "free_with_1", "free_when_full", "insts",
            "vcc_select", "sgpr_select", "vcc_carry", "div_fmas", "setprio", "trans", "mfma", "dpp", "lds_dma", "pk32", "pk32_d1", "pk32_d2", "pk32_d3", "pk32_e32"]
    if a.pk32:
        rows = [r for r in rows if r.get("pk32_d1", 0) + r.get("pk32_d2", 0) + r.get("pk32_d3", 0) > 0]
    if a.csv:
        import csv
        with open(a.csv, "w", newline="") as f:
            w = csv.DictWriter(f, cols)
            w.writeheader()
            for r in rows:
                w.writerow({c: r.get(c, "") for c in cols})
    wk = max((len(r["kernel"]) for r in rows), default=10)
    wk = min(wk, 110)
    print(f"{'kernel':{wk}} " + " ".join(f"{c:>9}" for c in cols[2:]))
    for r in rows:
        print(f"{r['kernel'][:wk]:{wk}} " + " ".join(f"{r.get(c, ''):>9}" for c in cols[2:]))
    print(f"{len(rows)} kernels")
    if a.sites:
        dm = demangle(sorted({k for k, *_ in sites}))
        keep = {r["kernel"] for r in rows}
        for k, dist, prod, cons in sites:
            if short(dm[k]) in keep:
                print(f"  d{dist}  {short(dm[k])[:70]:70s}  {prod}   ->   {cons}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
