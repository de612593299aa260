#!/bin/bash
# HBM traffic of the dominant kernel (implicit-GEMM conv) per launch, as MI355X_MICROARCH.md "HBM" prescribes:
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only), FETCH_SIZE doubled on gfx950 for 16-B-per-lane reads.
# usage (on the GPU box, from the repo root): bash tools/traffic.sh <out-prefix> <bench args...>
export TMPDIR=/tmp; R=$PWD; out=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py "$@" > $R/gpurun_out/pmc_$c.log 2>&1)
done
python3 - "$R" "$out" <<'PY'
import csv, glob, json, re, sys, collections
R, out = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{R}/gpurun_out/pmc_{c}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        m = re.search(r"(conv_patch_kernel|gemm256_kernel|gemm_kernel|ffn_fused_kernel|st_front_kernel|attn_kernel|gn_apply_kernel|layernorm_kernel|splitk_reduce_kernel|gn_silu_conv3x3_small_kernel|linear_small_kernel)<([^>]*)>", r["Kernel_Name"])
        key = (m.group(1) + "<" + m.group(2) + ">") if m else "other"
        agg[key].append(float(r["Counter_Value"]))
    res[c] = {k: (len(v), sum(v)) for k, v in agg.items()}
summary = {}
for k in sorted(set(res["FETCH_SIZE"]) | set(res["WRITE_SIZE"])):
    if k == "other":
        continue
    nf, f = res["FETCH_SIZE"].get(k, (0, 0.0)); nw, w = res["WRITE_SIZE"].get(k, (0, 0.0))
    n = max(nf, nw)
    summary[k] = {"launches": n, "fetch_kb_raw_per_launch": f / max(nf, 1), "write_kb_per_launch": w / max(nw, 1),
                  "hbm_bytes_per_launch": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0}
conv = [v for k, v in summary.items() if re.match(r"gemm_kernel<\w+, [12],", k) or k.startswith("conv_patch_kernel")]
tot_l = sum(v["launches"] for v in conv)
summary["_conv_all"] = {"launches": tot_l, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in conv) / max(tot_l, 1),
                        "note": "conv launches (conv_patch_kernel + gemm_kernel MODE 1|2), bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024, averaged per launch"}
p3 = [v for k, v in summary.items() if k.startswith("conv_patch_kernel") and ", 3, 3" in k and not k.endswith(", true>")]   # (", true>" = the 8x8 form)
tot3 = sum(v["launches"] for v in p3)
if tot3:
    summary["_conv_patch3"] = {"launches": tot3, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in p3) / tot3,
                               "note": "conv_patch_kernel<.., 3, 3, ..> launches of every residual form, without the 8x8 form (the dominant kernel bench.py prices)"}
sys.path.insert(0, R)
from vface_amd.utils.buildinfo import source_sha16
summary["_build"] = {"source_sha16": source_sha16(), "note": "sha256[:16] of vface_amd/csrc/{*.hip,*.hpp,*.cpp,Makefile}: bench.py quotes these figures only on the same sources"}
json.dump(summary, open(out, "w"), indent=1)
for k, v in summary.items():
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()} if isinstance(v, dict) else v)
PY
