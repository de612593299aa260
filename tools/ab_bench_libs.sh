#!/bin/bash
# Same-box A/B of two builds of libvface_hip.so on the headline bench (VFACE_HIP_LIB), interleaved: base, new, base, new, ...
# usage (GPU box): bash tools/ab_bench_libs.sh <base.so> <out-prefix> [rounds [bench args...]]
base=$1; out=$2; rounds=${3:-2}; shift 3 2>/dev/null || shift 2
for r in $(seq 1 $rounds); do
  for which in base new; do
    if [ $which = base ]; then export VFACE_HIP_LIB=$base; else unset VFACE_HIP_LIB; fi
    python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --inv-steps 0 "$@" > ${out}_${which}_$r.json 2>> ${out}.err || exit 1
    python - <<PY
import json
d=json.loads(open("${out}_${which}_$r.json").read().strip().splitlines()[-1])
bf=d["roofline"]["by_family"]
print("$which $r", round(d["value"],3), "f/s", round(d["ms_per_step"],2), "ms/step | instr", round(d["instrumented_pass"]["ms_per_step"],2), {k: round(v["ms_per_step"],2) for k,v in bf.items()}, d["config"]["timed_region_bits_equal_kernel_by_kernel"], "overlap", d["config"].get("launch_stream_overlap"), flush=True)
PY
  done
done
