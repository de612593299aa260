#!/bin/bash
# Same-box A/B of an environment switch on the headline bench, interleaved: off, on, off, on.
# usage (GPU box): bash tools/ab_env.sh <VAR> <value-A> <value-B> <out-prefix> [bench args...]
var=$1; va=$2; vb=$3; out=$4; shift 4
for r in 1 2; do
  for v in "$va" "$vb"; do
    env $var=$v python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --inv-steps 0 "$@" > ${out}_${v}_$r.json 2>> ${out}.err || exit 1
    python - <<PY
import json
d=json.loads(open("${out}_${v}_$r.json").read().strip().splitlines()[-1])
bf=d["roofline"]["by_family"]
print("$var=$v run $r:", round(d["value"],3), "f/s", round(d["ms_per_step"],2), "ms/step | instr", round(d["instrumented_pass"]["ms_per_step"],2), {k: round(x["ms_per_step"],2) for k,x in bf.items()}, d["config"]["timed_region_bits_equal_kernel_by_kernel"], flush=True)
PY
  done
done
