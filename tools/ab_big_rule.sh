set -e
for rule in off min:192 min:256 eff:1.15:256 min:512; do
  VFACE_BIG_RULE=$rule python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --inv-steps 0 > gpurun_out/r5f_bench_$(echo $rule | tr ':.' '__').json 2> gpurun_out/r5f_bench.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5f_bench_$(echo $rule | tr ':.' '__').json").read().strip().splitlines()[-1])
bf=d["roofline"]["by_family"]
print("$rule", round(d["value"],3), "f/s", round(d["ms_per_step"],2), "ms/step | instr", round(d["instrumented_pass"]["ms_per_step"],2), {k: round(v["ms_per_step"],2) for k,v in bf.items()}, d["config"]["timed_region_bits_equal_kernel_by_kernel"], flush=True)
PY
done
