for v in 0 1 0 1; do
VFACE_SPLIT_COUPLED=$v python bench.py --frames 16 --fusion flow_fix --steps 20 --warmup 3 --no-extras --no-cpu-baseline --inv-steps 0 > gpurun_out/r5k_ff_$v.json 2>> gpurun_out/r5k_ff.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r5k_ff_$v.json").read().strip().splitlines()[-1])
print("coupled=$v", round(d["value"],3), "f/s", round(d["ms_per_step"],2), "ms/step streams", d["config"]["launch_streams"], "overlap", d["config"]["launch_stream_overlap"], "bits", d["config"]["timed_region_bits_equal_kernel_by_kernel"], flush=True)
PY
done
