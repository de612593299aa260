set -o pipefail
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/r6p
rm -rf $R/gpurun_out/r6p/stats1 $R/gpurun_out/r6p/stats2; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6p/stats2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --inv-steps 0 --no-extras > $R/gpurun_out/r6p/stats2.json 2> $R/gpurun_out/r6p/stats2.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6p/stats1 -- python3 $R/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --inv-steps 0 --no-extras > $R/gpurun_out/r6p/stats1.json 2> $R/gpurun_out/r6p/stats1.log
cd $R && bash tools/traffic.sh gpurun_out/r6p/hbm_traffic.json --steps 3 --warmup 1 --no-cpu-baseline --inv-steps 0 --no-extras > gpurun_out/r6p/traffic.log 2>&1
cd $R && python3 tools/step_trace.py --frames 32 --fusion fft --reps 3 > gpurun_out/r6p/step_trace.txt 2>&1
ls gpurun_out/r6p; tail -3 gpurun_out/r6p/traffic.log
