#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + HBM traffic counters for bench.py's
# timed kernel (default: 64 steps of 8192 rollouts per launch), for the one-launch-per-step leg, and at
# a saturating single batch (calibration of the counters on this kernel's own access pattern).
# Raw output -> gpurun_out/prof_<tag>/, summary -> gpurun_out/profile_summary_<tag>.json.
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
COMMON="--no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-iterated --no-closed-loop"
# 1. kernel trace + stats of the DRIVER's bench command (--steps 20 --warmup 5: both legs, hipGraph replay; the launch shape does not depend on --steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
# 2./3. PMC passes (their own runs, kernel-trace only; eager launches so every dispatch is a plain kernel)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_fused -- python3 bench.py $COMMON --no-graph --no-single --steps 640 --warmup 64 --min-ms 1 > $OUT/pmc_${C}_fused.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_single -- python3 bench.py $COMMON --no-graph --steps-per-launch 1 --steps 200 --warmup 20 --min-ms 1 > $OUT/pmc_${C}_single.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_b4m -- python3 bench.py $COMMON --no-graph --steps-per-launch 1 --batch 4194304 --ring 2 --steps 20 --warmup 3 --min-ms 1 > $OUT/pmc_${C}_b4m.log 2>&1
done
# 4. the solver kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solve -- python3 bench.py --no-cpu-baseline --no-obstacle-source --no-single --no-configs --no-iterated --no-closed-loop --steps 640 > $OUT/bench_solve.log 2>&1
python3 tools/summarize_profiles.py $OUT > gpurun_out/profile_summary_$TAG.json
cat gpurun_out/profile_summary_$TAG.json
