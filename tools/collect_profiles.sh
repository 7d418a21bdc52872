#!/bin/bash
# Runs on the GPU box (via gpurun): the round's profile set.
#   1. rocprofv3 kernel trace + stats of the DRIVER's bench command (--steps 20 --warmup 5), every leg
#   2. HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of the timed rollout kernel at the
#      64-batch launch, the one-batch launch and a 4 M-rollout calibration batch
#   3. the solver kernel (trace)
#   4. the on-device iteration kernel: trace + traffic counters at K = 16
#   5. BASELINE configs 2 / 3 legs and the closed-loop Monte-Carlo: trace
# Raw output -> gpurun_out/prof_<tag>/, summary -> gpurun_out/profile_summary_<tag>.json.
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ROLL="--no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-iterated --no-closed-loop"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ROLL --steps 20 --warmup 5 > $OUT/bench_trace.log 2>&1
echo "[profiles] trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_fused -- python3 bench.py $ROLL --no-graph --no-single --steps 640 --warmup 64 --min-ms 1 > $OUT/pmc_${C}_fused.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_single -- python3 bench.py $ROLL --no-graph --steps-per-launch 1 --steps 200 --warmup 20 --min-ms 1 > $OUT/pmc_${C}_single.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_b4m -- python3 bench.py $ROLL --no-graph --steps-per-launch 1 --batch 4194304 --ring 2 --steps 20 --warmup 3 --min-ms 1 > $OUT/pmc_${C}_b4m.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${C}_iter16 -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-closed-loop --iterated-ks 16 --min-ms 1 > $OUT/pmc_${C}_iter16.log 2>&1
  echo "[profiles] $C passes done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solve -- python3 bench.py --no-cpu-baseline --no-obstacle-source --no-single --no-configs --no-iterated --no-closed-loop --steps 640 > $OUT/bench_solve.log 2>&1
echo "[profiles] solve done"
for K in 0 16 64; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/iter_$K -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-closed-loop --iterated-ks $K > $OUT/bench_iter_$K.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-iterated --no-closed-loop > $OUT/bench_configs.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-iterated --no-configs > $OUT/bench_loop.log 2>&1
echo "[profiles] legs done"
python3 tools/summarize_profiles.py $OUT > gpurun_out/profile_summary_$TAG.json
cat gpurun_out/profile_summary_$TAG.json
