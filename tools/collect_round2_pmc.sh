#!/bin/bash
# GPU box: SQ counters of the two round-2 kernels whose bound is NOT HBM -- the on-device iteration loop (K = 16, 8192 x N = 30) and the
# closed loop (4096 drones x 15 steps) -- to back "VALU-issue bound at one wavefront per SIMD"  and "dependent-instruction
# latency of one lane" (5.7).  Separate passes, kernel-trace only.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r2_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/iter_p$i -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-closed-loop --iterated-ks 16 --min-ms 1 > $OUT/iter_p$i.log 2>&1 || echo "iter pass $i failed"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/loop_p$i -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-iterated > $OUT/loop_p$i.log 2>&1 || echo "loop pass $i failed"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r2_pmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rollout_iterate_kernel" in n or "closed_loop_kernel" in n:
            name = n.split("(")[0].replace("void se3mpc::", "")
            acc[(name, r["Grid_Size"] if "Grid_Size" in r else "", r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]
    print(f"{k[0]:70s} grid {k[1]:>8s} {k[2]:22s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
P
