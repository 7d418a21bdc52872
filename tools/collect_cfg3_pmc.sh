#!/bin/bash
# GPU box: SQ counters of the fused rollout+obstacle kernel (config 3) at 1 M rollouts and of the plain horizon-50 rollout,
# to back the "VALU-bound on top of an HBM-bound rollout" statement of DESIGN.md section 5.3.  Separate passes, kernel-trace only.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/cfg3_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/gpu_probe_cfg3_small.py 1048576 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/cfg3_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rollout" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void se3mpc::", "")
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]
    print(f"{k[0]:60s} {k[1]:22s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
P
