#!/bin/bash
# GPU box: SQ / TCC counters of the one-shot config-3 kernel at the bench's batched launch (64 x 8192 x horizon 50, 16 spheres) for every workgroup
# shape tools/gpu_probe_cfg3_batched.py cycles through (3 / 4 / 8 wavefronts, register and reversible sweeps, packed-VALU and matrix-core residuals: the MF = true instantiations
# are the matrix-core form, whose MFMA counters are collected in passes of their own).  Separate passes, kernel-trace only.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/cfg3b_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA" "SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_COEXEC_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p$i -- python3 tools/gpu_probe_cfg3_batched.py 64 8192 50 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/cfg3b_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "rollout_obstacles_kernel" in n:
            name = n.split("(")[0].replace("void se3mpc::", "")
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]
    print(f"{k[0]:64s} {k[1]:22s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
P
