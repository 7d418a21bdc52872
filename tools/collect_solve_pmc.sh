#!/bin/bash
# GPU box: SQ / instruction-cache counters of the solve kernel (separate passes, kernel-trace only).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/solve_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" $OUT/counters.txt | sort -u > $OUT/icache_names.txt || true
cat $OUT/icache_names.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  i=$((i+1))
  for NB in "30 8192" "6 8192" "6 1024"; do
    set -- $NB
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p${i}_n$1_b$2 -- python3 tools/gpu_probe_solve_one.py $2 $1 4 > $OUT/p${i}_n$1_b$2.log 2>&1 || echo "pass $i N=$1 B=$2 failed"
  done
done
python3 - <<'P'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/solve_pmc/p*_n*_b*")):
    if not d[-1].isdigit(): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:48] + ("" if int(r.get("LDS_Block_Size", 0) or 0) < 40000 else " [tier 2]"), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d.split("/")[-1], k[0], k[1], "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)), "last=%.4g" % v[-1])
P
