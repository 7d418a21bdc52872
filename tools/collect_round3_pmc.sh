#!/bin/bash
# GPU box: counters of the round-3 kernels -- the obstacle-aware iteration loop (config 3 inside the loop: 8192 x horizon 50 x 16 spheres,
# K = 16), the one-shot fused rollout + obstacle kernel beside it (fresh counters: the committed ones were round 1's), and the one-launch
# Monte-Carlo.  Separate passes, kernel-trace only.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3_pmc
rm -rf $OUT; mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/cfg3_p$i -- python3 tools/gpu_probe_cfg3_loop.py > $OUT/cfg3_p$i.log 2>&1 || echo "cfg3 pass $i failed"
  if [ $i -le 2 ]; then
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/loop_p$i -- python3 bench.py --no-primary --no-cpu-baseline --no-solve --no-obstacle-source --no-configs --no-iterated > $OUT/loop_p$i.log 2>&1 || echo "loop pass $i failed"
  fi
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3_trace -- python3 tools/gpu_probe_cfg3_loop.py > $OUT/cfg3_trace.log 2>&1
python3 - <<'P'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3_pmc/*_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(s in n for s in ("rollout_iterate", "rollout_obstacles_kernel", "monte_carlo_kernel", "closed_loop_kernel")):
            name = n.split("(")[0].replace("void se3mpc::", "")
            acc[(name, r.get("Workgroup_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]
    print(f"{k[0]:72s} wg {k[1]:>4s} {k[2]:22s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
for f in glob.glob("gpurun_out/r3_pmc/cfg3_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "se3mpc" in r["Name"]:
            print("trace:", r["Name"].split("(")[0].replace("void se3mpc::", "")[:80], "calls", r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
P
