#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, separate passes, kernel-trace only; MI355X_MICROARCH.md's gfx950 correction) of every lane-layout
# kernel at 1 M trajectories (tools/gpu_probe_lane.py) against its algorithmic bytes -- the parity-form kernels' "traffic vs algorithmic" evidence.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/lane_pmc
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 tools/gpu_probe_lane.py > $OUT/$C.log 2>&1
done
python3 - <<'P'
import csv, glob, collections, json
N, B, K = 30, 1 << 20, 16
f4 = 4 * B
alg = {"init_kernel": f4 * (9 + 9 * N), "init4_kernel": f4 * (9 + 9 * N), "cost_grad_kernel<float, true>": f4 * (18 * N + 4), "cost_grad_kernel<float, false>": f4 * (9 * N + 4),
       "dynamics_residual_kernel": f4 * (15 * N + 6), "obstacle_residual_kernel": f4 * (3 * N + N * K + 2), "obstacle_residual4_kernel": f4 * (3 * N + N * K + 2),
       "obstacle_reduce_kernel": f4 * (3 * N + 2), "physical_constraints_kernel": f4 * (10 * N), "extract_kernel": f4 * (13 * N), "is_plan_valid_kernel": f4 * (6 * N + 1),
       "transpose_kernel": f4 * 18 * N}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/lane_pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for key in alg:
                base, _, targ = key.partition("<")
                if base in n and (not targ or ("<" + targ) in n.replace("se3mpc::", "")) and int(r.get("Grid_Size", "0") or 0) >= (B // 4):
                    acc[key][c].append(float(r["Counter_Value"]))
out = {}
for key, d in acc.items():
    if d["FETCH_SIZE"] and d["WRITE_SIZE"]:
        fx2 = 2 * 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]); wr = 1024 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out[key] = dict(launches=len(d["FETCH_SIZE"]), read_bytes=fx2, write_bytes=wr, algorithmic_bytes=alg[key], traffic_over_algorithmic=(fx2 + wr) / alg[key])
print(json.dumps(out, indent=1))
P
