"""Reduce the rocprofv3 CSVs written by tools/collect_profiles.sh to one JSON summary."""
import csv
import glob
import json
import os
import sys

import numpy as np

root = sys.argv[1]
N, B = 30, 8192


def find(sub, pattern):
    f = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return f[0] if f else None


def rows_of(sub, needle):
    f = find(sub, "*kernel_trace.csv")
    return [] if not f else [r for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"]]


def durations(rows, grid_y=None):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if grid_y is None or int(r["Grid_Size_Y"]) == grid_y]
    if not d:
        return None
    d = np.array(d[len(d) // 10:], dtype=float)     # drop the warm-up tenth
    return dict(calls=int(len(d)), avg_ns=float(d.mean()), median_ns=float(np.median(d)), min_ns=float(d.min()), max_ns=float(d.max()))


def counter(sub, needle, name, grid_y=None, grid_size=None):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None
    vals = []
    for r in csv.DictReader(open(f)):
        if needle in r["Kernel_Name"] and r["Counter_Name"] == name and (grid_size is None or int(r["Grid_Size"]) == grid_size):
            vals.append(float(r["Counter_Value"]))
    if not vals:
        return None
    vals = np.array(vals)
    if grid_y is not None:                # keep the dispatches of the dominant (largest) size
        vals = vals[vals > 0.5 * vals.max()]
    vals = vals[len(vals) // 10:]
    return dict(dispatches=int(len(vals)), mean=float(vals.mean()), median=float(np.median(vals)))


tr = rows_of("trace", "rollout_kernel")
out = {"rollout_fused_64x8192_graph": durations(tr, 64), "rollout_single_8192_graph": durations(tr, 1),
       "solve_kernel_single_problem": None, "solve_kernel_batch_8192": None}
sv = rows_of("solve", "solve_kernel")
if sv:
    # the packed solver: 64 / G problems per wavefront (G = 32 lanes per problem at horizon 30; a lone problem takes a whole wavefront).  A
    # launch whose wavefronts cannot all be resident with the full L-BFGS memory in LDS is two back-to-back launches: tier 1 (LDS for 2-4
    # pairs, all problems) and tier 2 (full LDS, every group but the overflowed ones exits at once) -- alternate dispatches of the same grid.
    single = sorted((r for r in sv if int(r["Grid_Size_X"]) == 64), key=lambda r: int(r["Dispatch_Id"]))
    out["solve_kernel_single_problem_f64"] = durations([r for r in single if "<double" in r["Kernel_Name"]])
    out["solve_kernel_single_problem_f32"] = durations([r for r in single if "<float" in r["Kernel_Name"]])
    grp = sorted((r for r in sv if int(r["Grid_Size_X"]) == 32 * B), key=lambda r: int(r["Dispatch_Id"]))
    out["solve_kernel_batch_8192_tier1"] = durations(grp[0::2])
    out["solve_kernel_batch_8192_tier2_empty"] = durations(grp[1::2])
    out["solve_kernel_registers"] = {"VGPR_Count": grp[0].get("VGPR_Count"), "Accum_VGPR_Count": grp[0].get("Accum_VGPR_Count"), "Scratch_Size": grp[0].get("Scratch_Size"),
                                     "LDS_Block_Size_tier1": grp[0].get("LDS_Block_Size")} if grp else None
    del out["solve_kernel_single_problem"], out["solve_kernel_batch_8192"]
for tag, rollouts, gy in (("fused", 64 * B, 64), ("single", B, None), ("b4m", 4194304, None)):
    alg_r = 4 * (3 * N + 9) * rollouts
    alg_w = 4 * (3 * N + 1) * rollouts
    fe = counter(f"pmc_FETCH_SIZE_{tag}", "rollout_kernel", "FETCH_SIZE", gy)
    wr = counter(f"pmc_WRITE_SIZE_{tag}", "rollout_kernel", "WRITE_SIZE", gy)
    out[f"pmc_{tag}"] = dict(rollouts_per_launch=rollouts, algorithmic_read_bytes=alg_r, algorithmic_write_bytes=alg_w,
                            FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr,
                            kernel_ns_under_pmc=durations(rows_of(f"pmc_FETCH_SIZE_{tag}", "rollout_kernel"), gy),
                            fetch_bytes_raw=None if fe is None else fe["mean"] * 1024,
                            write_bytes_raw=None if wr is None else wr["mean"] * 1024)
# ---- the on-device iteration kernel: duration per launch at K = 0 / 16 / 64 and its HBM traffic at K = 16
it = {}
for K in (0, 16, 64):
    it[f"K{K}"] = durations(rows_of(f"iter_{K}", "rollout_iterate_kernel"), 1)          # grid.y = 1: the single-batch launches
if it.get("K0") and it.get("K16") and it.get("K64"):
    it["us_per_iteration_K16"] = (it["K16"]["avg_ns"] - it["K0"]["avg_ns"]) / 16 / 1e3
    it["us_per_iteration_K64"] = (it["K64"]["avg_ns"] - it["K0"]["avg_ns"]) / 64 / 1e3
fe = counter("pmc_FETCH_SIZE_iter16", "rollout_iterate_kernel", "FETCH_SIZE", grid_size=192 * (B // 64))
wr = counter("pmc_WRITE_SIZE_iter16", "rollout_iterate_kernel", "WRITE_SIZE", grid_size=192 * (B // 64))
it["pmc_K16"] = dict(trajectories_per_launch=B, iterations=16, rollouts_per_launch=17 * B,
                     hbm_algorithmic_read_bytes=4 * (3 * N + 9) * B, hbm_algorithmic_write_bytes=4 * (6 * N + 1) * B,
                     FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr,
                     fetch_bytes_x2=None if fe is None else fe["mean"] * 1024 * 2, write_bytes=None if wr is None else wr["mean"] * 1024)
if fe is not None and wr is not None:
    it["pmc_K16"]["traffic_bytes_per_iteration"] = (fe["mean"] * 2048 + wr["mean"] * 1024) / 17
    it["pmc_K16"]["stand_alone_launch_bytes_per_iteration"] = 4 * (6 * N + 10) * B
it["batched_64xK16"] = durations(rows_of("iter_16", "rollout_iterate_kernel"), 64)
out["rollout_iterate"] = it
# ---- config legs and the closed loop
cf = {}
for name, needle, gy in (("cfg2_single_1024", "rollout_kernel", 1), ("cfg2_batched_64x1024", "rollout_kernel", 64),
                         ("cfg3_single_8192", "rollout_obstacles_kernel", 1), ("cfg3_batched_64x8192", "rollout_obstacles_kernel", 64)):
    cf[name] = durations(rows_of("configs", needle), gy)
out["configs"] = cf
lsv = sorted((r for r in rows_of("loop", "solve_kernel") if int(r["Grid_Size_X"]) == 8 * 4096), key=lambda r: int(r["Dispatch_Id"]))
out["closed_loop"] = {"closed_loop_kernel_4096x15": durations(rows_of("loop", "closed_loop_kernel")),
                      "solve_kernel_4096_h6_tier1": durations(lsv[0::2]), "solve_kernel_4096_h6_tier2_empty": durations(lsv[1::2]),
                      "monte_carlo_kernel_4096x33x15_f32": durations([r for r in rows_of("loop", "monte_carlo_kernel") if "<float" in r["Kernel_Name"]]),
                      "monte_carlo_kernel_4096x33x15_f64": durations([r for r in rows_of("loop", "monte_carlo_kernel") if "<double" in r["Kernel_Name"]])}
_obs = rows_of("iter_16", "rollout_iterate_obstacles_kernel")          # the leg times K = 0 / 1 / 4 / 16 in turn: keep the K = 16 launches (the longest cluster)
if _obs:
    _d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in _obs)
    _mx = _d[int(0.98 * (len(_d) - 1))]                                   # (a cold first launch is not the cluster)
    _obs = [r for r in _obs if 0.8 * _mx < int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 1.25 * _mx]
out["rollout_iterate_obstacles_cfg3_K16"] = durations(_obs)
# the shooting-form plan's two launches (solve leg): the descent (horizon 30, 8192 samples, K = 16; plain and around 16 spheres) and the one-wavefront tail
out["shooting_plan"] = {"rollout_iterate_kernel_K16": durations([r for r in rows_of("solve", "rollout_iterate_kernel") if int(r["Grid_Size_X"]) == 192 * (B // 64)]),
                        "rollout_iterate_obstacles_kernel_K16": durations(rows_of("solve", "rollout_iterate_obstacles_kernel")),
                        "shooting_finish_kernel": durations(rows_of("solve", "shooting_finish_kernel"))}
print(json.dumps(out, indent=1))
