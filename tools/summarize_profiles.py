"""Reduce the rocprofv3 CSVs written by tools/collect_profiles.sh to one JSON summary."""
import csv
import glob
import json
import os
import sys

import numpy as np

root = sys.argv[1]


def find(sub, pattern):
    f = glob.glob(os.path.join(root, sub, "**", pattern), recursive=True)
    return f[0] if f else None


def kernel_durations(sub, needle):
    f = find(sub, "*kernel_trace.csv")
    if not f:
        return None
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"]]
    if not d:
        return None
    d = np.array(d[len(d) // 10:], dtype=float)     # drop the warm-up tenth
    return dict(calls=int(len(d)), avg_ns=float(d.mean()), median_ns=float(np.median(d)), min_ns=float(d.min()), max_ns=float(d.max()))


def counter(sub, needle, name):
    f = find(sub, "*counter_collection.csv")
    if not f:
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if needle in r["Kernel_Name"] and r["Counter_Name"] == name]
    if not vals:
        return None
    vals = np.array(vals[len(vals) // 10:])
    return dict(dispatches=int(len(vals)), mean=float(vals.mean()), median=float(np.median(vals)))


out = {"rollout_b8192_graph": kernel_durations("trace", "rollout_kernel"), "solve_kernel": kernel_durations("solve", "solve_kernel")}
N = 30
for tag, B in (("b8192", 8192), ("b4m", 4194304)):
    alg_r = 4 * (3 * N + 9) * B
    alg_w = 4 * (3 * N + 1) * B
    fe = counter(f"pmc_FETCH_SIZE_{tag}", "rollout_kernel", "FETCH_SIZE")
    wr = counter(f"pmc_WRITE_SIZE_{tag}", "rollout_kernel", "WRITE_SIZE")
    dur = kernel_durations(f"pmc_FETCH_SIZE_{tag}", "rollout_kernel")
    out[f"pmc_{tag}"] = dict(batch=B, algorithmic_read_bytes=alg_r, algorithmic_write_bytes=alg_w, FETCH_SIZE_KB=fe, WRITE_SIZE_KB=wr,
                            kernel_ns_under_pmc=dur,
                            fetch_bytes_raw=None if fe is None else fe["mean"] * 1024,
                            write_bytes_raw=None if wr is None else wr["mean"] * 1024)
print(json.dumps(out, indent=1))
