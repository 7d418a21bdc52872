"""GPU box: every lane-layout kernel (all rollout variants, fused obstacle kernel, parity-form kernels, the on-device iteration loop) against the oracle
at EVERY horizon 1..64, f64 and f32 (tests/parity_checks.check_lane_kernels; the test-suite samples 11 horizons)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity_checks as pc
from dart_planner_amd.ops import Ops, TorchBackend
ops = Ops(TorchBackend("cuda:0"))
def harness(dt):
    return pc.Harness(ops, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"), lambda a: a.detach().cpu().numpy(), dt)
t0 = time.time(); done = []
for N in range(1, 65):
    for dt in (np.float64, np.float32):
        pc.check_lane_kernels(harness(dt), N, 64 + (N * 7) % 131, seed=N, variants=(0, 1, 2, 3, 4, 5, 6))
        pc.check_rollout_iterate(harness(dt), N, 64 + (N * 5) % 97, seed=N, iters=1 + N % 7)
    done.append(N)
    if N % 8 == 0: print("horizons", done[-8], "..", N, "ok", flush=True)
print(json.dumps(dict(horizons="1..64", dtypes=["float64", "float32"], variants=[0, 1, 2, 3, 4, 5, 6], result="all checks passed",
                      seconds=round(time.time() - t0, 1))))
