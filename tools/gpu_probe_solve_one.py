"""Developer probe (GPU box): a handful of batched solves, for `rocprofv3 --pmc ...` passes over the solve kernel.
usage: python3 tools/gpu_probe_solve_one.py [B] [N] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ops = Ops(); dev = ops.be.device
prm = Params.reference_defaults(horizon=N)
g = torch.Generator(device=dev); g.manual_seed(5)
p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
for _ in range(reps):
    ops.solve(prm, p0, v0, goal)
torch.cuda.synchronize()
print("done", B, N, reps)
