"""Developer probe (GPU box): a few se3mpc_extract launches at 1 M trajectories, N = 30, f32 -- for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
N, B = 30, 1 << 20
prm = Params.reference_defaults(horizon=N)
g = torch.Generator(device=dev); g.manual_seed(1)
T = torch.randn(3 * N, B, device=dev, generator=g) * 2.0
T[2::3] += 14.7
for _ in range(6):
    ops.extract(prm, T)
torch.cuda.synchronize()
