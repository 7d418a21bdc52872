"""Developer probe (GPU box): rollout bandwidth per horizon and variant at a saturating batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); lib = ops.lib; dev = ops.be.device
B = 1 << 20
g = torch.Generator(device=dev); g.manual_seed(1)
for N in (6, 7, 13, 16, 20, 24, 30, 32, 40, 50, 64):
    prm = Params.reference_defaults(horizon=N)
    p0 = torch.randn(3, B, device=dev, generator=g); v0 = torch.randn(3, B, device=dev, generator=g); goal = torch.randn(3, B, device=dev, generator=g)
    T = torch.randn(3 * N, B, device=dev, generator=g)
    cost = torch.empty(B, device=dev); grad = torch.empty(3 * N, B, device=dev)
    line = f"N={N:3d}"
    for var in (1, 6, 3):
        lib.set_rollout_variant(var)
        for _ in range(3): ops.rollout_cost_grad(prm, p0, v0, goal, T, out=(cost, grad))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.rollout_cost_grad(prm, p0, v0, goal, T, out=(cost, grad))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  var{var}: {4*(6*N+10)*B/ms/1e9:6.3f} TB/s"
    lib.set_rollout_variant(0)
    print(line + "   (var1 = exact-N registers where instantiated else reversible; var6 = register bucket; var3 = reversible)", flush=True)
