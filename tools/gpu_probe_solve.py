"""Developer probe (GPU box): batched-solve time vs batch size (occupancy / issue-bound check)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
for N in (30, 6):
    prm = Params.reference_defaults(horizon=N)
    for B in (1, 256, 1024, 2048, 4096, 8192, 16384, 65536):
        g = torch.Generator(device=dev); g.manual_seed(5)
        p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
        v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
        goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
        for _ in range(3): ops.solve(prm, p0, v0, goal)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.solve(prm, p0, v0, goal)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"N={N} B={B}: {ms*1e3:.1f} us  -> {B/ms/1e3:.2f} M solves/s", flush=True)
