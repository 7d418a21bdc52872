"""Developer probe (GPU box): batched-solve time vs batch size and vs the solver's lanes-per-problem (occupancy / issue-bound check).
`python tools/gpu_probe_solve.py [horizons] [groups]`, e.g. `6,30 0,8,16,32,64` (group 0 = the library's own choice)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
horizons = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "30,6").split(",")]
groups = [int(s) for s in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
batches = [int(s) for s in (sys.argv[3] if len(sys.argv) > 3 else "1,256,1024,2048,4096,8192,16384,65536").split(",")]
dtypes = {"f32": torch.float32, "f64": torch.float64}
for name in (sys.argv[4] if len(sys.argv) > 4 else "f32").split(","):
    dt = dtypes[name]
    for N in horizons:
        prm = Params.reference_defaults(horizon=N)
        for G in groups:
            if G and G < N:
                continue
            ops.lib.set_solver_variant(G << 8)
            for B in batches:
                g = torch.Generator(device=dev); g.manual_seed(5)
                p0 = (torch.rand(B, 3, device=dev, generator=g) * 40 - 20).to(dt)
                v0 = (torch.rand(B, 3, device=dev, generator=g) * 10 - 5).to(dt)
                goal = (torch.rand(B, 3, device=dev, generator=g) * 40 - 20).to(dt)
                for _ in range(3): ops.solve(prm, p0, v0, goal)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.solve(prm, p0, v0, goal)
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                print(f"{name} N={N} group={G or 'auto'} B={B}: {ms*1e3:.1f} us  -> {B/ms/1e3:.2f} M solves/s", flush=True)
ops.lib.set_solver_variant(0)
