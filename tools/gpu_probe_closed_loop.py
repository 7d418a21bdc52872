"""Developer probe (GPU box): the receding-horizon Monte-Carlo (bench.py's closed_loop leg: 4096 runs x 33 cycles x 15 steps) against the
solver's lanes-per-problem, with the iteration / evaluation counts of its solves.  `python tools/gpu_probe_closed_loop.py [groups]`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.control.closed_loop import ClosedLoopMonteCarlo
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
S, cycles, substeps, sim_dt = 4096, 33, 15, 0.01
prm = Params.reference_defaults()
cp, sp = ops.lib.controller_default_params(), ops.lib.simulator_default_params()
groups = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "0,8,16,32,64").split(",")]
for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
    g = torch.Generator(device=dev); g.manual_seed(5)
    p0 = torch.tensor([0.0, 0.0, 2.0], dtype=dtype, device=dev).repeat(S, 1) + 0.2 * torch.randn(S, 3, dtype=dtype, device=dev, generator=g)
    v0 = 0.3 * torch.randn(S, 3, dtype=dtype, device=dev, generator=g)
    goal = torch.tensor([8.0, 0.0, 5.0], dtype=dtype, device=dev).repeat(S, 1).contiguous()
    wind = torch.randn(S, 3, dtype=dtype, device=dev, generator=g).contiguous()
    for G in groups:
        ops.lib.set_solver_variant(G << 8)
        mc = ClosedLoopMonteCarlo(ops, prm, cp, sp)
        run = lambda: mc.run(p0, v0, goal, cycles, substeps, sim_dt, wind=wind)
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        # the first cycle's solve, for its counts
        o = ops.solve(prm, p0, v0, goal, want_trajectory=False)
        info = ops.info_to_host(o["info"])
        print(f"{name} group={G or 'auto'}: {np.median(ts) * 1e3:.3f} ms per Monte-Carlo; first-cycle solves: nit histogram "
              f"{dict(zip(*np.unique(info['nit'], return_counts=True)))}, mean nfev {info['nfev'].mean():.2f}", flush=True)
ops.lib.set_solver_variant(0)
