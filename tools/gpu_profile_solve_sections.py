"""Developer probe (GPU box): where the cycles of one solve go.  Needs tools/probes/libse3mpc_solve_profile.so (solve_kernel.hip built with
-DSE3MPC_SOLVE_PROFILE: every wavefront writes per-section s_memtime sums over its `thrust` row).  Prints the mean share per section."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["SE3MPC_LIBRARY"] = os.path.join(ROOT, "tools", "probes", "libse3mpc_solve_profile.so")
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
names = ["start-up + first evaluation", "later evaluations", "Cauchy point", "subspace minimisation", "line search (without evaluations)",
         "convergence tests + BFGS update", "results", "total"]
for N, B in ((30, 8192), (30, 1024), (6, 8192)):
    prm = Params.reference_defaults(horizon=N)
    g = torch.Generator(device=dev); g.manual_seed(5)
    p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
    goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    for prec, t in (("f32", torch.float32), ("f64", torch.float64)):
        if (N * (4 if prec == "f32" else 8)) < 64:
            continue
        for _ in range(3):
            out = ops.solve(prm, p0.to(t), v0.to(t), goal.to(t))
        torch.cuda.synchronize()
        raw = out["thrusts"].cpu().numpy().view(np.uint8).reshape(B, -1)[:, :64].copy().view(np.uint64).astype(float)
        info = ops.info_to_host(out["info"])
        tot = raw[:, 7].mean()
        print(f"N={N} B={B} {prec}: mean cycles per solve {tot:.0f} (s_memtime ticks), nit {info['nit'].mean():.2f}, nfev {info['nfev'].mean():.2f}")
        for i in range(7):
            print(f"   {names[i]:38s} {raw[:, i].mean():9.0f}  {100 * raw[:, i].mean() / tot:5.1f} %")
