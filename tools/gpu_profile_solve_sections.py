"""Developer probe (GPU box): where the cycles of one solve go.  Needs tools/probes/libse3mpc_solve_profile.so (solve_kernel.hip built with
-DSE3MPC_SOLVE_PROFILE by tools/build_solve_profile.sh: every wavefront writes per-section s_memtime sums over its `attitudes` row).  Prints the mean share per section."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["SE3MPC_LIBRARY"] = os.path.join(ROOT, "tools", "probes", "libse3mpc_solve_profile.so")
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
names = ["start-up + first evaluation", "later evaluations", "Cauchy point: rest", "subspace: cmprlb, subsm, projection", "line search (without evaluations)",
         "convergence tests + BFGS update", "results", "total", "Cauchy: pass 1 (breakpoints)", "Cauchy: closed-form pass", "Cauchy: p = W'd + first bmv",
         "Cauchy: breakpoint loop", "subspace: formk + factor", "line search: set-up (d, dtd, stpmx)", "-", "crossings"]
for N, B, G in ((30, 8192, 32), (30, 8192, 64), (6, 8192, 8), (6, 8192, 16), (6, 8192, 64)):
    prm = Params.reference_defaults(horizon=N)
    ops.lib.set_solver_variant(G << 8)
    P = 64 // G                                   # problems per wavefront: lane 0 reports into the row of the wavefront's first problem
    g = torch.Generator(device=dev); g.manual_seed(5)
    p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
    goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    for prec, t in (("f32", torch.float32), ("f64", torch.float64)):
        if (3 * N * (4 if prec == "f32" else 8)) < 128:
            continue
        for _ in range(3):
            out = ops.solve(prm, p0.to(t), v0.to(t), goal.to(t))
        torch.cuda.synchronize()
        raw = out["attitudes"].cpu().numpy().view(np.uint8).reshape(B, -1)[::P, :128].copy().view(np.uint64).astype(float)
        info = ops.info_to_host(out["info"])
        tot = raw[:, 7].mean()
        print(f"N={N} B={B} group={G} {prec}: mean cycles per wavefront {tot:.0f} (s_memtime ticks), nit {info['nit'].mean():.2f}, nfev {info['nfev'].mean():.2f}, "
              f"breakpoints crossed one by one {raw[:, 15].mean():.2f}")
        for i in (0, 1, 8, 9, 10, 11, 2, 12, 3, 13, 4, 5, 6):
            print(f"   {names[i]:42s} {raw[:, i].mean():9.0f}  {100 * raw[:, i].mean() / tot:5.1f} %")
