"""CPU only: the two largest disagreements of the 840 000-problem solver fuzz (profiles/r03d_solver_fuzz_20000.jsonl) are line-search knife edges,
shown without the GPU.  (1) problem 6437 of the re-weighted option set (kernel (nit, nfev) = (3, 26), SciPy (3, 24), end points 1.8 m apart): SciPy ALONE
lands on the kernel's branch when the same objective is summed in another order (np.sum over the horizon instead of the reference's loop order).
(2) problem 1883 (kernel (3, 24), SciPy (3, 27), 0.30 m): the kernel with the published sequential Cauchy search (se3mpc_set_solver_variant(1), host
emulation of the product sources) takes (3, 26) and SciPy's end point -- the default closed-form Cauchy point differs from SciPy's accumulated one by
~1e-9 relative and that decides a later safeguarded step.  usage: python tools/cpu_knife_edge_cases.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    sys.path.insert(0, p)
import build_emu                                   # noqa: E402  (test infrastructure: the product kernels compiled for the host)
from numpy_backend import NumpyBackend             # noqa: E402
from scipy.optimize import minimize                # noqa: E402
import parity_checks as pc                         # noqa: E402
from dart_planner_amd import capi                  # noqa: E402
from dart_planner_amd.ops import Ops               # noqa: E402
from oracle import se3mpc_oracle as orc            # noqa: E402

N = 20
prm = capi.Params.reference_defaults(horizon=N, position_weight=3.0, velocity_weight=40.0, thrust_weight=2.5, acceleration_weight=0.2)
cfg = pc.oracle_cfg(prm)
p0, v0, goal, _ = pc.random_batch(np.random.default_rng(1238), 20000, N)       # the fuzz's draw for this option set (seed 1238, 20000 problems)
box = [(lo, hi) for lo, hi in orc.bounds(cfg)]


def scipy_run(i, fun):
    res = minimize(fun=lambda x: fun(x, goal[i], cfg), x0=orc.straight_line_init(p0[i], v0[i], goal[i], cfg), method="L-BFGS-B",
                   jac=lambda x: orc.gradient(x, goal[i], cfg), bounds=box,
                   options={"maxiter": cfg.max_iterations, "gtol": cfg.convergence_tolerance, "ftol": cfg.convergence_tolerance * 10})
    return res


ops = Ops(NumpyBackend(), capi.Library(build_emu.build()))
for i in (6437, 1883):
    a, b = scipy_run(i, orc.objective_ordered), scipy_run(i, orc.objective)
    print(f"problem {i}: SciPy, reference loop order: (nit, nfev) = ({a.nit}, {a.nfev}) f = {a.fun:.10f} | SciPy, np.sum order: ({b.nit}, {b.nfev}) f = {b.fun:.10f} | "
          f"end points {np.abs(a.x - b.x)[:3 * N].max():.3g} m apart")
    for variant, name in ((0, "closed-form Cauchy point (default)"), (1, "published sequential search")):
        ops.lib.set_solver_variant(variant)
        out = ops.solve(prm, np.ascontiguousarray(p0[i:i + 1]), np.ascontiguousarray(v0[i:i + 1]), np.ascontiguousarray(goal[i:i + 1]))
        info = ops.info_to_host(out["info"])[0]
        x = np.asarray(out["x"], float)[0]
        print(f"    kernel (host emulation, float64), {name}: ({int(info['nit'])}, {int(info['nfev'])}) f = {float(info['fun']):.10f}; "
              f"from SciPy/loop order {np.abs(x[:3 * N] - a.x[:3 * N]).max():.3g} m, from SciPy/np.sum order {np.abs(x[:3 * N] - b.x[:3 * N]).max():.3g} m")
ops.lib.set_solver_variant(0)
