#!/usr/bin/env python3
"""One pinned problem on which L-BFGS-B's outcome is decided by the LAST BIT of an objective value -- by RUNNING THE REFERENCE ITSELF.

Found by tools/gpu_fuzz_solver.py (profiles/r02n, r02t, r02w _solver_fuzz_3000.jsonl: "worst_on_count_mismatch 0.376 m"; problem 2802 of
the re-weighted horizon-20 set, seed 1238, float32-representable inputs).  In the third iteration the line search makes 20 evaluations
whose steps shrink below rounding (f = 2728.0502535728...).  What happens next depends on the last place of f at the 4th evaluation
(2728.0502535728724 against ...873, 1.7e-16 relative):
  * the REFERENCE (its own per-step `cost += ...` loops, planner.py:516-550, + SciPy 1.15.3) runs into the 20-evaluation limit, restores
    the iterate, drops the L-BFGS memory, takes a steepest-descent step and stops at (nit, nfev) = (3, 27), f = 2680.18;
  * the same SciPy handed the oracle's BATCHED objective (np.sum over all steps at once: another summation order, the other last bit)
    ends that search with a warning task and stops on the ftol test at (3, 24), f = 2728.05 -- 0.376 m away.
Rounds 1-2 compared the HIP solver with the second and logged a "mismatch"; the solver was on the reference's branch all along.  Since
round 3 the oracle's solve() hands SciPy an objective that accumulates in the reference's order (oracle.objective_ordered, bit-identical
to the reference-shaped loops), and this fixture pins the problem with BOTH end points: the reference's (the expected answer) and the
other branch (to show what a last-bit difference does here).

Writes bifurcation_case.npz / bifurcation_case.json.  Same stand-ins as make_golden.py; runs only in the build container."""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.environ.get("SE3MPC_GOLDEN_OUT", HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_golden import _install_standins  # noqa: E402

WEIGHTS = dict(position_weight=3.0, velocity_weight=40.0, thrust_weight=2.5, acceleration_weight=0.2)
N, SEED, INDEX, BATCH = 20, 1238, 2802, 3000


def main():
    import parity_checks as pc
    from dart_planner_amd.capi import Params
    from oracle import se3mpc_oracle as orc
    rng = np.random.default_rng(SEED)
    p0, v0, goal, _ = pc.random_batch(rng, BATCH, N)
    r32 = lambda a: a[INDEX].astype(np.float32).astype(float)
    p0, v0, goal = r32(p0), r32(v0), r32(goal)
    tmp = _install_standins()
    try:
        import logging
        logging.disable(logging.CRITICAL)
        import scipy
        import dart_planner.planning.se3_mpc_planner as ref_mod
        from dart_planner.planning.se3_mpc_planner import SE3MPCPlanner, SE3MPCConfig
        from dart_planner.common.types import DroneState
        real_minimize = ref_mod.minimize
        captured = {}

        def spy_minimize(*a, **k):
            captured["res"] = real_minimize(*a, **k)
            return captured["res"]

        ref_mod.minimize = spy_minimize
        pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=N, **WEIGHTS))
        st = DroneState(timestamp=0.0, position=p0.copy(), velocity=v0.copy(), attitude=np.zeros(3), angular_velocity=np.zeros(3))
        tr = pl.plan_trajectory(st, goal.copy())
        res = captured["res"]
        dt = float(pl.se3_config.dt)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # the other branch: the same SciPy call on the oracle's batched objective (np.sum over all steps: another last bit)
    from scipy.optimize import minimize
    cfg = pc.oracle_cfg(Params.reference_defaults(horizon=N, dt=dt, **WEIGHTS))
    x0 = orc.straight_line_init(p0, v0, goal, cfg)
    alt = minimize(fun=lambda x: float(orc.objective(x, goal, cfg)), x0=x0, method="L-BFGS-B", jac=lambda x: orc.gradient(x, goal, cfg),
                   bounds=[(lo, hi) for lo, hi in orc.bounds(cfg)],
                   options={"maxiter": cfg.max_iterations, "gtol": cfg.convergence_tolerance, "ftol": cfg.convergence_tolerance * 10})
    assert (int(res.nit), int(res.nfev)) != (int(alt.nit), int(alt.nfev)), "both summation orders agree: no longer a knife edge with this SciPy / NumPy"
    np.savez_compressed(os.path.join(OUT_DIR, "bifurcation_case.npz"), p0=p0, v0=v0, goal=goal,
                        x_reference=np.array(res.x, float), positions_reference=np.array(tr.positions, float),
                        x_other_branch=np.array(alt.x, float),
                        accelerations_reference=np.array(tr.accelerations, float), thrusts_reference=np.array(tr.thrusts, float))
    with open(os.path.join(OUT_DIR, "bifurcation_case.json"), "w") as f:
        json.dump(dict(scipy=scipy.__version__, numpy=np.__version__, N=N, dt=dt, weights=WEIGHTS, fuzz_seed=SEED, fuzz_index=INDEX,
                       reference=dict(nit=int(res.nit), nfev=int(res.nfev), status=int(res.status), fun=float(res.fun)),
                       other_branch=dict(nit=int(alt.nit), nfev=int(alt.nfev), status=int(alt.status), fun=float(alt.fun)),
                       gap_m=float(np.max(np.abs(np.array(res.x)[:3 * N] - np.array(alt.x)[:3 * N])))), f, indent=1)
    print("wrote bifurcation_case.npz / .json")


if __name__ == "__main__":
    main()
