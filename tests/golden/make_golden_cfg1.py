#!/usr/bin/env python3
"""Golden solves for BASELINE.json config 1 -- the reference's own CPU case -- by RUNNING THE REFERENCE ITSELF.

Problem set: the horizon-20 configuration the cloud controller builds
(/root/reference/src/dart_planner/cloud/main_improved_se3.py:49-58: prediction_horizon=20, dt=0.1 [overwritten with
1/400 by the constructor, planner.py:99-105], max_velocity=8.0, max_acceleration=4.0, position_weight=100,
velocity_weight=10, obstacle_weight=1000, safety_margin=1.5), state p=(0,0,1), v=0, and the goal sequence of
/root/reference/tests/test_planner_performance.py:31-32 -- the contract goal (5,3,2) followed by 100 x U(-5,5)^3
(default_rng(0)) -- planned one after another by ONE planner instance, as that test does (so the 0.5 m goal
hysteresis of planner.py:197-201 is in force; the goal each solve really used is recorded).

The only solve set in the fixtures with a non-default box (|v| <= 8).  Same stand-ins as make_golden.py (identity
units for the missing `pint`); runs only in the build container.  Writes cfg1_solves.npz / cfg1_solves.json.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.environ.get("SE3MPC_GOLDEN_OUT", HERE)      # tests/test_golden_reproducible.py writes to a scratch directory
sys.path.insert(0, HERE)
from make_golden import _install_standins  # noqa: E402


def main():
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
            res = real_minimize(*a, **k)
            captured["res"] = res
            captured["bounds"] = np.array([[float(lo), float(hi)] for lo, hi in k["bounds"]])
            return res

        ref_mod.minimize = spy_minimize
        pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=20, dt=0.1, max_velocity=8.0, max_acceleration=4.0,
                                        position_weight=100.0, velocity_weight=10.0, obstacle_weight=1000.0,
                                        safety_margin=1.5))
        goals = np.concatenate([[[5.0, 3.0, 2.0]], np.random.default_rng(0).uniform(-5, 5, (100, 3))])
        st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3), attitude=np.zeros(3),
                        angular_velocity=np.zeros(3))
        out = dict(p0=np.array(st.position, float), v0=np.array(st.velocity, float), goals_asked=goals)
        used, X, fun, info = [], [], [], []
        traj = {k: [] for k in ("positions", "velocities", "accelerations", "attitudes", "body_rates", "thrusts")}
        for g in goals:
            tr = pl.plan_trajectory(st, np.array(g, float))
            res = captured["res"]
            used.append(np.array(pl.goal_position, float))
            X.append(np.array(res.x, float)); fun.append(float(res.fun))
            info.append([int(res.nit), int(res.nfev), int(res.status)])
            for k in traj:
                traj[k].append(np.array(getattr(tr, k), float))
        out.update(goals_used=np.array(used), x=np.array(X), fun=np.array(fun), info=np.array(info, dtype=np.int32),
                   bounds=captured["bounds"], **{k: np.array(v) for k, v in traj.items()})
        np.savez_compressed(os.path.join(OUT_DIR, "cfg1_solves.npz"), **out)
        c = pl.se3_config
        with open(os.path.join(OUT_DIR, "cfg1_solves.json"), "w") as f:
            json.dump(dict(scipy=scipy.__version__, numpy=np.__version__, n=len(goals), N=int(c.prediction_horizon), dt=float(c.dt),
                           max_velocity=float(c.max_velocity), max_acceleration=float(c.max_acceleration),
                           tol=float(c.convergence_tolerance), maxiter=int(c.max_iterations),
                           hysteresis_kept_previous_goal=int(np.sum(np.any(np.array(used) != goals, axis=1))),
                           nit_histogram={str(k): int(v) for k, v in zip(*np.unique(np.array(info)[:, 0], return_counts=True))}), f, indent=1)
        print("wrote cfg1_solves.npz / .json:", len(goals), "solves")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
