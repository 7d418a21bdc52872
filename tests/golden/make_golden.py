#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box never sees the
reference, it only sees the .npz/.json files this script wrote.  The reference's source is
imported unchanged from where it lies.  Two third-party packages the reference imports are
not installed in this image and cannot be installed (no network):

* ``pint``  -- units tagging only.  Every quantity on the hot path is an SI magnitude that
  flows straight into bare-float NumPy/SciPy arithmetic, so the
  generator puts an *identity-units* stand-in on sys.path: ``Q_(x, unit)`` returns
  ``np.asarray(x, float)``.  No arithmetic of the path lives in pint.
* ``zmq``   -- only imported when the DI container is built; never exercised here.

All arithmetic that produces the vectors is the reference's own NumPy code plus the
installed SciPy (1.15.3) L-BFGS-B.  The stand-ins are written to a temp dir and deleted.

Usage:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz, *.json)
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.environ.get("SE3MPC_GOLDEN_OUT", HERE)      # tests/test_golden_reproducible.py writes to a scratch directory
REF_SRC = "/root/reference/src"

PINT_INIT = '''
import numpy as np
class Quantity: pass                      # isinstance(x, Quantity) is always False
class UnitRegistry:
    def __init__(self, *a, **k): pass
    def setup_matplotlib(self, *a, **k): pass
    def __contains__(self, item): return True
    def define(self, s): pass
    def Quantity(self, value, unit=None): return np.asarray(value, dtype=float)
'''
PINT_ERRORS = "class DimensionalityError(Exception): pass\n"


def _install_standins():
    d = tempfile.mkdtemp(prefix="se3mpc_golden_")
    os.makedirs(os.path.join(d, "pint"))
    with open(os.path.join(d, "pint", "__init__.py"), "w") as f:
        f.write(PINT_INIT)
    with open(os.path.join(d, "pint", "errors.py"), "w") as f:
        f.write(PINT_ERRORS)
    sys.path.insert(0, d)
    sys.path.insert(1, REF_SRC)
    sys.dont_write_bytecode = True
    os.environ.setdefault("DART_ENV", "test")
    os.environ.setdefault("DART_SECRET_KEY", "golden")
    os.environ.setdefault("DART_ZMQ_SECRET", "golden")
    return d


def main():
    tmp = _install_standins()
    try:
        import logging
        logging.disable(logging.CRITICAL)
        import scipy
        import dart_planner.planning.se3_mpc_planner as ref_mod
        from dart_planner.planning.se3_mpc_planner import SE3MPCPlanner, SE3MPCConfig
        from dart_planner.common.types import DroneState, Trajectory

        real_minimize = ref_mod.minimize
        captured = {}

        def spy_minimize(*a, **k):
            res = real_minimize(*a, **k)
            captured["res"] = res
            captured["x0"] = np.array(k.get("x0", a[1] if len(a) > 1 else None), dtype=float)
            captured["bounds"] = np.array([[float(lo), float(hi)] for lo, hi in k["bounds"]])
            return res

        ref_mod.minimize = spy_minimize

        def make_planner(N, dt=None, tol=None, maxiter=None):
            kw = dict(prediction_horizon=N)
            if tol is not None:
                kw["convergence_tolerance"] = tol
            if maxiter is not None:
                kw["max_iterations"] = maxiter
            p = SE3MPCPlanner(SE3MPCConfig(**kw))
            if dt is not None:
                # the ctor forces dt = 1/400 (planner.py:99-105); a different dt is a
                # different *config*, injected the way a TimingConfig would.
                c = p.se3_config
                p.se3_config = SE3MPCConfig(
                    prediction_horizon=c.prediction_horizon, dt=dt,
                    max_iterations=c.max_iterations,
                    convergence_tolerance=c.convergence_tolerance)
            return p

        def state(p0, v0):
            return DroneState(timestamp=123.0, position=np.array(p0, float),
                              velocity=np.array(v0, float), attitude=np.zeros(3),
                              angular_velocity=np.zeros(3))

        # ------------------------------------------------------------------ solves
        rng = np.random.default_rng(20261004)
        cases = []
        cases.append(dict(N=6, p0=[0, 0, 1], v0=[0, 0, 0], goal=[5, 3, 2], tag="contract"))
        for N in (6, 20, 30, 50):
            for _ in range(6):
                cases.append(dict(N=N, p0=rng.uniform(-20, 20, 3), v0=rng.uniform(-5, 5, 3),
                                  goal=rng.uniform(-20, 20, 3), tag="random"))
        for N in (6, 30):
            p0 = rng.uniform(-5, 5, 3)
            cases.append(dict(N=N, p0=p0, v0=[0.3, -0.2, 0.1], goal=p0 + [0.05, -0.08, 0.1], tag="near_goal"))
            cases.append(dict(N=N, p0=[90, -95, 40], v0=[4, -4, 1], goal=[150, -130, 99], tag="outside_box"))
            cases.append(dict(N=N, p0=rng.uniform(-20, 20, 3), v0=rng.uniform(-5, 5, 3),
                              goal=rng.uniform(-20, 20, 3), dt=0.1, tag="dt0.1"))
        cases.append(dict(N=1, p0=[1, 2, 3], v0=[0.5, 0, 0], goal=[2, 2, 2], tag="N1"))
        cases.append(dict(N=2, p0=[1, 2, 3], v0=[0.5, 0, 0], goal=[2, 2, 2], tag="N2"))
        # tight tolerances: force the solver through many L-BFGS-B iterations (col up to 10)
        for N, tol, mi in ((6, 1e-10, 15), (6, 1e-10, 60), (20, 1e-9, 15), (30, 1e-9, 40), (50, 1e-8, 25),
                           (30, 1e-3, 15), (20, 1e-2, 15)):
            cases.append(dict(N=N, p0=rng.uniform(-20, 20, 3), v0=rng.uniform(-5, 5, 3),
                              goal=rng.uniform(-20, 20, 3), tol=tol, maxiter=mi, tag="tight"))
        cases.append(dict(N=6, p0=[0, 0, 1], v0=[0, 0, 0], goal=[5, 3, 2], tol=1e-10, maxiter=30, dt=0.1, tag="tight_dt0.1"))

        out = {}
        meta = []
        for i, c in enumerate(cases):
            pl = make_planner(c["N"], dt=c.get("dt"), tol=c.get("tol"), maxiter=c.get("maxiter"))
            evals_x, evals_f, evals_g = [], [], []
            of, og = pl._objective_function, pl._objective_gradient

            def f_spy(x, of=of):
                v = of(x); evals_x.append(np.array(x, float)); evals_f.append(float(v)); return v

            def g_spy(x, og=og):
                v = og(x); evals_g.append(np.array(v, float)); return v

            pl._objective_function, pl._objective_gradient = f_spy, g_spy
            st = state(c["p0"], c["v0"])
            tr = pl.plan_trajectory(st, np.array(c["goal"], float))
            res = captured["res"]
            k = f"s{i:02d}_"
            out[k + "p0"] = np.array(c["p0"], float)
            out[k + "v0"] = np.array(c["v0"], float)
            out[k + "goal"] = np.array(c["goal"], float)
            out[k + "x0"] = captured["x0"]
            out[k + "bounds"] = captured["bounds"]
            out[k + "evals_x"] = np.array(evals_x)
            out[k + "evals_f"] = np.array(evals_f)
            out[k + "evals_g"] = np.array(evals_g)
            out[k + "x"] = np.array(res.x, float)
            out[k + "fun"] = np.array(float(res.fun))
            for name in ("positions", "velocities", "accelerations", "attitudes", "body_rates",
                         "thrusts", "yaws", "yaw_rates"):
                out[k + name] = np.array(getattr(tr, name), float)
            out[k + "timestamps_rel"] = np.array(tr.timestamps - tr.timestamps[0], float)
            meta.append(dict(key=k, N=int(c["N"]), dt=float(pl.se3_config.dt), tag=c["tag"],
                             tol=float(pl.se3_config.convergence_tolerance),
                             maxiter=int(pl.se3_config.max_iterations),
                             nit=int(res.nit), nfev=int(res.nfev), status=int(res.status),
                             success=bool(res.success), message=str(res.message)))
        np.savez_compressed(os.path.join(OUT_DIR, "solve_cases.npz"), **out)
        with open(os.path.join(OUT_DIR, "solve_cases.json"), "w") as f:
            json.dump(dict(scipy=scipy.__version__, numpy=np.__version__, cases=meta), f, indent=1)

        # ------------------------------------------------- direct calls of path functions
        out = {}
        meta = []
        rng = np.random.default_rng(7)
        idx = 0
        for N in (1, 6, 30, 50):
            for with_goal in (True, False):
                pl = make_planner(N, dt=(0.05 if N == 6 else None))
                goal = rng.uniform(-20, 20, 3)
                if with_goal:
                    pl.set_goal(goal)
                p0, v0 = rng.uniform(-20, 20, 3), rng.uniform(-5, 5, 3)
                st = state(p0, v0)
                X = np.concatenate([rng.uniform(-120, 120, (4, 3 * N)), rng.uniform(-15, 15, (4, 3 * N)),
                                    rng.normal(0, 6, (4, 3 * N)) + np.tile([0, 0, 14.715], N)], axis=1)
                k = f"d{idx:02d}_"; idx += 1
                out[k + "p0"], out[k + "v0"], out[k + "goal"] = p0, v0, goal
                out[k + "X"] = X
                out[k + "f"] = np.array([pl._objective_function(x) for x in X])
                out[k + "g"] = np.array([pl._objective_gradient(x) for x in X])
                out[k + "dyn"] = np.array([pl._dynamics_constraints(x, st, N) for x in X])
                out[k + "phys"] = np.array([pl._physical_constraints(x, N) for x in X])
                out[k + "x0_init"] = pl._create_straight_line_initialization(st, N)
                out[k + "bounds"] = np.array([[float(a), float(b)] for a, b in pl._setup_optimization_bounds(N)])
                # obstacles as the mapper hands them over: radius 1.0, centres on the 0.5 m grid
                K = 5 if N != 30 else 16
                centres = np.round(rng.uniform(0, 15, (K, 3)) * 2) / 2
                radii = np.where(np.arange(K) % 3 == 0, 1.0, rng.uniform(0.3, 2.5, K))
                for cc, rr in zip(centres, radii):
                    pl.add_obstacle(cc, np.float64(rr))
                out[k + "obs_c"], out[k + "obs_r"] = centres, radii
                out[k + "obs"] = np.array([pl._obstacle_constraints(x, N) for x in X])
                # extraction (a11, a12) on the same X plus degenerate thrust rows
                Xe = X.copy()
                T = Xe[:, 6 * N:].reshape(4, N, 3)
                if N >= 6:
                    T[0, 1] = 0.0                      # zero thrust: attitude 0, prev_R kept
                    T[0, 3] = [7.0, 0.0, 0.0]          # thrust along +x: b1 fallback branch
                    T[1, 0] = 0.0                      # zero thrust at k=0: prev_R stays None
                    T[1, 2] = [-3.0, 0.0, 0.0]
                    T[2, 4] = [0.0, 5.0, 0.0]
                    T[3, 2] = [1e-7, 0.0, 0.0]         # below the 1e-6 threshold
                sols = [pl._extract_solution_from_result(x, N) for x in Xe]
                out[k + "Xe"] = Xe
                for name in ("accelerations", "attitudes", "body_rates", "thrusts"):
                    out[k + "ex_" + name] = np.array([s[name] for s in sols])
                em = pl._generate_emergency_trajectory(st)
                out[k + "em_positions"], out[k + "em_velocities"] = np.array(em.positions), np.array(em.velocities)
                out[k + "em_accelerations"] = np.array(em.accelerations)
                out[k + "em_timestamps"] = np.array(em.timestamps)
                meta.append(dict(key=k, N=N, dt=float(pl.se3_config.dt), with_goal=with_goal, K=K))
        # warm start (dead code in the reference; callable directly)
        pl = make_planner(8)
        pl.set_goal(np.array([4.0, -2.0, 3.0]))
        st = state([0.5, 0.2, 1.0], [0.1, 0.0, -0.1])
        for plen in (8, 5, 12):
            pl.last_solution = dict(positions=rng.uniform(-5, 5, (plen, 3)), velocities=rng.uniform(-2, 2, (plen, 3)),
                                    thrust_vectors=rng.normal(0, 2, (plen, 3)) + [0, 0, 14.715])
            k = f"w{plen:02d}_"
            for nm in ("positions", "velocities", "thrust_vectors"):
                out[k + nm] = pl.last_solution[nm]
            out[k + "p0"], out[k + "v0"], out[k + "goal"] = np.array(st.position), np.array(st.velocity), np.array(pl.goal_position)
            out[k + "x0_warm"] = pl._create_warm_start(st, 8)
        # is_plan_valid (a16)
        valid_cases = []
        base_p = np.tile([1.0, 2.0, 3.0], (5, 1))
        vc = [("ok", base_p, np.zeros((5, 3))),
              ("low_alt", base_p * [1, 1, 0.01], np.zeros((5, 3))),
              ("nan", np.where(np.arange(15).reshape(5, 3) == 7, np.nan, base_p), np.zeros((5, 3))),
              ("inf", np.where(np.arange(15).reshape(5, 3) == 2, np.inf, base_p), np.zeros((5, 3))),
              ("fast", base_p, np.where(np.arange(15).reshape(5, 3) == 4, -20.5, 0.0)),
              ("edge_v20", base_p, np.full((5, 3), 20.0)),
              ("edge_z0.1", base_p * [1, 1, 0] + [0, 0, 0.1], np.zeros((5, 3)))]
        for j, (tag, P, V) in enumerate(vc):
            tr = Trajectory(timestamps=np.arange(5) * 0.1, positions=P, velocities=V)
            out[f"v{j}_P"], out[f"v{j}_V"] = P, V
            valid_cases.append(dict(key=f"v{j}_", tag=tag, valid=bool(pl.is_plan_valid(tr))))
        np.savez_compressed(os.path.join(OUT_DIR, "path_functions.npz"), **out)
        with open(os.path.join(OUT_DIR, "path_functions.json"), "w") as f:
            json.dump(dict(scipy=scipy.__version__, numpy=np.__version__, cases=meta, valid=valid_cases,
                           constants=dict(mass=float(pl.mass), gravity=float(pl.gravity),
                                          hover=float(pl.hover_thrust), default_dt=1.0 / 400.0)), f, indent=1)
        # ---------------------------------------------- obstacle source: mapper -> sphere list
        from dart_planner.perception.explicit_geometric_mapper import ExplicitGeometricMapper
        import contextlib, io
        out = {}
        meta = []
        scenes = [
            ("dense", [([3.0, 1.0, 2.0], 1.6), ([-4.0, -2.5, 1.0], 2.2), ([6.5, 6.0, 4.0], 1.1), ([0.5, -6.0, 3.5], 0.9)], [0.0, 0.0, 2.0], 20.0, 20),
            ("sparse", [([2.0, 2.0, 2.0], 0.6)], [0.0, 0.0, 2.0], 20.0, 20),
            ("empty", [], [1.0, -1.0, 3.0], 20.0, 20),
            ("test_file_divisor", [([3.0, 0.0, 3.0], 1.5), ([7.0, 1.0, 4.0], 1.0)], [0.0, 0.0, 2.0], 15.0, 10),
        ]
        for tag, obstacles, centre, size, target in scenes:
            with contextlib.redirect_stdout(io.StringIO()):
                mapper = ExplicitGeometricMapper(resolution=0.5, max_range=40.0)      # tests/test_se3_mpc_with_mapper.py:12
            for c_, r_ in obstacles:
                mapper.add_obstacle(np.array(c_, float), r_)
            grid, occ = mapper.get_local_occupancy_grid(np.array(centre, float), size=size)
            # the three selection lines of cloud/main_improved_threelayer.py:387-398 (target 20) and of
            # tests/test_se3_mpc_with_mapper.py:29-33 (target 10), applied to the reference mapper's own output
            occupied_points = grid[occ > 0.6]
            step = max(1, occupied_points.shape[0] // target)
            chosen = occupied_points[::step] if occupied_points.size else np.zeros((0, 3))
            k = f"m_{tag}_"
            out[k + "grid"] = grid.reshape(-1, 3)
            out[k + "occ"] = occ.reshape(-1)
            out[k + "spheres"] = np.concatenate([chosen, np.ones((len(chosen), 1))], axis=1)
            meta.append(dict(key=k, centre=centre, size=size, resolution=0.5, target=target, n_occupied=int(occupied_points.shape[0]),
                             n_spheres=int(len(chosen)), num_cells=int(grid.shape[0])))
        np.savez_compressed(os.path.join(OUT_DIR, "mapper_spheres.npz"), **out)
        with open(os.path.join(OUT_DIR, "mapper_spheres.json"), "w") as f:
            json.dump(dict(cases=meta), f, indent=1)
        # ---------------------------------------------- wire format: the reference's own serializer
        from dart_planner.communication.secure_serializer import SecureSerializer
        ser = SecureSerializer(secret_key="golden-secret-key", test_mode=True, message_ttl=10 ** 9)
        payloads = [
            {"status": "ok", "n": 3, "goal": np.array([5.0, 3.0, 2.0]), "nested": {"a": [1, 2.5, 3], "b": [[1.0, 2.0], [3.0, 4.0]]}},
            {"trajectory": {"timestamps": np.arange(3) * 0.0025, "positions": np.arange(9.0).reshape(3, 3) / 7,
                            "velocities": None, "thrusts": np.array([14.7, 14.6, 14.5])}},
            [1, 2, 3],
        ]
        wire = []
        for pl_ in payloads:
            raw = ser.serialize(pl_)
            back = ser.deserialize(raw)
            wire.append(dict(raw=raw.decode("utf-8")))
        with open(os.path.join(OUT_DIR, "wire_messages.json"), "w") as f:
            json.dump(dict(secret="golden-secret-key", messages=wire), f, indent=1)
        print("wrote", os.listdir(OUT_DIR))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
