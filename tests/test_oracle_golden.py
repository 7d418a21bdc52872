"""The oracle (oracle/se3mpc_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py) and the SURVEY.md Appendix-B known answer.  CPU only."""
import os

import numpy as np
import pytest

from oracle import se3mpc_oracle as orc
from oracle.se3mpc_oracle import OracleConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TOL = dict(rtol=1e-12, atol=1e-10)


def cfg_for(c, **kw):
    return OracleConfig(prediction_horizon=c["N"], dt=c["dt"], **kw)


def test_appendix_b_known_answer():
    """N=6, dt=1/400, p0=(0,0,1), goal=(5,3,2)."""
    tr, info = orc.plan([0, 0, 1.0], [0, 0, 0.0], [5, 3, 2.0], OracleConfig())
    exp = np.array([[2.1897985430479023, 2.1897985430479023, 3.1679005576174233],
                    [3.1679005576174233, 2.776659751789615, 3.3635209605313277],
                    [4.146002572186944, 3.3635209605313277, 3.5591413634452316],
                    [5.124104586756465, 3.95038216927304, 3.351838834438215],
                    [6.102206601325986, 4.537243378014753, 2.6759194172191076],
                    [5.0, 3.0, 2.0]])
    assert np.allclose(tr["positions"], exp, rtol=0, atol=1e-12)
    assert np.allclose(tr["velocities"][0], 0) and np.allclose(tr["velocities"][1:], 9.562040291390419, atol=1e-12)
    assert np.allclose(tr["thrusts"], 14.650554228878104, atol=1e-12)
    assert np.allclose(tr["accelerations"], [0, 0, -0.04296384741459747], atol=1e-12)
    assert np.allclose(tr["attitudes"], [0, 0, -np.pi / 2], atol=1e-12)
    assert np.allclose(tr["body_rates"], 0, atol=1e-12)
    assert (info["nit"], info["nfev"], info["status"]) == (1, 3, 0)


def test_solve_cases(golden_solve):
    data, meta = golden_solve
    for c in meta["cases"]:
        k = c["key"]
        cfg = cfg_for(c, max_iterations=c["maxiter"], convergence_tolerance=c["tol"])
        p0, v0, goal = data[k + "p0"], data[k + "v0"], data[k + "goal"]
        x0 = orc.straight_line_init(p0, v0, goal, cfg)
        assert np.allclose(x0, data[k + "x0"], **TOL), k
        assert np.allclose(orc.bounds(cfg), data[k + "bounds"], **TOL), k
        # every (x, f, g) the reference's callbacks saw during its solve
        ex = data[k + "evals_x"]
        assert np.allclose(orc.objective(ex, goal, cfg), data[k + "evals_f"], rtol=1e-13, atol=1e-9), k
        assert np.allclose(orc.gradient(ex, goal, cfg), data[k + "evals_g"], **TOL), k
        for x, f, g in list(zip(ex, data[k + "evals_f"], data[k + "evals_g"]))[:3]:
            assert np.isclose(orc.objective_loops(x, goal, cfg), f, rtol=1e-14, atol=0), k
            assert np.allclose(orc.gradient_loops(x, goal, cfg), g, rtol=1e-15, atol=0), k
        tr, info = orc.plan(p0, v0, goal, cfg)
        assert (info["nit"], info["nfev"], info["status"]) == (c["nit"], c["nfev"], c["status"]), (k, info)
        assert np.allclose(tr["x"], data[k + "x"], **TOL), k
        for name in ("positions", "velocities", "accelerations", "attitudes", "body_rates", "thrusts",
                     "yaws", "yaw_rates"):
            assert np.allclose(tr[name], data[k + name], rtol=1e-11, atol=1e-9), (k, name)
        # timestamps = time.time() + arange(N)*dt (planner.py:663): the wall-clock offset (1.7e9 s)
        # costs ~2e-7 s of float64 resolution, so only the spacing is comparable
        assert np.allclose(tr["timestamps_rel"], data[k + "timestamps_rel"], rtol=0, atol=1e-6), k


def test_reference_shaped_plan_matches(golden_solve):
    data, meta = golden_solve
    for c in meta["cases"][:8]:
        k = c["key"]
        cfg = cfg_for(c, max_iterations=c["maxiter"], convergence_tolerance=c["tol"])
        sol = orc.plan_reference_shaped(data[k + "p0"], data[k + "v0"], data[k + "goal"], cfg)
        assert np.allclose(sol["positions"], data[k + "positions"], **TOL), k
        assert np.allclose(sol["body_rates"], data[k + "body_rates"], rtol=1e-11, atol=1e-9), k


def test_path_functions(golden_path):
    data, meta = golden_path
    for c in meta["cases"]:
        k = c["key"]
        cfg = cfg_for(c)
        goal = data[k + "goal"] if c["with_goal"] else None
        p0, v0, X = data[k + "p0"], data[k + "v0"], data[k + "X"]
        assert np.allclose(orc.objective(X, goal, cfg), data[k + "f"], rtol=1e-13, atol=1e-9), k
        assert np.allclose(orc.gradient(X, goal, cfg), data[k + "g"], **TOL), k
        assert np.allclose(orc.dynamics_residual(X, p0, v0, cfg), data[k + "dyn"], **TOL), k
        assert np.allclose(orc.physical_constraints(X, cfg), data[k + "phys"], **TOL), k
        assert np.allclose(orc.obstacle_residual(X, data[k + "obs_c"], data[k + "obs_r"], cfg), data[k + "obs"], **TOL), k
        assert np.allclose(orc.straight_line_init(p0, v0, goal, cfg), data[k + "x0_init"], **TOL), k
        assert np.allclose(orc.bounds(cfg), data[k + "bounds"], **TOL), k
        ex = orc.extract_solution_batch(data[k + "Xe"], cfg)
        for name in ("accelerations", "attitudes", "body_rates", "thrusts"):
            assert np.allclose(ex[name], data[k + "ex_" + name], rtol=1e-11, atol=1e-9), (k, name)
        em = orc.emergency_trajectory(p0, 123.0, cfg)
        for name in ("positions", "velocities", "accelerations", "timestamps"):
            assert np.allclose(em[name], data[k + "em_" + name], **TOL), (k, name)


def test_warm_start(golden_path):
    data, _ = golden_path
    cfg = OracleConfig(prediction_horizon=8)
    for plen in (8, 5, 12):
        k = f"w{plen:02d}_"
        prev = {n: data[k + n] for n in ("positions", "velocities", "thrust_vectors")}
        x = orc.warm_start(prev, data[k + "p0"], data[k + "v0"], data[k + "goal"], cfg)
        assert np.allclose(x, data[k + "x0_warm"], **TOL), k


def test_is_plan_valid(golden_path):
    data, meta = golden_path
    for v in meta["valid"]:
        assert orc.is_plan_valid(data[v["key"] + "P"], data[v["key"] + "V"]) == v["valid"], v["tag"]
    assert orc.is_plan_valid(np.zeros((0, 3))) is False


def test_shooting_rollout_is_pinned_by_reference_functions(golden_path):
    """The build's forward rollout is defined by a8 == 0 and costed by a5 (both pinned above);
    its adjoint gradient is checked against central differences of a5 o rollout."""
    rng = np.random.default_rng(5)
    for N, dt in ((6, 1 / 400), (30, 1 / 400), (20, 0.1)):
        cfg = OracleConfig(prediction_horizon=N, dt=dt)
        B = 5
        p0, v0, goal = rng.uniform(-20, 20, (B, 3)), rng.uniform(-5, 5, (B, 3)), rng.uniform(-20, 20, (B, 3))
        T = rng.normal(0, 2, (B, N, 3)) + [0, 0, cfg.hover_thrust]
        P, V = orc.rollout(p0, v0, T, cfg)
        X = orc.pack(P, V, T)
        res = np.stack([orc.dynamics_residual(X[b], p0[b], v0[b], cfg) for b in range(B)])
        assert np.max(np.abs(res)) < 1e-12
        cost, G = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
        assert np.allclose(cost, orc.objective(X, goal, cfg), rtol=1e-14)
        eps = 0.25   # the cost is exactly quadratic in T: central differences are exact, a large eps only removes rounding noise
        for _ in range(12):
            b, k, a = rng.integers(B), rng.integers(N), rng.integers(3)
            Tp, Tm = T.copy(), T.copy()
            Tp[b, k, a] += eps; Tm[b, k, a] -= eps
            fd = (orc.rollout_cost(p0[b], v0[b], goal[b], Tp[b], cfg) - orc.rollout_cost(p0[b], v0[b], goal[b], Tm[b], cfg)) / (2 * eps)
            assert np.isclose(G[b, k, a], fd, rtol=1e-8, atol=1e-6), (N, b, k, a, G[b, k, a], fd)


def test_oracle_reproduces_config1_solves(golden_cfg1):
    """BASELINE.json config 1 (horizon 20, the cloud controller's |v| <= 8 box, 101 goals): the oracle's solve is
    what the reference returned -- counts exactly, decision vectors and trajectory arrays to 1e-10."""
    data, meta = golden_cfg1
    cfg = orc.OracleConfig(prediction_horizon=meta["N"], dt=meta["dt"], max_velocity=meta["max_velocity"],
                           max_acceleration=meta["max_acceleration"], max_iterations=meta["maxiter"], convergence_tolerance=meta["tol"])
    assert np.array_equal(orc.bounds(cfg), data["bounds"])
    for i in range(meta["n"]):
        x, info = orc.solve(data["p0"], data["v0"], data["goals_used"][i], cfg)
        assert (info["nit"], info["nfev"], info["status"]) == tuple(int(v) for v in data["info"][i])
        assert np.max(np.abs(x - data["x"][i])) <= 1e-10
        ex = orc.extract_solution(x, cfg)
        for name in ("accelerations", "attitudes", "body_rates", "thrusts"):
            assert np.max(np.abs(ex[name] - data[name][i])) <= 1e-9, name


def test_objective_ordered_rounds_like_the_reference_loops():
    """oracle.objective_ordered (what solve() hands SciPy) == the reference-shaped per-step loops, BIT FOR BIT."""
    rng = np.random.default_rng(0)
    for N in (1, 2, 6, 7, 20, 30, 50, 64):
        cfg = orc.OracleConfig(prediction_horizon=N)
        for t in range(60):
            x = rng.normal(0, 5, 9 * N)
            goal = rng.uniform(-20, 20, 3) if t % 5 else None
            assert orc.objective_ordered(x, goal, cfg) == orc.objective_loops(x, goal, cfg)


def test_oracle_reproduces_the_reference_on_the_knife_edge_problem():
    """tests/golden/bifurcation_case: the reference ends at (3, 27), f = 2680.18; SciPy on a batched-sum objective at (3, 24), f = 2728.05,
    0.376 m away.  The oracle's solve() (reference-ordered objective) must be on the reference's side, exactly."""
    import json
    data = np.load(os.path.join(GOLDEN, "bifurcation_case.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "bifurcation_case.json")))
    cfg = orc.OracleConfig(prediction_horizon=meta["N"], dt=meta["dt"], **meta["weights"])
    x, info = orc.solve(data["p0"], data["v0"], data["goal"], cfg)
    ref = meta["reference"]
    assert (info["nit"], info["nfev"], info["status"]) == (ref["nit"], ref["nfev"], ref["status"])
    assert np.max(np.abs(x - data["x_reference"])) <= 1e-12 and abs(info["fun"] - ref["fun"]) <= 1e-9
    assert meta["gap_m"] > 0.3 and meta["other_branch"]["nfev"] == 24
