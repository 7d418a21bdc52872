"""GPU suite (`-m gpu`): the HIP kernels of libse3mpc.so on a real MI355X, called through the C ABI
(ctypes) with PyTorch-ROCm tensors, against the oracle and the reference's golden vectors, plus
size-independent properties at BASELINE.json's full sizes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import parity_checks as pc  # noqa: E402
from oracle import se3mpc_oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def gpu_ops():
    import torch
    assert torch.cuda.is_available(), "the gpu suite needs an MI355X"
    from dart_planner_amd.ops import Ops, TorchBackend
    ops = Ops(TorchBackend("cuda:0"))
    assert ops.lib.device_count() >= 1, "no gfx950 device visible to libse3mpc"
    assert os.path.basename(ops.lib.path) == "libse3mpc.so"
    return ops


def harness(ops, dt):
    import torch
    return pc.Harness(ops, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"),
                      lambda a: a.detach().cpu().numpy(), dt)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(1, 5), (6, 70), (20, 257), (30, 1000), (30, 1024), (50, 129), (64, 65), (7, 9), (33, 200), (24, 130), (17, 64), (32, 100)])
def test_lane_kernels(gpu_ops, dt, N, B):
    pc.check_lane_kernels(harness(gpu_ops, dt), N, B, seed=N, variants=(0, 1, 2, 3, 4, 5, 6))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_transpose(gpu_ops, dt):
    pc.check_transpose(harness(gpu_ops, dt))
    pc.check_transpose(harness(gpu_ops, dt), shapes=((270, 100003), (100003, 270), (450, 8192)))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("rows,B", [(90, 8192), (150, 200001), (3, 64)])
def test_population_sums(gpu_ops, dt, rows, B):
    pc.check_population_sums(harness(gpu_ops, dt), rows=rows, B=B, seed=rows)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 300), (20, 257), (30, 1000), (50, 129), (7, 9), (24, 130), (64, 65), (1, 5)])
def test_rollout_iterate(gpu_ops, dt, N, B):
    bitwise = pc.check_rollout_iterate(harness(gpu_ops, dt), N, B, seed=N, iters=6)
    print(f"N={N} {np.dtype(dt).name}: one launch vs rollout_cost_grad + projected_step chain bit-identical: {bitwise}")


def test_rollout_iterate_full_size(gpu_ops):
    """The metric's own batch (N = 30, B = 8192): 16 iterations in one launch == 16 one-iteration launches bit for bit, a 256-trajectory
    sample against the host-chained oracle, every trajectory inside the box and no worse than where it started."""
    import torch
    from dart_planner_amd.capi import Params
    ops = gpu_ops
    dev = ops.be.device
    N, B, K, step = 30, 8192, 16, 0.9
    prm = Params.reference_defaults(horizon=N)
    g = torch.Generator(device=dev); g.manual_seed(1)
    p0 = torch.rand(3, B, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(3, B, device=dev, generator=g) * 10 - 5
    goal = torch.rand(3, B, device=dev, generator=g) * 40 - 20
    T = torch.randn(3 * N, B, device=dev, generator=g) * 2
    T[2::3] += 14.715
    out = ops.rollout_iterate(prm, p0, v0, goal, T, K, step, want_first_cost=True)
    Tc = T
    for _ in range(K):
        Tc = ops.rollout_iterate(prm, p0, v0, goal, Tc, 1, step, want_grad=False)["T"]
    assert torch.equal(Tc, out["T"])
    assert bool((out["cost"] <= out["cost_first"] * (1 + 1e-6)).all())
    cfg = orc.OracleConfig(prediction_horizon=N)
    pick = np.random.default_rng(0).choice(B, 256, replace=False)
    hst = lambda a: a[:, pick].double().cpu().numpy().T
    Tr, cr, gr, _ = pc.oracle_iterate(hst(p0), hst(v0), hst(goal), hst(T).reshape(256, N, 3), cfg, K, step)
    assert np.max(np.abs(hst(out["T"]).reshape(256, N, 3) - Tr)) <= 2e-4
    assert np.max(np.abs(out["cost"][pick].cpu().numpy() - cr) / cr) <= 5e-5


def test_keys_with_nonfinite_costs(gpu_ops):
    pc.check_key_nonfinite(harness(gpu_ops, np.float32))


def test_lane_kernels_other_dt(gpu_ops):
    pc.check_lane_kernels(harness(gpu_ops, np.float64), 20, 300, seed=2, dt=0.05)
    pc.check_lane_kernels(harness(gpu_ops, np.float32), 20, 300, seed=2, dt=0.1)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_reproduces_every_reference_solve(gpu_ops, golden_solve, dt):
    data, meta = golden_solve
    worst = pc.check_solver_golden(harness(gpu_ops, dt), data, meta)
    print(f"worst position error vs the reference ({np.dtype(dt).name}): {worst:.3e} m")
    assert worst <= (1e-4 if dt == np.float32 else 1e-9)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_published_cauchy_search(gpu_ops, golden_solve, golden_cfg1, dt):
    """se3mpc_set_solver_variant(1): the published sequential Cauchy search everywhere.  Every golden solve again, f64 thrust block
    held to 1e-9 (no carve-out), config 1's set, and a batch against SciPy."""
    data, meta = golden_solve
    gpu_ops.lib.set_solver_variant(1)
    try:
        worst = pc.check_solver_golden(harness(gpu_ops, dt), data, meta, thrust_tol=1e-9 if dt == np.float64 else None)
        assert worst <= (1e-4 if dt == np.float32 else 1e-9)
        pc.check_solver_cfg1(harness(gpu_ops, dt), *golden_cfg1)
        w, mism = pc.check_solver_vs_oracle(harness(gpu_ops, dt), 30, 128, seed=2)
        assert w <= (1e-4 if dt == np.float32 else 1e-9) and mism <= pc.mismatch_budget(128, dt)
    finally:
        gpu_ops.lib.set_solver_variant(0)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_config1_exact_problem_set(gpu_ops, golden_cfg1, dt):
    """BASELINE.json config 1: all 101 reference solves of the horizon-20 / |v| <= 8 configuration in one launch."""
    data, meta = golden_cfg1
    worst = pc.check_solver_cfg1(harness(gpu_ops, dt), data, meta)
    print(f"config 1, {np.dtype(dt).name}: worst position error vs the reference {worst:.3e} m")
    assert worst <= (1e-4 if dt == np.float32 else 1e-9)


def test_planner_mirror_config1_sequence(gpu_ops, golden_cfg1):
    """The same set the way the reference's test runs it: ONE planner object (the cloud controller's configuration),
    plan_trajectory goal after goal, goal hysteresis in force -> the reference's trajectories."""
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
    from dart_planner_amd.common.types import DroneState
    data, meta = golden_cfg1
    pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=20, dt=0.1, max_velocity=8.0, max_acceleration=4.0, position_weight=100.0,
                                    velocity_weight=10.0, obstacle_weight=1000.0, safety_margin=1.5))
    st = DroneState(timestamp=0.0, position=data["p0"].copy(), velocity=data["v0"].copy())
    for i, g in enumerate(data["goals_asked"]):
        tr = pl.plan_trajectory(st, g.copy())
        assert np.array_equal(pl.goal_position, data["goals_used"][i])
        assert (pl.last_result["nit"], pl.last_result["nfev"], pl.last_result["status"]) == tuple(int(v) for v in data["info"][i])
        assert np.max(np.abs(np.asarray(tr.positions) - data["positions"][i])) <= 1e-9
        assert np.max(np.abs(np.asarray(tr.velocities) - data["velocities"][i])) <= 1e-9
        assert np.max(np.abs(np.asarray(tr.thrusts) - data["thrusts"][i])) <= 1e-7


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N", [6, 20, 30, 50, 64])
def test_solver_extraction_and_cold_start(gpu_ops, dt, N):
    pc.check_solver_extraction(harness(gpu_ops, dt), N, 37)


@pytest.mark.parametrize("N,B", [(6, 256), (20, 128), (30, 256), (50, 96), (64, 48), (13, 100), (33, 64), (1, 40), (2, 40)])
def test_batched_solve_matches_scipy_problem_by_problem(gpu_ops, N, B):
    """Config 1/2 of BASELINE.json: random (state, goal) pairs, reference options."""
    for dt, tol in ((np.float64, 1e-9), (np.float32, 1e-4)):
        worst, mismatch = pc.check_solver_vs_oracle(harness(gpu_ops, dt), N, B, seed=N)
        print(f"N={N} B={B} {np.dtype(dt).name}: max position error {worst:.3e} m, iteration-count mismatches {mismatch:.4f}")
        assert worst <= tol
        assert mismatch <= pc.mismatch_budget(B, dt)


@pytest.mark.parametrize("N,B", [(20, 64), (30, 96), (50, 64), (64, 48)])
def test_batched_solve_published_cauchy_search_matches_scipy(gpu_ops, N, B):
    """The published sequential Cauchy search (se3mpc_set_solver_variant(1)) at the three register-slot counts (J = 3, 6, 9), both dtypes,
    problem by problem against SciPy -- the J = 9 float32 kernel once returned a wrong Cauchy point under this variant only."""
    gpu_ops.lib.set_solver_variant(1)
    try:
        for dt, tol in ((np.float64, 1e-9), (np.float32, 1e-4)):
            worst, mismatch = pc.check_solver_vs_oracle(harness(gpu_ops, dt), N, B, seed=100 + N)
            assert worst <= tol and mismatch <= pc.mismatch_budget(B, dt), (N, np.dtype(dt).name, worst, mismatch)
    finally:
        gpu_ops.lib.set_solver_variant(0)


@pytest.mark.parametrize("group", [8, 16, 32, 64])
def test_solver_group_sizes(gpu_ops, golden_solve, group):
    """The packed solver at every lanes-per-problem: golden solves whose horizon fits the group, and batches of random problems (several
    wavefronts of 64 / group co-resident problems with different iteration counts, a ragged last one) against SciPy one by one."""
    data, meta = golden_solve
    keys = {c["key"] for c in meta["cases"] if c["N"] <= group}
    for dt, tol in ((np.float64, 1e-9), (np.float32, 1e-4)):
        assert pc.check_solver_golden(harness(gpu_ops, dt), data, meta, keys=keys, group=group) <= tol
        for N in {8: (6, 8, 3), 16: (13, 16, 6), 32: (30, 20, 6), 64: (40, 64, 30)}[group]:
            B = 5 * (64 // group) + 3
            worst, mism = pc.check_solver_vs_oracle(harness(gpu_ops, dt), N, B, seed=group + N, group=group)
            assert worst <= tol and mism <= pc.mismatch_budget(B, dt), (group, N, np.dtype(dt).name, worst, mism)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_on_the_knife_edge_problem(gpu_ops, dt):
    """tests/golden/bifurcation_case: the fuzz sweeps' one 0.376 m "mismatch" -- a problem on which the last bit of an objective value
    picks between two outcomes.  The result must be one of the two (the reference's own, or the other branch); printed: which."""
    for group in (None, 64):
        branch, err = pc.check_solver_bifurcation_case(harness(gpu_ops, dt), group=group)
        print(f"knife-edge problem, {np.dtype(dt).name}, group {group}: {branch}, {err:.2e} m")


def test_batched_solve_tight_tolerances(gpu_ops):
    """Many L-BFGS-B iterations per problem (memory fills up, line searches fail on the reference's
    inconsistent gradient): f64 must follow SciPy exactly."""
    worst, mismatch = pc.check_solver_vs_oracle(harness(gpu_ops, np.float64), 20, 48, seed=5, pgtol=1e-9, ftol=1e-8,
                                                max_iterations=40)
    assert mismatch == 0.0 and worst <= 1e-8


# ------------------------------------------------------------------ full sizes: properties
def test_full_size_rollout_properties(gpu_ops):
    """B = 65536, N = 30 (config 4's per-node batch) and B = 8192, N = 50 with K = 16 spheres (config 3):
    too big for the per-problem oracle, so check invariants: (1) the rolled-out states zero the
    reference's dynamics residual; (2) cost(rollout) == objective(packed rollout); (3) the adjoint
    gradient satisfies the directional-derivative identity of a quadratic; (4) fused key == argmin;
    (5) obstacle min/violation reductions agree with the materialised residuals."""
    import torch
    from dart_planner_amd.capi import Params
    ops = gpu_ops
    dev = ops.be.device
    for N, B in ((30, 65536), (50, 8192)):
        prm = Params.reference_defaults(horizon=N)
        g = torch.Generator(device=dev); g.manual_seed(N)
        p0 = torch.rand(3, B, device=dev, generator=g) * 40 - 20
        v0 = torch.rand(3, B, device=dev, generator=g) * 10 - 5
        goal = torch.rand(3, B, device=dev, generator=g) * 40 - 20
        T = torch.randn(3 * N, B, device=dev, generator=g) * 2
        T[2::3] += 14.715
        key = torch.zeros(1, dtype=torch.int64, device=dev)
        cost, gT, P, V = ops.rollout_cost_grad(prm, p0, v0, goal, T, want_states=True, key=key)
        X = torch.cat([P, V, T], dim=0).contiguous()
        R = ops.dynamics_residual(prm, X, p0, v0)
        assert float(R.abs().max()) <= 2e-4                                        # (1) f32 roundoff of |P| ~ 1e2
        f, _ = ops.cost_grad(prm, X, goal, want_grad=False)
        assert float(((f - cost).abs() / cost).max()) <= 2e-6                       # (2)
        dT = torch.randn(3 * N, B, device=dev, generator=g)
        # (3) the cost is quadratic in T, so central differences are exact: evaluate them with the f64
        # entry point (float32 costs of ~1e6 cannot resolve a difference of ~1e2)
        d64 = lambda a: a.double().contiguous()
        cp, _, _, _ = ops.rollout_cost_grad(prm, d64(p0), d64(v0), d64(goal), d64(T) + d64(dT), want_grad=False)
        cm, _, _, _ = ops.rollout_cost_grad(prm, d64(p0), d64(v0), d64(goal), d64(T) - d64(dT), want_grad=False)
        dd = (gT.double() * dT.double()).sum(0)
        fd = (cp - cm) / 2
        assert float(((dd - fd).abs() / (gT.double() * dT.double()).abs().sum(0)).max()) <= 2e-5
        idx, kc = ops.decode_key(key)
        assert idx == int(torch.argmin(cost)) and kc == float(cost.min())           # (4)
        sph = torch.cat([torch.round(torch.rand(16, 3, device=dev, generator=g) * 30) / 2, torch.ones(16, 1, device=dev)], 1)
        Cm, cmin, viol = ops.obstacle_residual(prm, X, sph)
        assert torch.equal(Cm.min(0).values, cmin)                                  # (5)
        assert float((viol - torch.clamp(-Cm, min=0).sum(0)).abs().max()) <= 1e-3 * max(1.0, float(viol.max()))
        # (6) the FUSED kernel at this size (config 3 is N = 50, B = 8192, K = 16): every trajectory against the
        # unfused pair above (same cost / gradient; min and violation of the materialised residuals of the same
        # rolled-out states), and a 256-trajectory sample against the oracle
        wk = torch.zeros((B + 63) // 64, dtype=torch.int64, device=dev)
        fc, fg, fmin, fviol = ops.rollout_obstacles(prm, p0, v0, goal, T, sph, wave_keys=wk, index_base=0)
        assert float(((fc - cost).abs() / cost).max()) <= 1e-6 and float((fg - gT).abs().max()) <= 1e-5 * float(gT.abs().max())
        assert float((fmin - cmin).abs().max()) <= 2e-4 * max(1.0, float(cmin.abs().max()))
        assert float((fviol - viol).abs().max()) <= 1e-3 * max(1.0, float(viol.max()))
        fk = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.reduce_keys(wk.view(1, -1), fk)
        assert ops.decode_key(fk)[0] == int(torch.argmin(fc))
        cfg = orc.OracleConfig(prediction_horizon=N)
        pick = np.random.default_rng(N).choice(B, 256, replace=False)
        hst = lambda a: a[:, pick].double().cpu().numpy().T
        Tn = hst(T).reshape(256, N, 3)
        c_ref, g_ref = orc.rollout_cost_grad(hst(p0), hst(v0), hst(goal), Tn, cfg)
        P_ref, V_ref = orc.rollout(hst(p0), hst(v0), Tn, cfg)
        sp = sph.double().cpu().numpy()
        C_ref = orc.obstacle_residual(orc.pack(P_ref, V_ref, Tn), sp[:, :3], sp[:, 3], cfg)
        assert np.max(np.abs(fc[pick].cpu().numpy() - c_ref) / c_ref) <= 5e-6
        assert np.max(np.abs(hst(fg).reshape(256, N, 3) - g_ref)) <= 5e-6 * np.max(np.abs(g_ref))
        assert np.max(np.abs(fmin[pick].cpu().numpy() - C_ref.min(1))) <= 2e-5 * max(1.0, np.max(np.abs(C_ref)))
        assert np.max(np.abs(fviol[pick].cpu().numpy() - np.maximum(0, -C_ref).sum(1))) <= 4e-5 * max(1.0, np.max(np.abs(C_ref)))


def test_monte_carlo_restarts_f64_vs_f32(gpu_ops):
    """Config 5 shape (initial states x restarts), reduced to 256 x 32 for test time: restart 0 is the
    reference cold start, the others perturb the thrust block; the f32 solve tracks the f64 solve
    and restart 0 equals the plain cold-start solve."""
    import torch
    from dart_planner_amd.capi import Params
    ops = gpu_ops
    dev = ops.be.device
    N, S, Rr = 20, 256, 32
    prm = Params.reference_defaults(horizon=N)
    cfg = orc.OracleConfig(prediction_horizon=N)
    rng = np.random.default_rng(9)
    p0, v0, goal, _ = pc.random_batch(rng, S, N)
    x0 = orc.straight_line_init(p0, v0, goal, cfg)                                   # (S, 9N)
    X0 = np.repeat(x0[:, None, :], Rr, axis=1)
    X0[:, 1:, 6 * N:] += rng.normal(0, 1.0, (S, Rr - 1, 3 * N))
    rep = lambda a: np.repeat(a[:, None, :], Rr, axis=1).reshape(S * Rr, 3)
    outs = {}
    for dt in (np.float64, np.float32):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)
        o = ops.solve(prm, t(rep(p0)), t(rep(v0)), t(rep(goal)), x0=t(X0.reshape(S * Rr, 9 * N)), want_trajectory=False)
        outs[dt] = (o["x"].cpu().numpy().astype(float), ops.info_to_host(o["info"]))
    x64, i64 = outs[np.float64]; x32, i32 = outs[np.float32]
    same = (i64["nit"] == i32["nit"]) & (i64["nfev"] == i32["nfev"])
    assert same.mean() >= 0.98
    assert np.max(np.abs(x64[same][:, :3 * N] - x32[same][:, :3 * N])) <= 1e-4
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cold = ops.solve(prm, t64(p0), t64(v0), t64(goal), want_trajectory=False)["x"].cpu().numpy()
    assert np.max(np.abs(cold - x64.reshape(S, Rr, -1)[:, 0])) <= 1e-9


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_spheres_from_grid(gpu_ops, golden_mapper, dt):
    pc.check_spheres_from_grid(harness(gpu_ops, dt), *golden_mapper)


def test_planner_obstacles_from_mapper_grid(gpu_ops, golden_mapper):
    """The cloud loop's mapper -> planner hand-off (main_improved_threelayer.py:381-398) through the mirror."""
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCPlanner
    from dart_planner_amd.common.types import DroneState, Trajectory
    data, meta = golden_mapper
    c = meta["cases"][0]
    pl = SE3MPCPlanner()
    nc = c["num_cells"]
    k = pl.refresh_obstacles_from_grid(data[c["key"] + "grid"].reshape(nc, nc, nc, 3), data[c["key"] + "occ"].reshape(nc, nc, nc))
    exp = data[c["key"] + "spheres"]
    assert k == len(exp) == len(pl.obstacles)
    assert all(np.array_equal(o[0], e[:3]) and o[1] == 1.0 for o, e in zip(pl.obstacles, exp))
    tr = pl.plan_trajectory(DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 2.0])), np.array([10.0, 0.0, 5.0]))
    clr = pl.obstacle_clearance(tr)
    P = np.asarray(tr.positions)
    ref = min(np.sum((P[i] - o[0]) ** 2) - (o[1] + 1.5) ** 2 for i in range(len(P)) for o in pl.obstacles)
    assert abs(clr["min_residual"] - ref) <= 1e-9


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_edge_inputs(gpu_ops, dt):
    pc.check_solver_edge_inputs(harness(gpu_ops, dt))


def test_entry_points_are_graph_capturable(gpu_ops):
    """The library never allocates or synchronises: a solve and a rollout captured into a hipGraph replay
    to the same results as the eager calls."""
    import torch
    from dart_planner_amd.capi import Params
    ops = gpu_ops
    dev = ops.be.device
    N, B = 20, 300
    prm = Params.reference_defaults(horizon=N)
    g = torch.Generator(device=dev); g.manual_seed(3)
    p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
    goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    eager = ops.solve(prm, p0, v0, goal)
    xe, ie = eager["x"].clone(), ops.info_to_host(eager["info"]).copy()
    buf = ops.be.empty((ops.packed_size(B, N, "f32"),), "u8")
    inputs = torch.stack([p0, v0, goal]).contiguous()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ops.solve_packed(prm, inputs, out=buf)
    torch.cuda.current_stream().wait_stream(side)
    buf.zero_()
    graph.replay()
    torch.cuda.synchronize()
    res = ops.unpack_solution(buf.cpu().numpy(), B, N, "f32")
    assert np.array_equal(res["x"], xe.cpu().numpy().astype(np.float64))
    assert np.array_equal(res["info"]["nfev"], ie["nfev"]) and np.array_equal(res["info"]["fun"], ie["fun"])


def test_monte_carlo_full_size(gpu_ops):
    """BASELINE.json config 5 at full size: 4096 initial states x 256 restarts = 1 048 576 solves (f32) in one
    launch.  Checked through properties: restart 0 of every state equals the plain cold-start solve; every
    solve ends with a SciPy status; the best restart's objective is <= the cold start's; a 4096-solve f64
    sample agrees with f32 to the north_star tolerance."""
    import torch
    from dart_planner_amd.capi import Params
    ops = gpu_ops
    dev = ops.be.device
    N, S, Rr = 6, 4096, 256
    prm = Params.reference_defaults(horizon=N)
    g = torch.Generator(device=dev); g.manual_seed(7)
    p0 = torch.rand(S, 3, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(S, 3, device=dev, generator=g) * 10 - 5
    goal = torch.rand(S, 3, device=dev, generator=g) * 40 - 20
    cold = ops.solve(prm, p0, v0, goal, want_trajectory=False)
    lane = lambda a: a.t().contiguous()
    x0 = ops.transpose(ops.init(prm, lane(p0), lane(v0), lane(goal)))                         # (S, 9N) cold starts
    X0 = x0[:, None, :].repeat(1, Rr, 1)
    X0[:, 1:, 6 * N:] += torch.randn(S, Rr - 1, 3 * N, device=dev, generator=g)
    rep = lambda a: a[:, None, :].repeat(1, Rr, 1).reshape(-1, 3).contiguous()
    out = ops.solve(prm, rep(p0), rep(v0), rep(goal), x0=X0.reshape(S * Rr, 9 * N).contiguous(), want_trajectory=False)
    info = ops.info_to_host(out["info"]).reshape(S, Rr)
    assert set(np.unique(info["status"])) <= {0, 1, 2}
    x = out["x"].reshape(S, Rr, 9 * N)
    assert float((x[:, 0] - cold["x"]).abs().max()) <= 1e-4
    fun = info["fun"]
    assert np.all(fun.min(axis=1) <= fun[:, 0] + 1e-9)
    sub = slice(0, 16)
    d = lambda a: a.double().contiguous()
    o64 = ops.solve(prm, d(rep(p0[sub])), d(rep(v0[sub])), d(rep(goal[sub])), x0=d(X0[sub].reshape(-1, 9 * N)), want_trajectory=False)
    i64 = ops.info_to_host(o64["info"])
    same = (i64["nit"] == info[sub].reshape(-1)["nit"]) & (i64["nfev"] == info[sub].reshape(-1)["nfev"])
    assert same.mean() >= 1 - pc.mismatch_budget(len(same), np.float32)
    err = (o64["x"] - x[sub].reshape(-1, 9 * N).double()).abs()[torch.from_numpy(same).to(dev)][:, :3 * N].max()
    assert float(err) <= 1e-4
    # 512 PERTURBED restarts (never restart 0) against SciPy from the same x0, problem by problem: the f64 solve must
    # follow SciPy exactly (nit, nfev, status; positions <= 1e-9 m), the f32 solve of the full-size launch within
    # the north_star tolerance wherever it took the same path
    cfg = orc.OracleConfig(prediction_horizon=N)
    rng = np.random.default_rng(3)
    si, ri = rng.integers(0, S, 512), rng.integers(1, Rr, 512)
    flat = torch.from_numpy(si * Rr + ri).to(dev)
    X0s = X0.reshape(S * Rr, 9 * N)[flat]
    o64 = ops.solve(prm, d(p0[si]), d(v0[si]), d(goal[si]), x0=d(X0s), want_trajectory=False)
    i64 = ops.info_to_host(o64["info"])
    x64 = o64["x"].cpu().numpy()
    x32 = x.reshape(S * Rr, 9 * N)[flat].cpu().numpy().astype(float)
    i32 = info.reshape(-1)[si * Rr + ri]
    hp0, hv0, hg, hx0 = (a.double().cpu().numpy() for a in (p0[si], v0[si], goal[si], X0s))
    worst64 = worst32 = 0.0
    same32 = 0
    for j in range(512):
        xr, ir = orc.solve(hp0[j], hv0[j], hg[j], cfg, x0=hx0[j])
        assert (int(i64["nit"][j]), int(i64["nfev"][j]), int(i64["status"][j])) == (ir["nit"], ir["nfev"], ir["status"]), j
        worst64 = max(worst64, float(np.max(np.abs(x64[j, :3 * N] - xr[:3 * N]))))
        e32 = float(np.max(np.abs(x32[j, :3 * N] - xr[:3 * N])))
        if (int(i32["nit"][j]), int(i32["nfev"][j]), int(i32["status"][j])) == (ir["nit"], ir["nfev"], ir["status"]):
            same32 += 1
            worst32 = max(worst32, e32)
        elif e32 > 1e-4:
            # unconditional: a float32 solve that counts differently must still end at SciPy's point or at an objective no worse
            assert float(i32["fun"][j]) <= ir["fun"] * (1 + 1e-6) + 1e-12, (j, e32, float(i32["fun"][j]), ir["fun"])
    print(f"config 5, 512 perturbed restarts vs SciPy: f64 {worst64:.2e} m, f32 {worst32:.2e} m ({same32}/512 on SciPy's path)")
    assert worst64 <= 1e-9 and worst32 <= 1e-4 and same32 >= (1 - pc.mismatch_budget(512, np.float32)) * 512


def test_wave_ops_selftest(gpu_ops):
    """The DPP reductions behind every dot product of the solver, on known data."""
    import torch
    ops = gpu_ops
    dev = ops.be.device
    # cost = lane index pattern: the fused key must find the minimum wherever it sits in the wavefront
    from dart_planner_amd.capi import Params
    prm = Params.reference_defaults(horizon=6)
    for pos in (0, 1, 15, 16, 31, 32, 47, 48, 62, 63):
        B = 64
        p0 = torch.zeros(3, B, device=dev); v0 = torch.zeros(3, B, device=dev)
        goal = torch.full((3, B), 50.0, device=dev)
        goal[:, pos] = 0.0                                    # trajectory `pos` is already at its goal: smallest cost
        T = torch.zeros(18, B, device=dev); T[2::3] = 14.715
        key = torch.full((1,), -1, dtype=torch.int64, device=dev)
        cost, *_ = ops.rollout_cost_grad(prm, p0, v0, goal, T, want_grad=False, key=key)
        assert ops.decode_key(key)[0] == pos == int(torch.argmin(cost))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 300), (20, 257), (30, 1000), (50, 129), (7, 9), (24, 130), (64, 65), (1, 5)])
def test_rollout_iterate_obstacles(gpu_ops, dt, N, B):
    """The obstacle-aware iteration loop (running cost + obstacle_weight * sum max(0, -c_kj)^2; the build's extension) at the horizons the
    plain loop is checked at: host-chained oracle with a closed-form penalty gradient, K-in-one == K x 1 bit for bit, narrow == wide workgroup shape,
    no sphere in reach == the plain loop, multi-batch == batch by batch."""
    worst_pen = pc.check_rollout_iterate_obstacles(harness(gpu_ops, dt), N, B, seed=N, iters=5, K=min(5 + N // 4, 16))
    print(f"N={N} {np.dtype(dt).name}: largest penalty in the batch {worst_pen:.3g}")


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("K", [17, 28, 40])
def test_rollout_iterate_obstacles_table_sizes(gpu_ops, dt, K):
    """Sphere tables past the named 16: the helper wavefronts keep 8 / 16 / 24 / 32 spheres in registers (f64: 8 / 16), longer tables are
    swept from LDS by every wavefront; same checks, among them narrow == wide workgroup shape bit for bit."""
    pc.check_rollout_iterate_obstacles(harness(gpu_ops, dt), 20, 150, seed=K, iters=4, K=K)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 70), (30, 8192), (1, 1), (50, 1000), (64, 65)])
def test_shooting_finish(gpu_ops, dt, N, B):
    """The tail of a shooting-form plan in one launch (fold keys, roll the winner out, extract, penalty) against the oracle."""
    pc.check_shooting_finish(harness(gpu_ops, dt), N, B, seed=N)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 200), (30, 70001), (50, 8192), (20, 33)])
def test_iteration_loop_keys(gpu_ops, dt, N, B):
    """The argmin keys of both iteration loops against np.argmin of the launch's own costs, slot by slot: plain loop, obstacle-aware loop narrow (plain
    stores) and wide (two workgroups per slot through atomicMin on slots the launcher presets); ragged batches, an index base."""
    pc.check_iterate_keys(harness(gpu_ops, dt), N, B, seed=N, index_base=0 if B != 33 else 5000)


def test_rollout_iterate_obstacles_full_size(gpu_ops):
    """BASELINE config 3 inside the loop: horizon 50, 8192 trajectories, 16 spheres, 16 iterations in ONE launch == 16 one-iteration
    launches bit for bit; a 128-trajectory sample against the host-chained oracle; the penalty never grows along the descent of a
    trajectory that starts inside a margin (weights chosen so that the step is a descent step)."""
    import torch
    N, B, K, iters, step, w = 50, 8192, 16, 16, 2e-4, 200.0          # a step inside the contraction regime: 1e-3 amplifies an input perturbation 3e4 x in 16 iterations (oracle, f64)
    prm = pc.Params.reference_defaults(horizon=N, dt=0.05)
    cfg = pc.oracle_cfg(prm)
    rng = np.random.default_rng(2)
    p0, v0, goal, T = pc.random_batch(rng, B, N, spread=2.0)
    p0 *= 0.15; goal *= 0.3
    sph = np.concatenate([np.round(rng.uniform(-4, 4, (K, 3)) * 2) / 2, np.full((K, 1), 1.0)], axis=1)     # mapper-style: radius 1.0, 0.5 m grid
    h = harness(gpu_ops, np.float32)
    lane = lambda a: h.lane(a, B)
    dsph = h.to_dev(sph.astype(np.float32))
    run = lambda Tin, it, **kw: gpu_ops.rollout_iterate(prm, lane(p0), lane(v0), lane(goal), Tin, it, step, spheres=dsph, obstacle_weight=w, **kw)
    lT = lane(T)
    out = run(lT, iters, want_first_cost=True)
    Tc = lT
    for _ in range(iters):
        Tc = run(Tc, 1, want_grad=False)["T"]
    assert torch.equal(Tc, out["T"])
    r = lambda a: np.asarray(a).astype(np.float32).astype(float)
    sub = slice(0, 128)
    Tr, cr, gr, c0r, penr = pc.oracle_iterate_obstacles(r(p0[sub]), r(v0[sub]), r(goal[sub]), r(T[sub]), r(sph), cfg, iters, step, w)
    Tk = out["T"].cpu().numpy().T.reshape(B, N, 3)[sub].astype(float)
    pc.vec_close(Tk, Tr, 5e-6 * 200, "8192 x 16 iterations vs oracle")
    assert np.max(np.abs(out["penalty"].cpu().numpy()[sub] - penr)) <= 1e-3 * max(1.0, penr.max())
    pen0 = run(lT, 0)["penalty"].cpu().numpy()
    pen1 = out["penalty"].cpu().numpy()
    inside = pen0 > 0
    print(f"config 3 inside the loop: {int(inside.sum())} of {B} trajectories start inside a margin; mean penalty {pen0[inside].mean():.3g} -> {pen1[inside].mean():.3g}")
    assert inside.sum() > 100 and pen1[inside].mean() < 0.95 * pen0[inside].mean()
    assert np.all(out["cost"].cpu().numpy() <= out["cost_first"].cpu().numpy() * (1 + 1e-5))


def test_shooting_plan_avoids_the_mappers_obstacle(gpu_ops):
    """plan_shooting with the planner's obstacle list (filled from the device voxel map the way the reference's cloud loop does it,
    cloud/main_improved_threelayer.py:381-398) against the obstacle-blind plan: the blind plan flies through the obstacle and is rejected by
    the mapper's is_trajectory_safe (mapper.py:195-219); the obstacle-aware one is accepted.  dt = 0.1 s (a 10 Hz timing manager) so that a
    30-step plan covers 3 s."""
    from dart_planner_amd.common.timing_alignment import TimingConfig, get_timing_manager, reset_timing_manager
    from dart_planner_amd.common.types import DroneState
    from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
    reset_timing_manager()
    try:
        get_timing_manager(TimingConfig(control_frequency=10.0))
        pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), device="cuda:0")
        assert abs(pl.se3_config.dt - 0.1) < 1e-12
        mapper = ExplicitGeometricMapper(resolution=0.5, max_range=20.0, ops=gpu_ops)
        mapper.add_obstacle(np.array([3.0, 0.0, 2.0]), 1.0)
        st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 2.0]), velocity=np.zeros(3))
        goal = np.array([8.0, 0.5, 2.0])                       # 5 m behind the obstacle's centre: outside every sphere's margin (1 m voxels' spheres + 1.5 m)
        spheres = mapper.local_obstacle_spheres(st.position, 20.0, 0.6, 20, 1.0)
        assert len(spheres) >= 1
        pl.clear_obstacles()
        for c in spheres:
            pl.add_obstacle(c[:3], float(c[3]))
        kw = dict(n_samples=2048, iters=40, step=1e-3, sigma=2.0, seed=0, precision="f32")
        blind = pl.plan_shooting(st, goal, obstacles=False, **kw)
        assert "penalty" not in pl.last_result
        aware = pl.plan_shooting(st, goal, **kw)                       # default: the planner's own obstacle list
        res = dict(pl.last_result)
        safe_blind, first_blind = mapper.is_trajectory_safe(blind.positions, safety_margin=1.0)
        safe_aware, _ = mapper.is_trajectory_safe(aware.positions, safety_margin=1.0)
        dmin = lambda tr: float(np.min(np.linalg.norm(np.asarray(tr.positions) - np.array([3.0, 0.0, 2.0]), axis=1)))
        print(f"blind plan: safe={safe_blind} (first unsafe step {first_blind}), closest approach {dmin(blind):.2f} m; "
              f"obstacle-aware plan: safe={safe_aware}, closest approach {dmin(aware):.2f} m, remaining penalty {res['penalty']:.3g}, "
              f"{len(spheres)} spheres from the map")
        assert not safe_blind and safe_aware
        assert dmin(aware) > dmin(blind) + 1.0
        assert np.linalg.norm(np.asarray(aware.positions)[-1] - goal) < 2.0          # it still gets there
        # the eager (multi-rank) path gives the same plan as the captured one
        from dart_planner_amd.distributed import sharded_shooting_plan
        sph = pl._obstacle_table(None)
        e = sharded_shooting_plan(gpu_ops, pl._params(), st.position, st.velocity, goal, kw["n_samples"], kw["iters"], kw["step"], kw["sigma"], kw["seed"],
                                  "f32", spheres=sph, obstacle_weight=pl.se3_config.obstacle_weight)
        assert e["sample"] == res["sample"] and np.array_equal(e["T"], res["T"])
    finally:
        reset_timing_manager()


def test_planner_shooting_plan_shapes(gpu_ops):
    """The captured plan_shooting over odd sample counts (1, not a multiple of 64, one more than a multiple), zero iterations, both precisions and
    three horizons: the same winner and thrust sequence as the eager chain of the same launches; at most four captured graphs are kept."""
    from dart_planner_amd.common.types import DroneState
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
    from dart_planner_amd.distributed import sharded_shooting_plan
    st = DroneState(timestamp=0.0, position=np.array([0.3, -0.2, 1.0]), velocity=np.array([0.1, 0.0, -0.2]))
    goal = np.array([2.0, 1.0, 2.5])
    for N in (6, 30, 50):
        pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=N), precision="f64")
        pl._ops = gpu_ops
        for S, K, prec in ((1, 0, "f32"), (100, 3, "f32"), (4097, 5, "f64"), (64, 1, "f64"), (65, 2, "f32")):
            tr = pl.plan_shooting(st, goal, n_samples=S, iters=K, step=0.5, sigma=1.0, seed=2, precision=prec)
            e = sharded_shooting_plan(gpu_ops, pl._params(), st.position, st.velocity, goal, S, K, 0.5, 1.0, 2, prec)
            assert e["sample"] == pl.last_result["sample"] and np.array_equal(e["T"], pl.last_result["T"]), (N, S, K, prec)
            assert np.all(np.isfinite(tr.positions)) and tr.positions.shape == (N, 3)
        assert len(pl._shooting_graphs) == 4


def test_wide_lane_kernels_equal_the_dword_forms(gpu_ops):
    """16 bytes per lane (four trajectories) for the write-heavy float32 streams == the dword kernels bit for bit, forced at small sizes and by
    default at a saturating batch."""
    pc.check_wide_kernels(harness(gpu_ops, np.float32), N=30, B=4096, seed=5, K=16)
    pc.check_wide_kernels(harness(gpu_ops, np.float32), N=50, B=1028, seed=6, K=3)


def test_planner_shooting_plan(gpu_ops):
    """SE3MPCPlanner.plan_shooting: 8192 thrust samples x 16 iterations in one launch, argmin, rollout and extraction of the winner.  The
    winner's thrust sequence is what the host-chained oracle produces from the same sample, the Trajectory satisfies the reference's dynamics
    constraints (the rollout is their zero set), and it is at least as cheap as the hover sample after the same iterations."""
    from dart_planner_amd.common.types import DroneState
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
    N, S, K = 30, 8192, 16
    pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=N), precision="f64")
    st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.array([0.5, -0.2, 0.1]))
    goal = np.array([5.0, 3.0, 2.0])
    tr = pl.plan_shooting(st, goal, n_samples=S, iters=K, step=0.9, sigma=2.0, seed=3, precision="f32")
    T_first = pl.last_result["T"].copy()
    assert tr.positions.shape == (N, 3) and tr.thrusts.shape == (N,) and np.all(tr.thrusts > 0)
    assert np.allclose(tr.positions[0], st.position) and np.allclose(tr.velocities[0], st.velocity)
    s_win = pl.last_result["sample"]
    cfg = orc.OracleConfig(prediction_horizon=N)
    import torch
    from dart_planner_amd.distributed import shooting_samples
    T0 = shooting_samples(pl._params(), S, 2.0, 3, gpu_ops.be.device, torch.float32)[:, s_win].double().cpu().numpy().reshape(1, N, 3)
    f32 = lambda a: np.asarray(a, np.float32).astype(float)
    Tr, cr, _, _ = pc.oracle_iterate(f32(st.position)[None], f32(st.velocity)[None], f32(goal)[None], f32(T0), cfg, K, 0.9)
    Tw = np.asarray(tr.accelerations) * cfg.mass + [0, 0, cfg.mass * cfg.gravity]                # accelerations = T/m - g e3
    assert np.max(np.abs(Tw - Tr[0])) <= 2e-4
    assert abs(pl.last_result["cost"] - cr[0]) <= 5e-5 * cr[0]
    # the hover sample after the same iterations costs at least as much
    Th, ch, _, _ = pc.oracle_iterate(f32(st.position)[None], f32(st.velocity)[None], f32(goal)[None], np.tile([0, 0, cfg.hover_thrust], (1, N, 1)), cfg, K, 0.9)
    assert pl.last_result["cost"] <= ch[0] * (1 + 1e-5)
    # dynamics constraints of the reference (planner.py:426-462) on the returned plan: zero
    X = orc.pack(np.asarray(tr.positions)[None], np.asarray(tr.velocities)[None], Tw[None])
    R = orc.dynamics_residual(X, st.position[None], st.velocity[None], cfg)
    assert np.max(np.abs(R)) <= 1e-9
    # plan_shooting replays one captured hipGraph; the eager chain of the same launches (what several ranks run) gives the same winner and numbers
    from dart_planner_amd.distributed import sharded_shooting_plan
    eager = sharded_shooting_plan(gpu_ops, pl._params(), st.position, st.velocity, goal, S, K, 0.9, 2.0, 3, "f32")
    assert eager["sample"] == s_win and np.array_equal(eager["T"], T_first)
    # replays are reproducible and follow their inputs
    tr2 = pl.plan_shooting(st, goal, n_samples=S, iters=K, step=0.9, sigma=2.0, seed=3, precision="f32")
    assert np.array_equal(tr2.positions, tr.positions) and np.array_equal(tr2.thrusts, tr.thrusts)
    st3 = DroneState(timestamp=0.0, position=np.array([1.0, -2.0, 3.0]), velocity=np.array([0.0, 0.3, 0.0]))
    tr3 = pl.plan_shooting(st3, np.array([-4.0, 0.0, 1.0]), n_samples=S, iters=K, step=0.9, sigma=2.0, seed=3, precision="f32")
    e3 = sharded_shooting_plan(gpu_ops, pl._params(), st3.position, st3.velocity, np.array([-4.0, 0.0, 1.0]), S, K, 0.9, 2.0, 3, "f32")
    assert pl.last_result["sample"] == e3["sample"] and np.array_equal(e3["T"], pl.last_result["T"]) and np.allclose(tr3.positions[0], st3.position)
