"""CPU suite: the UNMODIFIED product kernels (dart_planner_amd/csrc/*.hip) compiled for the host by
tests/emu (one std::thread per lane) and driven through the same C ABI and the same Ops front-end
as on the GPU, checked against the oracle and the reference's golden vectors.  This is test
infrastructure (lanes are fibers) to debug kernel arithmetic and host logic without a GPU; the product never loads
tests/emu/libse3mpc_emu.so, and the `-m gpu` suite repeats every check on the real library."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import build_emu  # noqa: E402
from numpy_backend import NumpyBackend  # noqa: E402

from dart_planner_amd import capi  # noqa: E402
from dart_planner_amd.ops import Ops, INFO_DTYPE  # noqa: E402
import parity_checks as pc  # noqa: E402


@pytest.fixture(scope="module")
def emu_ops():
    lib = capi.Library(build_emu.build())
    return Ops(NumpyBackend(), lib)


def harness(ops, dt):
    return pc.Harness(ops, lambda a: a, lambda a: a, dt)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(1, 5), (6, 70), (30, 66), (7, 9), (24, 65)])
def test_lane_kernels(emu_ops, dt, N, B):
    # B = 70 / 66 leave a partial last wavefront (tail lanes); N = 7 takes the generic (non-template) path
    pc.check_lane_kernels(harness(emu_ops, dt), N, B, seed=N, variants=(0, 1, 2, 3, 4, 5, 6))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 70), (30, 66), (7, 9), (24, 65)])
def test_rollout_iterate(emu_ops, dt, N, B):
    # N = 6 / 30: exact-N register kernels (f64: 30 takes the through-memory path), 24: the 32-step bucket (f32), 7: through memory
    pc.check_rollout_iterate(harness(emu_ops, dt), N, B, seed=N, iters=4)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 70), (30, 40), (7, 9), (24, 65)])
def test_rollout_iterate_obstacles(emu_ops, dt, N, B):
    # N = 6 / 30: exact-N register kernels (f64: 30 takes the through-memory path), 24: the 32-step bucket (f32), 7: through memory
    pc.check_rollout_iterate_obstacles(harness(emu_ops, dt), N, B, seed=N, iters=3)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("K", [17, 28, 40])
def test_rollout_iterate_obstacles_table_sizes(emu_ops, dt, K):
    # the helpers' register-resident sphere table holds 8 / 16 / 24 / 32 spheres (f64: 8 / 16); longer tables are swept from LDS by everyone
    pc.check_rollout_iterate_obstacles(harness(emu_ops, dt), 6, 20, seed=K, iters=2, K=K)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("N,B", [(6, 70), (30, 200), (1, 1), (64, 65)])
def test_shooting_finish(emu_ops, dt, N, B):
    pc.check_shooting_finish(harness(emu_ops, dt), N, B, seed=N)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_iteration_loop_keys(emu_ops, dt):
    pc.check_iterate_keys(harness(emu_ops, dt), 6, 200, seed=1)
    pc.check_iterate_keys(harness(emu_ops, dt), 6, 33, seed=2, index_base=1000)


def test_keys_with_nonfinite_costs(emu_ops):
    pc.check_key_nonfinite(harness(emu_ops, np.float32))


def test_lane_kernels_other_dt(emu_ops):
    pc.check_lane_kernels(harness(emu_ops, np.float64), 20, 8, seed=2, dt=0.05)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_reproduces_reference_solves(emu_ops, golden_solve, dt):
    data, meta = golden_solve
    # f64: every solve the reference produced; f32: every other one (contract case, N = 1/2, box-clipped
    # starts, dt = 0.1, ABNORMAL line searches are in both halves)
    keys = None if dt == np.float64 else {c["key"] for i, c in enumerate(meta["cases"]) if i % 2 == 0 or c["tag"] != "random"}
    worst = pc.check_solver_golden(harness(emu_ops, dt), data, meta, keys=keys)
    assert worst <= (1e-4 if dt == np.float32 else 1e-9)


def test_solver_published_cauchy_search(emu_ops, golden_solve):
    """se3mpc_set_solver_variant(1): the published sequential breakpoint search everywhere (no closed form while the memory is empty).
    Same counts as the reference on every golden solve, and the f64 thrust block to 1e-9 -- the 1e-7 carve-out of the default path
    (closed-form Cauchy point) is not needed."""
    data, meta = golden_solve
    emu_ops.lib.set_solver_variant(1)
    try:
        worst = pc.check_solver_golden(harness(emu_ops, np.float64), data, meta, thrust_tol=1e-9)
    finally:
        emu_ops.lib.set_solver_variant(0)
    assert worst <= 1e-9


def test_solver_config1_sample(emu_ops, golden_cfg1):
    data, meta = golden_cfg1
    two_its = [int(i) for i in np.nonzero(data["info"][:, 0] != 1)[0]]          # the one solve that takes two iterations
    worst = pc.check_solver_cfg1(harness(emu_ops, np.float64), data, meta, rows=[0, 1, 2] + two_its)
    assert worst <= 1e-9


def test_batched_solve_matches_scipy_odd_horizons(emu_ops):
    # horizons that are not template instantiations (J = 2, 5 with padding lanes), a few problems each
    for N, B in ((13, 6), (33, 4)):
        worst, mism = pc.check_solver_vs_oracle(harness(emu_ops, np.float64), N, B, seed=N)
        assert mism == 0.0 and worst <= 1e-9


@pytest.mark.parametrize("group", [8, 16, 32, 64])
def test_solver_group_sizes(emu_ops, golden_solve, group):
    """The packed solver: 64 / group problems share a wavefront and run their own control flow.  Every group size on the reference's
    golden solves (horizons that do not fit a group run at the next size), and batches of random problems -- different iteration
    counts side by side in one wavefront, a ragged last wavefront -- against SciPy one by one."""
    data, meta = golden_solve
    keys = {c["key"] for i, c in enumerate(meta["cases"]) if c["N"] <= group and (i % 3 == 0 or c["tag"] != "random")}
    assert pc.check_solver_golden(harness(emu_ops, np.float64), data, meta, keys=keys, group=group) <= 1e-9
    N = {8: 6, 16: 13, 32: 30, 64: 40}[group]
    B = 2 * (64 // group) + 3 if group < 64 else 3
    worst, mism = pc.check_solver_vs_oracle(harness(emu_ops, np.float64), N, B, seed=group, group=group)
    assert mism == 0.0 and worst <= 1e-9
    if group <= 16:
        worst, mism = pc.check_solver_vs_oracle(harness(emu_ops, np.float32), N, B, seed=group + 1, group=group)
        assert mism <= 0.1 and worst <= 1e-4


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_on_the_knife_edge_problem(emu_ops, dt):
    for group in (None, 64):
        branch, err = pc.check_solver_bifurcation_case(harness(emu_ops, dt), group=group)
        print(f"knife-edge problem, {np.dtype(dt).name}, group {group}: {branch}, {err:.2e} m")


def test_solver_extraction_and_cold_start(emu_ops):
    pc.check_solver_extraction(harness(emu_ops, np.float64), 6, 4)
    pc.check_solver_extraction(harness(emu_ops, np.float32), 20, 3)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_spheres_from_grid(emu_ops, golden_mapper, dt):
    # (float32 under emulation: two of the four golden grids; the GPU suite sweeps all four in both types)
    pc.check_spheres_from_grid(harness(emu_ops, dt), *golden_mapper, only=None if dt == np.float64 else {"m_sparse_", "m_test_file_divisor_"})


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_solver_edge_inputs(emu_ops, dt):
    pc.check_solver_edge_inputs(harness(emu_ops, dt))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_transpose(emu_ops, dt):
    pc.check_transpose(harness(emu_ops, dt), shapes=((270, 300), (300, 270), (576, 257), (257, 576), (5, 512), (512, 5), (70, 70), (90, 255)))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_population_sums(emu_ops, dt):
    pc.check_population_sums(harness(emu_ops, dt), rows=18, B=333)


def test_empty_batch_and_error_codes(emu_ops):
    lib, be = emu_ops.lib, emu_ops.be
    prm = capi.Params.reference_defaults(horizon=6)
    z = np.zeros((3, 0), dtype=np.float32)
    assert emu_ops.init(prm, z, z, z).shape == (54, 0)                       # B == 0 is a no-op
    out = emu_ops.solve(prm, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)))
    assert out["x"].shape == (0, 54)
    a = np.zeros((3, 4), dtype=np.float32); X = np.zeros((54, 4), dtype=np.float32); f = np.zeros(4, dtype=np.float32)
    st = lambda *args, **kw: lib.call_status(*args, **kw)
    assert st("cost_grad", "f32", 4, 4, 0, be.ptr(a), be.ptr(f), 0, 0, params=prm) == -1          # NULL X
    assert st("cost_grad", "f32", 4, 4, be.ptr(X), 0, be.ptr(f), 0, 0, params=prm) == -1          # NULL goal with has_goal
    assert st("cost_grad", "f32", 4, 4, be.ptr(X), 0, be.ptr(f), 0, 0, params=prm.copy(has_goal=0)) == 0
    assert st("cost_grad", "f32", 5, 4, be.ptr(X), be.ptr(a), be.ptr(f), 0, 0, params=prm) == -3  # ld < B
    assert st("cost_grad", "f32", -1, 4, be.ptr(X), be.ptr(a), be.ptr(f), 0, 0, params=prm) == -3
    for bad, code in ((dict(horizon=0), -2), (dict(horizon=65), -2), (dict(dt=0.0), -4), (dict(mass=float("nan")), -4),
                      (dict(max_corrections=11), -4), (dict(max_thrust=1.0), -4)):
        assert lib.check_params(prm.copy(**bad)) == code, bad
        assert st("cost_grad", "f32", 4, 4, be.ptr(X), be.ptr(a), be.ptr(f), 0, 0, params=prm.copy(**bad)) == code
    assert st("obstacle_residual", "f32", 4, 4, be.ptr(X), 0, 257, 0, 0, 0, 0, params=prm) == -3  # K > SE3MPC_MAX_SPHERES
    assert st("rollout_cost_grad", "f32", 4, 4, be.ptr(a), be.ptr(a), be.ptr(a), be.ptr(X), be.ptr(f), 0, be.ptr(X), 0,
              0, 0, 0, params=prm) == -1                                                          # P without V
    with pytest.raises(capi.Se3mpcError):
        lib.call("cost_grad", "f32", 4, 4, 0, be.ptr(a), be.ptr(f), 0, 0, params=prm)
    # size limits follow the tallest operand and the element size (32-bit buffer offsets): bench.py --sweep's
    # 4 M-rollout batch is legal for the rollout (3N rows of f32), a full 9N-row f64 block of that width is not
    p30 = capi.Params.reference_defaults(horizon=30)
    assert st("rollout_cost_grad", "f32", 0, 1 << 22, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, params=p30) == 0
    assert st("rollout_cost_grad", "f64", 0, 1 << 23, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, params=p30) == -3
    assert st("cost_grad", "f32", 0, 1 << 21, 0, 0, 0, 0, 0, params=p30) == 0
    assert st("cost_grad", "f64", 0, 1 << 21, 0, 0, 0, 0, 0, params=p30) == -3


def test_defaults_and_key_codec(emu_ops):
    lib = emu_ops.lib
    assert lib.default_params().as_dict() == capi.Params.reference_defaults().as_dict()
    assert lib.abi_version() == 1
    for c in (0.0, 1.5, 3.0e6, -2.0, float("inf")):
        bits = np.float32(c).view(np.uint32)
        ob = (~bits) & 0xFFFFFFFF if bits & 0x80000000 else bits | 0x80000000
        key = (int(ob) << 32) | 1234
        assert lib.key_index(key) == 1234 and lib.key_cost(key) == np.float32(c)


# ------------------------------------------------------------------ consumer side of the contract (closed_loop.hip)
import json  # noqa: E402
import controller_checks as cc  # noqa: E402


@pytest.fixture(scope="module")
def golden_controller():
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return np.load(os.path.join(g, "controller_cases.npz")), json.load(open(os.path.join(g, "controller_cases.json")))


def test_controller_defaults(emu_ops):
    cc.check_defaults(harness(emu_ops, np.float64))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_control_reproduces_reference_sequences(emu_ops, golden_controller, dt):
    data, meta = golden_controller
    cc.check_control_sequences(harness(emu_ops, dt), data, meta)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_control_fast_reproduces_reference_sequences(emu_ops, golden_controller, dt):
    data, meta = golden_controller
    cc.check_fast_sequences(harness(emu_ops, dt), data, meta)
    cc.check_fast_batch_vs_oracle(harness(emu_ops, dt), B=70, calls=6, seed=5)


def test_wide_lane_kernels_equal_the_dword_forms(emu_ops):
    pc.check_wide_kernels(harness(emu_ops, np.float32), N=6, B=132, seed=3)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_controller_building_blocks(emu_ops, dt):
    cc.check_building_blocks(harness(emu_ops, dt), B=60, seed=7)


def test_closed_loop_reproduces_reference_loops(emu_ops, golden_controller):
    data, meta = golden_controller
    assert cc.check_closed_loops_golden(harness(emu_ops, np.float64), data, meta) <= 1e-8


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_closed_loop_random_batch_vs_oracle(emu_ops, dt):
    cc.check_closed_loop_vs_oracle(harness(emu_ops, dt), B=70, N=12, nsteps=24, seed=1)
    if dt == np.float64:
        cc.check_closed_loop_vs_oracle(harness(emu_ops, dt), B=5, N=12, nsteps=24, seed=2, per_drone_plans=False)


def test_round2_entry_points_error_codes_and_empty_batches(emu_ops):
    """se3mpc_rollout_iterate / projected_step / control / control_fast / control_plan / simulator_step / closed_loop / controller_reset:
    B == 0 is a no-op, NULL and shape errors come back as status codes, bad parameter structs are refused."""
    lib, be = emu_ops.lib, emu_ops.be
    prm = capi.Params.reference_defaults(horizon=6)
    st = lambda *args, **kw: lib.call_status(*args, **kw)
    a = np.zeros((3, 4), dtype=np.float32); T = np.zeros((18, 4), dtype=np.float32); f = np.zeros(4, dtype=np.float32)
    it = lambda B, ld, nb, iters, step, T_in, T_out, cost: st("rollout_iterate", "f32", B, ld, nb, iters, step, be.ptr(a), be.ptr(a), be.ptr(a), T_in, T_out, 0,
                                                              cost, 0, 0, 0, 0, params=prm)
    assert it(0, 4, 1, 3, 0.5, 0, 0, 0) == 0                                   # B == 0
    assert it(4, 4, 1, 3, 0.5, be.ptr(T), be.ptr(T), be.ptr(f)) == 0          # in place
    assert it(4, 4, 1, 3, 0.5, 0, be.ptr(T), be.ptr(f)) == -1                 # NULL T_in
    assert it(4, 4, 1, -1, 0.5, be.ptr(T), be.ptr(T), be.ptr(f)) == -3        # iters < 0
    assert it(4, 4, 0, 3, 0.5, be.ptr(T), be.ptr(T), be.ptr(f)) == -3         # nbatch < 1
    assert it(5, 4, 1, 3, 0.5, be.ptr(T), be.ptr(T), be.ptr(f)) == -3         # ld < B
    assert it(4, 4, 1, 3, float("nan"), be.ptr(T), be.ptr(T), be.ptr(f)) == -4
    assert st("projected_step", "f32", 4, 4, 0.5, be.ptr(T), 0, be.ptr(T), 0, params=prm) == -1
    assert st("projected_step", "f32", 0, 4, 0.5, 0, 0, 0, 0, params=prm) == 0
    cp, sp = lib.controller_default_params(), lib.simulator_default_params()
    ls = lambda *args: lib.loop_status(*args)
    z3 = np.zeros((2, 3)); t = np.zeros(2); state = emu_ops.controller_state(cp, 2); th = np.zeros(2); tq = np.zeros((2, 3))
    p = be.ptr
    ctl = lambda cpx, B, time, pos, stt: ls("control", "f64", cpx, B, time, pos, p(z3), p(z3), p(z3), p(z3), p(z3), 0, 0, 0, stt, p(th), p(tq), 0, 0, 0, 0)
    assert ctl(cp, 0, 0, 0, 0) == 0
    assert ctl(cp, 2, p(t), p(z3), p(state)) == 0
    assert ctl(cp, 2, 0, p(z3), p(state)) == -1 and ctl(cp, 2, p(t), p(z3), 0) == -1 and ctl(cp, -1, p(t), p(z3), p(state)) == -3
    bad = lib.controller_default_params(); bad.mass = 0.0
    assert ctl(bad, 2, p(t), p(z3), p(state)) == -4
    bad = lib.controller_default_params(); bad.kp_pos[1] = float("inf")
    assert ctl(bad, 2, p(t), p(z3), p(state)) == -4
    bad = lib.controller_default_params(); bad.yaw_fallback_method = 7
    assert ctl(bad, 2, p(t), p(z3), p(state)) == -4
    assert lib._dll.se3mpc_controller_reset(None, 2, p(state), 0) == -1
    fst = lambda cpx, m, g, B, dt, pos, stt: ls("control_fast", "f64", cpx, m, g, B, dt, pos, p(z3), p(z3), p(z3), p(z3), p(z3), 0, 0, 0, stt, p(th), p(tq), 0, 0)
    assert fst(cp, 1.0, 9.80665, 0, 0.0025, 0, 0) == 0 and fst(cp, 1.0, 9.80665, 2, 0.0025, p(z3), p(state)) == 0
    assert fst(cp, 1.0, 9.80665, 2, 0.5, p(z3), p(state)) == 0                 # an invalid dt is an answer of the path (hover thrust), not an error
    assert th[0] == 9.80665 and not tq.any()
    assert fst(cp, 1.0, 9.80665, 2, 0.0025, 0, p(state)) == -1 and fst(cp, 1.0, 9.80665, 2, 0.0025, p(z3), 0) == -1
    assert fst(cp, 1.0, 9.80665, -1, 0.0025, p(z3), p(state)) == -3
    assert fst(cp, 0.0, 9.80665, 2, 0.0025, p(z3), p(state)) == -4 and fst(cp, 1.0, float("nan"), 2, 0.0025, p(z3), p(state)) == -4
    assert fst(cp, 1.0, 9.80665, 2, float("nan"), p(z3), p(state)) == -4 and fst(bad, 1.0, 9.80665, 2, 0.0025, p(z3), p(state)) == -4
    ts = np.array([0.0, 1.0]); P = np.zeros((2, 3))
    loop = lambda spx, B, nsteps, dt, N, tsp, Pp: ls("closed_loop", "f64", cp, spx, B, nsteps, dt, N, tsp, 0, Pp, 0, 0, 0, 0, 0, p(t), p(z3), p(z3), p(z3), p(z3),
                                                     p(state), 0, 0, -1, None, 1, 0, 0, 0, 0, 0)
    assert loop(sp, 2, 3, 0.01, 2, p(ts), p(P)) == 0
    assert loop(sp, 0, 3, 0.01, 2, 0, 0) == 0 and loop(sp, 2, 0, 0.01, 2, 0, 0) == 0      # nothing to do
    assert loop(sp, 2, 3, 0.01, 2, 0, p(P)) == -1 and loop(sp, 2, 3, 0.01, 2, p(ts), 0) == -1
    assert loop(sp, 2, -1, 0.01, 2, p(ts), p(P)) == -3 and loop(sp, 2, 3, 0.01, 0, p(ts), p(P)) == -3
    assert loop(sp, 2, 3, float("inf"), 2, p(ts), p(P)) == -4
    bads = lib.simulator_default_params(); bads.inertia[2] = 0.0
    assert loop(bads, 2, 3, 0.01, 2, p(ts), p(P)) == -4
    assert ls("closed_loop", "f64", cp, sp, 2, 3, 0.01, 2, p(ts), 0, p(P), 0, 0, 0, 0, 0, p(t), p(z3), p(z3), p(z3), p(z3), p(state), 0, 0, 1, None, 1, 0, 0, 0, 0, 0) == -1   # gust step without a gust vector
    assert ls("simulator_step", "f64", sp, 2, 0.01, p(th), p(tq), 0, 0, p(t), p(z3), p(z3), p(z3), p(z3), 0) == 0
    assert ls("simulator_step", "f64", sp, 2, 0.01, 0, p(tq), 0, 0, p(t), p(z3), p(z3), p(z3), p(z3), 0) == -1
    assert ls("control_plan", "f64", cp, 2, p(t), p(t), p(z3), p(z3), p(z3), p(z3), 2, p(ts), 0, p(P), 0, 0, 0, 0, 0, p(state), p(th), p(tq), 0, 0, 0, 0, 0) == 0
    assert ls("control_plan", "f64", cp, 2, p(t), 0, p(z3), p(z3), p(z3), p(z3), 2, p(ts), 0, p(P), 0, 0, 0, 0, 0, p(state), p(th), p(tq), 0, 0, 0, 0, 0) == -1
    assert lib._dll.se3mpc_set_solver_variant(2) == -3 and lib._dll.se3mpc_set_solver_variant(24 << 8) == -3 and lib._dll.se3mpc_set_solver_variant(0) == 0


def test_controller_mirror_quaternion_states_reproduce_the_reference(emu_ops, golden_controller):
    """Quaternion attitudes through the mirror's GeometricController.compute_control (normalised, identity below 1e-6: controller.py:770-803) against
    the commands the REFERENCE's controller returned for the same states (tests/golden/make_golden_controller.py block F: lengths 1, 3, 0.25, 1e-3
    and 1e-9), float64 to 1e-9."""
    from numpy_backend import TorchCpuBackend
    from dart_planner_amd.common.types import DroneState
    from dart_planner_amd.control.geometric_controller import GeometricController
    g = golden_controller[0] if isinstance(golden_controller, tuple) else golden_controller
    ops = Ops(TorchCpuBackend(), emu_ops.lib)
    n = len(g["quat_thrust"])
    assert n == 8
    for i in range(n):
        c = GeometricController(tuning_profile="sitl_optimized", precision="f64")
        c._ops = ops; c._device = "cpu"
        st = DroneState(timestamp=5.0, position=g["quat_pos"][i], velocity=g["quat_vel"][i], attitude=g["quat_quat"][i], angular_velocity=g["quat_omega"][i])
        cmd = c.compute_control(st, g["quat_dpos"][i], g["quat_dvel"][i], g["quat_dacc"][i], float(g["quat_yaw"][i]), float(g["quat_yaw_rate"][i]))
        assert abs(cmd.thrust - g["quat_thrust"][i]) <= 1e-9 * abs(g["quat_thrust"][i]), i
        assert np.max(np.abs(np.asarray(cmd.torque) - g["quat_torque"][i])) <= 1e-9, i


def test_round3_entry_points_error_codes(emu_ops):
    """se3mpc_rollout_iterate_obstacles / shooting_finish: status codes for NULL, shape and parameter errors; keys that were never written or belong
    to another index range still give a defined read (sample 0)."""
    lib, be = emu_ops.lib, emu_ops.be
    prm = capi.Params.reference_defaults(horizon=6)
    st = lambda *args, **kw: lib.call_status(*args, **kw)
    p = be.ptr
    a = np.zeros((3, 4), dtype=np.float32); T = np.ones((18, 4), dtype=np.float32); f = np.zeros(4, dtype=np.float32)
    sph = np.zeros((2, 4), dtype=np.float32)
    ito = lambda B, K, w, sp: st("rollout_iterate_obstacles", "f32", B, 4, 1, 2, 0.5, p(a), p(a), p(a), p(T), p(T), 0, p(f), 0, sp, K, w, 0, 0, 0, 0, params=prm)
    assert ito(0, 2, 1.0, p(sph)) == 0 and ito(4, 2, 1.0, p(sph)) == 0 and ito(4, 0, 1.0, 0) == 0
    assert ito(4, 2, 1.0, 0) == -1 and ito(4, -1, 1.0, p(sph)) == -3 and ito(4, 257, 1.0, p(sph)) == -3
    assert ito(4, 2, -1.0, p(sph)) == -4 and ito(4, 2, float("nan"), p(sph)) == -4
    keys = np.full(1, -1, dtype=np.int64); state = np.zeros(9); out = np.zeros(19 * 6 + 3); sph64 = np.zeros((2, 4))
    T = np.ones((18, 4), dtype=np.float32)                                       # (the descent above ran in place)
    fin = lambda B, Tp, kp, ns, stp, sp, K, w, op: st("shooting_finish", "f32", B, 4, Tp, kp, ns, 0, stp, sp, K, w, op, 0, 0, params=prm)
    assert fin(4, p(T), p(keys), 1, p(state), p(sph64), 2, 1.0, p(out)) == 0
    assert np.array_equal(out[36:54].reshape(6, 3), np.ones((6, 3)))             # sentinel keys: sample 0's column, a defined read
    assert fin(0, p(T), p(keys), 1, p(state), 0, 0, 0.0, p(out)) == -3           # a plan needs a sample
    assert fin(4, 0, p(keys), 1, p(state), 0, 0, 0.0, p(out)) == -1 and fin(4, p(T), 0, 1, p(state), 0, 0, 0.0, p(out)) == -1
    assert fin(4, p(T), p(keys), 1, 0, 0, 0, 0.0, p(out)) == -1 and fin(4, p(T), p(keys), 1, p(state), 0, 0, 0.0, 0) == -1
    assert fin(4, p(T), p(keys), 0, p(state), 0, 0, 0.0, p(out)) == -3 and fin(4, p(T), p(keys), 1, p(state), 0, 2, 1.0, p(out)) == -1
    assert fin(4, p(T), p(keys), 1, p(state), p(sph64), 2, float("inf"), p(out)) == -4


def test_plan_host_entry_point(emu_ops):
    """se3mpc_plan_host_*: launch + completion ticket in one call.  Same numbers as se3mpc_solve_* on the same buffers; the ticket lands in
    the completion word; a batch that needs more than one wavefront, a NULL completion word and a stale ticket are refused."""
    import ctypes as C
    lib = emu_ops.lib
    for suf, dt in (("f32", np.float32), ("f64", np.float64)):
        for N, B in ((6, 1), (6, 8), (30, 2), (40, 1)):
            prm = capi.Params.reference_defaults(horizon=N)
            rng = np.random.default_rng(N + B)
            p0, v0, goal = (rng.uniform(-5, 5, (B, 3)).astype(dt) for _ in range(3))
            outs = []
            for mode in ("solve", "plan_host"):
                X = np.zeros((B, 9 * N), dt); info = np.zeros((B,), INFO_DTYPE)
                acc, att, rates = (np.zeros((B, N, 3), dt) for _ in range(3)); thr = np.zeros((B, N), dt)
                ptr = lambda a: a.ctypes.data
                args = [B, ptr(p0), ptr(v0), ptr(goal), 0, ptr(X), ptr(info), ptr(acc), ptr(att), ptr(rates), ptr(thr)]
                if mode == "solve":
                    lib.call("solve", suf, *args, None, params=prm)
                else:
                    done = np.zeros(1, np.uint64)
                    lib.call("plan_host", suf, *args, ptr(done), 7, 1e6, None, params=prm)
                    assert int(done[0]) == 7
                    assert lib.call_status("plan_host", suf, *args, ptr(done), 7, 1e6, None, params=prm) == -3      # stale ticket
                    assert lib.call_status("plan_host", suf, *args, 0, 8, 1e6, None, params=prm) == -1               # no completion word
                outs.append((X, info, acc, att, rates, thr))
            for a, b in zip(*outs):
                assert np.array_equal(a, b)
    # more than one wavefront's worth of problems
    prm = capi.Params.reference_defaults(horizon=30)
    B = 3
    z = np.zeros((B, 3), np.float32); X = np.zeros((B, 270), np.float32); info = np.zeros((B,), INFO_DTYPE); done = np.zeros(1, np.uint64)
    assert lib.call_status("plan_host", "f32", B, z.ctypes.data, z.ctypes.data, z.ctypes.data, 0, X.ctypes.data, info.ctypes.data, 0, 0, 0, 0,
                           done.ctypes.data, 1, 1e6, None, params=prm) == -3
