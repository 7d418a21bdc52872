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
from dart_planner_amd.ops import Ops  # noqa: E402
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


def test_closed_loop_reproduces_reference_loops(emu_ops, golden_controller):
    data, meta = golden_controller
    assert cc.check_closed_loops_golden(harness(emu_ops, np.float64), data, meta) <= 1e-8


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_closed_loop_random_batch_vs_oracle(emu_ops, dt):
    cc.check_closed_loop_vs_oracle(harness(emu_ops, dt), B=70, N=12, nsteps=24, seed=1)
    if dt == np.float64:
        cc.check_closed_loop_vs_oracle(harness(emu_ops, dt), B=5, N=12, nsteps=24, seed=2, per_drone_plans=False)
