"""CPU suite: the fused receding-horizon Monte-Carlo (se3mpc_monte_carlo_*: every planning cycle of every drone inside ONE kernel) against
the two-launch form it fuses (se3mpc_solve_* + se3mpc_closed_loop_* per cycle, dart_planner_amd/control/closed_loop.py) -- the same code,
so the same BITS -- on the product kernels compiled for the host (tests/emu)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))


@pytest.fixture(scope="module")
def cpu_ops():
    import build_emu
    from numpy_backend import TorchCpuBackend
    from dart_planner_amd import capi
    from dart_planner_amd.ops import Ops
    return Ops(TorchCpuBackend(), capi.Library(build_emu.build()))


def scene(B, dtype, seed=5):
    g = torch.Generator(); g.manual_seed(seed)
    p0 = torch.tensor([0.0, 0.0, 2.0], dtype=dtype).repeat(B, 1) + 0.2 * torch.randn(B, 3, dtype=dtype, generator=g)
    v0 = 0.3 * torch.randn(B, 3, dtype=dtype, generator=g)
    goal = torch.tensor([8.0, 0.0, 5.0], dtype=dtype).repeat(B, 1).contiguous()
    wind = torch.randn(B, 3, dtype=dtype, generator=g).contiguous()
    return p0, v0, goal, wind


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("N,B", [(6, 19), (13, 5), (30, 3)])
def test_fused_monte_carlo_equals_the_two_launch_form(cpu_ops, dtype, N, B):
    from dart_planner_amd.capi import Params
    from dart_planner_amd.control.closed_loop import ClosedLoopMonteCarlo
    prm = Params.reference_defaults(horizon=N)
    mc = ClosedLoopMonteCarlo(cpu_ops, prm)
    p0, v0, goal, wind = scene(B, dtype)
    cycles, substeps, sim_dt = 4, 5, 0.01
    for w in (wind, None, wind[0].contiguous()):
        a = mc.run(p0, v0, goal, cycles, substeps, sim_dt, wind=w)
        b = mc.run_fused(p0, v0, goal, cycles, substeps, sim_dt, wind=w, want_last_plan=True)
        for key in ("pos", "vel", "att", "omega", "time", "controller_state"):
            assert torch.equal(a[key], b[key]), key
        # the last cycle's plan: what a solve from the state before the last act phase returns
        assert b["last_plan"]["x"].shape == (B, 9 * N) and torch.isfinite(b["last_plan"]["x"]).all()
    assert float((a["pos"] - p0).abs().max()) > 1e-3                      # the drones did move


def test_fused_monte_carlo_argument_checks(cpu_ops):
    from dart_planner_amd.capi import Params
    lib = cpu_ops.lib
    prm, cp, sp = Params.reference_defaults(), lib.controller_default_params(), lib.simulator_default_params()
    B = 3
    f = lambda *s: torch.zeros(*s, dtype=torch.float32)
    time, st, over = torch.zeros(B, dtype=torch.float64), torch.zeros(B, 12, dtype=torch.float64), torch.zeros(1, dtype=torch.int32)
    pos, vel, att, om, goal = f(B, 3), f(B, 3), f(B, 3), f(B, 3), f(B, 3)
    args = lambda **kw: [prm, cp, sp, kw.get("B", B), kw.get("cycles", 1), 2, 0.01, goal.data_ptr(), 0, 0, time.data_ptr(), pos.data_ptr(), vel.data_ptr(),
                         att.data_ptr(), om.data_ptr(), st.data_ptr(), 0, 0, 0, kw.get("over", over.data_ptr()), None]
    assert lib.loop_status("monte_carlo", "f32", *args()) == 0
    assert lib.loop_status("monte_carlo", "f32", *args(B=0)) == 0
    assert lib.loop_status("monte_carlo", "f32", *args(B=-1)) == -3
    assert lib.loop_status("monte_carlo", "f32", *args(cycles=-1)) == -3
    assert lib.loop_status("monte_carlo", "f32", *args(over=0)) == -1
