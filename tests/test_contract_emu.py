"""CPU: the 11 contract cases with the planner's real host code driving the product kernels compiled
for the host (tests/emu).  The GPU suite repeats them on libse3mpc.so."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import build_emu  # noqa: E402
from numpy_backend import TorchCpuBackend  # noqa: E402

import contract_cases as cc  # noqa: E402
from dart_planner_amd import capi  # noqa: E402
from dart_planner_amd.ops import Ops  # noqa: E402


@pytest.fixture(scope="module")
def attach():
    ops = Ops(TorchCpuBackend(), capi.Library(build_emu.build()))

    def _attach(planner):
        planner._ops = ops
    return _attach


@pytest.mark.parametrize("case", cc.ALL_CASES, ids=lambda c: c.__name__)
def test_contract_case(attach, case):
    rig = cc.Rig(attach)
    if case is cc.case_performance_benchmark:
        case(rig, plan_ms=60000.0)      # the emulation is a single host thread: the 50 ms bound is for the GPU suite
    else:
        case(rig)


@pytest.mark.parametrize("case", cc.OTHER_REFERENCE_CASES, ids=lambda c: c.__name__)
def test_other_reference_planner_tests(attach, case):
    rig = cc.Rig(attach)
    if case is cc.case_se3_mpc_speed:
        case(rig, n=10, mean_ms=60000.0, single_ms=60000.0)      # timing bounds are for the GPU suite
    else:
        case(rig)


def test_private_path_methods_match_reference(attach, golden_path):
    cc.case_private_path_methods(cc.Rig(attach), *golden_path)


def test_receding_horizon_warm_start(attach):
    cc.case_receding_horizon_warm_start(cc.Rig(attach))


def test_planner_refuses_to_run_without_a_gpu():
    """No CPU fallback: on a box without a HIP device the first plan raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rig = cc.Rig()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rig.planner.plan_trajectory(rig.initial_state, rig.goal_position)
