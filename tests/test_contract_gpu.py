"""GPU: the 11 contract cases on the real library, f64 and f32 planner precision."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import contract_cases as cc  # noqa: E402


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", cc.ALL_CASES, ids=lambda c: c.__name__)
def test_contract_case(case, precision):
    def attach(planner):
        planner.precision = precision
    rig = cc.Rig(attach)
    if case is cc.case_performance_benchmark:
        ms = case(rig)
        print(f"plan_trajectory ({precision}): {ms:.3f} ms")
    elif case is cc.case_planner_outputs_complete_trajectory:
        case(rig, tol=1e-4 if precision == "f32" else 1e-9)
    else:
        case(rig)
    assert rig.planner._get_ops().lib.path.endswith("libse3mpc.so")


@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 1e-4)])
def test_receding_horizon_warm_start(precision, tol):
    def attach(planner):
        planner.precision = precision
    cc.case_receding_horizon_warm_start(cc.Rig(attach), tol=tol)
