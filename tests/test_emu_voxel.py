"""CPU suite for the voxel map: the UNMODIFIED dart_planner_amd/csrc/voxel_map.hip compiled for the host by tests/emu
and driven through the C ABI and the product's host classes, checked against the reference mapper's golden outputs.
Test infrastructure; the `-m gpu` suite (tests/test_gpu_voxel.py) repeats every check on the real library."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
import build_emu  # noqa: E402
from numpy_backend import NumpyBackend, TorchCpuBackend  # noqa: E402

from dart_planner_amd import capi  # noqa: E402
from dart_planner_amd.ops import Ops  # noqa: E402
import voxel_checks as vc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu_ops():
    return Ops(NumpyBackend(), capi.Library(build_emu.build()))


@pytest.fixture(scope="module")
def golden_map():
    return (np.load(os.path.join(HERE, "golden", "mapper_map.npz")), json.load(open(os.path.join(HERE, "golden", "mapper_map.json"))))


@pytest.mark.parametrize("scene", [0, 2])          # scene 1 (456 rays at 0.2 m, ~3e5 emulated lanes per launch) runs in the GPU suite only
def test_scene_matches_reference_mapper(emu_ops, golden_map, scene):
    data, meta = golden_map
    # (the array form of update_map is re-checked on the edge-case scene here and on all three on the GPU)
    vc.check_scene(emu_ops, data, meta["scenes"][scene], max_grid_cells=70_000, array_form=scene == 2)


def test_ray_walk_matches_reference(emu_ops, golden_map):
    vc.check_trace_rays(emu_ops, *golden_map)
    vc.check_trace_ray_method(emu_ops, *golden_map)


def test_random_scenes_match_oracle(emu_ops):
    vc.check_random_scenes(emu_ops, n_scenes=2, n_rays=40)


def test_edges_and_statuses(emu_ops):
    vc.check_edges(emu_ops)


def test_mapper_feeds_planner(emu_ops):
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCPlanner
    tops = Ops(TorchCpuBackend(), emu_ops.lib)

    def factory():
        pl = SE3MPCPlanner()
        pl._ops = tops
        return pl
    vc.check_mapper_planner_loop(tops, factory, steps=2)
    vc.check_restarts_filtered_by_map(tops, factory, n_restarts=12)
