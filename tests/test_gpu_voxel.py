"""GPU suite (`-m gpu`) for the voxel map: dart_planner_amd/csrc/voxel_map.hip on a real MI355X through the C ABI and
the product's host classes, bit-exact against the reference mapper's golden outputs (tests/golden/mapper_map.npz),
plus properties at the reference's full local-grid size (20 m at 0.2 m = 10^6 cells)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import voxel_checks as vc  # noqa: E402
from oracle import mapper_oracle as mo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gpu_ops():
    import torch
    assert torch.cuda.is_available(), "the gpu suite needs an MI355X"
    from dart_planner_amd.ops import Ops, TorchBackend
    ops = Ops(TorchBackend("cuda:0"))
    assert os.path.basename(ops.lib.path) == "libse3mpc.so"
    return ops


@pytest.fixture(scope="module")
def golden_map():
    return (np.load(os.path.join(HERE, "golden", "mapper_map.npz")), json.load(open(os.path.join(HERE, "golden", "mapper_map.json"))))


@pytest.mark.parametrize("scene", [0, 1, 2])
def test_scene_matches_reference_mapper(gpu_ops, golden_map, scene):
    data, meta = golden_map
    vc.check_scene(gpu_ops, data, meta["scenes"][scene])


def test_ray_walk_matches_reference(gpu_ops, golden_map):
    vc.check_trace_rays(gpu_ops, *golden_map)
    vc.check_trace_ray_method(gpu_ops, *golden_map)


def test_random_scenes_match_oracle(gpu_ops):
    vc.check_random_scenes(gpu_ops, n_scenes=16, n_rays=200)


def test_edges_and_statuses(gpu_ops):
    vc.check_edges(gpu_ops)


def test_mapper_feeds_planner(gpu_ops):
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCPlanner
    vc.check_mapper_planner_loop(gpu_ops, lambda: SE3MPCPlanner(), steps=5)
    vc.check_restarts_filtered_by_map(gpu_ops, lambda: SE3MPCPlanner(precision="f64"), n_restarts=256)


def test_update_is_order_exact_under_repetition(gpu_ops, golden_map):
    """The same scans applied twice in a row to two maps give identical tables (the ray-after-ray update has no
    race), and a map rebuilt from an exported snapshot answers every query identically."""
    from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper
    data, meta = golden_map
    sc = meta["scenes"][1]
    maps = []
    for _ in range(2):
        m = ExplicitGeometricMapper(resolution=sc["resolution"], max_range=sc["max_range"], ops=gpu_ops)
        for s in sc["scans"]:
            m.update_map(vc.observations(data, s))
            m.update_map(vc.observations(data, s))
        maps.append(m.map.items())
    for a, b in zip(*maps):
        assert np.array_equal(a, b)
    clone = ExplicitGeometricMapper(resolution=sc["resolution"], max_range=sc["max_range"], ops=gpu_ops)
    clone.map.insert(maps[0][0], prob=maps[0][1], counts=maps[0][2])
    q = np.random.default_rng(3).uniform(-15, 15, (20000, 3))
    assert np.array_equal(clone.query_occupancy_batch(q), m.query_occupancy_batch(q))


def test_full_size_local_grid(gpu_ops):
    """The reference's default sweep (cloud/main_improved_threelayer.py:387: size 20 m at the default 0.2 m resolution
    = 100^3 cells) -- too slow to produce with the reference's dict walk inside the golden generator, so checked through
    properties: the fused sphere list equals the selection applied to the device's own occupancy of the same cells, the
    occupancy equals the oracle's on a random sample of cells, and the count is consistent."""
    from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper
    m = ExplicitGeometricMapper(resolution=0.2, max_range=50.0, ops=gpu_ops)
    oracle_map = mo.VoxelMap(0.2, 50.0)
    rng = np.random.default_rng(11)
    for _ in range(6):
        c, r = rng.uniform(-8, 8, 3) + [0, 0, 2], float(rng.uniform(0.4, 1.6))
        m.add_obstacle(c, r)
        oracle_map.add_obstacle(c, r)
    k, p, _ = m.map.items()
    ko, po, _ = oracle_map.items()
    assert np.array_equal(k, ko) and np.array_equal(p, po)
    centre = np.array([0.3, -0.2, 2.1])
    grid, occ = m.get_local_occupancy_grid(centre, size=20.0)
    assert occ.shape == (100, 100, 100)
    flat, pts = occ.reshape(-1), grid.reshape(-1, 3)
    sample = rng.choice(len(flat), 3000, replace=False)
    assert np.array_equal(flat[sample], oracle_map.query(pts[sample]))
    for target in (20, 10, 1000):
        fused = m.local_obstacle_spheres(centre, 20.0, 0.6, target, 1.0)
        assert np.array_equal(fused, mo.spheres_from_occupancy(pts, flat, 0.6, target, 1.0))
    assert int((flat > 0.6).sum()) > 1000
