"""The mapper oracle (oracle/mapper_oracle.py) against what the reference's own mapper produced
(tests/golden/mapper_map.npz, written by tests/golden/make_golden_mapper.py).  Bit-exact: integer voxel
keys and counts, and float64 probabilities that went through the same IEEE operations."""
import json
import os

import numpy as np
import pytest

from oracle import mapper_oracle as mo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden_map():
    data = np.load(os.path.join(HERE, "golden", "mapper_map.npz"))
    meta = json.load(open(os.path.join(HERE, "golden", "mapper_map.json")))
    return data, meta


def replay_scene(data, sc, upto=None):
    """Rebuild the oracle map of a golden scene: obstacles, then scans 0..upto."""
    m = mo.VoxelMap(sc["resolution"], sc["max_range"])
    for c, r in sc["obstacles"]:
        m.add_obstacle(c, r)
    states = [m.items()]
    for s in sc["scans"][:upto]:
        k = s["key"]
        hits = [None if np.isnan(h) else float(h) for h in data[k + "hits"]]
        n = m.update_map(data[k + "origin"], data[k + "dirs"], hits, [s["obs_max_range"]] * len(hits))
        assert n == s["updated_voxels"] and len(m.voxels) == s["total_voxels"]
        states.append(m.items())
    return m, states


def test_trace_ray_matches_reference(golden_map):
    data, meta = golden_map
    for r in meta["rays"]:
        vox = mo.VoxelMap(r["resolution"]).trace_ray(r["start"], r["direction"], r["distance"])
        assert np.array_equal(np.array(vox, dtype=np.int64).reshape(-1, 3), data[r["key"] + "voxels"]), r
    # the reference's own assertion (tests/test_mapper_trace_ray.py:6-17): 6-connected, no gaps
    v = np.array(mo.VoxelMap(0.5).trace_ray([0, 0, 0], np.array([1.0, 1.0, 0.0]) / np.sqrt(2), 5.0))
    assert np.all(np.abs(np.diff(v, axis=0)).sum(1) == 1)


def test_update_query_safety_grid_match_reference(golden_map):
    data, meta = golden_map
    for sc in meta["scenes"]:
        k = sc["key"]
        m, states = replay_scene(data, sc)
        keys0, prob0, _ = states[0]
        assert np.array_equal(keys0, data[k + "obst_keys"]) and np.array_equal(prob0, data[k + "obst_prob"])
        for s, (keys, prob, cnt) in zip(sc["scans"], states[1:]):
            assert np.array_equal(keys, data[s["key"] + "keys"]), s["key"]
            assert np.array_equal(prob, data[s["key"] + "prob"]), s["key"]          # bit-exact float64
            assert np.array_equal(cnt, data[s["key"] + "count"]), s["key"]
        assert np.array_equal(m.query(data[k + "query_pos"]), data[k + "query_occ"])
        for P, (ok, first, margin, thr) in zip(data[k + "traj_P"], data[k + "traj_safe"]):
            assert m.is_trajectory_safe(P, margin, thr) == (bool(ok), int(first))
        for g in sc["grids"]:
            grid, occ = m.local_grid(g["centre"], g["size"])
            n = g["num_cells"]
            G = grid.reshape(n, n, n, 3)
            assert np.array_equal(G[0, :, 0, 0], data[g["key"] + "x"]) and np.array_equal(G[:, 0, 0, 1], data[g["key"] + "y"])
            assert np.array_equal(G[0, 0, :, 2], data[g["key"] + "z"])
            where = np.flatnonzero(occ != 0.5)
            assert np.array_equal(where, data[g["key"] + "occ_where"]) and np.array_equal(occ[where], data[g["key"] + "occ_vals"])
            sp = mo.spheres_from_occupancy(grid, occ, 0.6, g["target"], 1.0)
            assert np.array_equal(sp, data[g["key"] + "spheres"]) and len(sp) == g["n_spheres"]


def test_reference_quirks_are_kept():
    assert mo.VoxelMap.bayes(0.5, hit=False) == pytest.approx(0.6)       # a pass-through RAISES occupancy (mapper.py:319-323)
    assert mo.VoxelMap.bayes(0.99, hit=True) == 0.99 and mo.VoxelMap.bayes(0.0, hit=False) == 0.01
    m = mo.VoxelMap(0.5, 10.0)
    m.update_map([[0.1, 0.1, 0.1]], [[1.0, 0.0, 0.0]], [0.0], [4.0])       # hit distance 0.0: full-length ray, endpoint = hit
    keys, prob, cnt = m.items()
    assert len(keys) == 9 and prob[-1] == pytest.approx(0.7) and np.all(prob[:-1] == pytest.approx(0.6))
