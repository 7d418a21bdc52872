"""Shared parity checks of the device voxel map (dart_planner_amd/csrc/voxel_map.hip through the C ABI and the
reference-shaped ExplicitGeometricMapper on top of it) against the reference mapper's own outputs
(tests/golden/mapper_map.npz) and the oracle.  Run by tests/test_emu_voxel.py (host emulation) and
tests/test_gpu_voxel.py (MI355X).  Everything is bit-exact: integer keys and counts, float64 probabilities."""
import numpy as np

from oracle import mapper_oracle as mo
from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper, SensorObservation


def observations(data, scan):
    k = scan["key"]
    return [SensorObservation(position=o, direction=d, hit_distance=(None if np.isnan(h) else float(h)),
                              max_range=scan["obs_max_range"], timestamp=1.0)
            for o, d, h in zip(data[k + "origin"], data[k + "dirs"], data[k + "hits"])]


def check_scene(ops, data, sc, capacity=64, max_grid_cells=None, array_form=True):
    """One golden scene end to end: add_obstacle, update_map scan by scan, queries, trajectory safety, local grids."""
    k = sc["key"]
    m = ExplicitGeometricMapper(resolution=sc["resolution"], max_range=sc["max_range"], capacity=capacity, ops=ops)
    for c, r in sc["obstacles"]:
        m.add_obstacle(np.array(c, float), r)
    keys, prob, cnt = m.map.items()
    assert np.array_equal(keys, data[k + "obst_keys"]) and np.array_equal(prob, data[k + "obst_prob"]) and not cnt.any()
    for s in sc["scans"]:
        res = m.update_map(observations(data, s))
        assert res["updated_voxels"] == s["updated_voxels"] and res["total_voxels"] == s["total_voxels"], s["key"]
        keys, prob, cnt = m.map.items()
        assert np.array_equal(keys, data[s["key"] + "keys"]), s["key"]
        assert np.array_equal(cnt, data[s["key"] + "count"]), s["key"]
        assert np.array_equal(prob, data[s["key"] + "prob"]), s["key"]                 # bit-exact float64
    assert m.get_mapping_stats()["total_voxels"] == len(data[sc["scans"][-1]["key"] + "keys"])
    # the array form of update_map gives the same map
    m2 = None if not array_form else ExplicitGeometricMapper(resolution=sc["resolution"], max_range=sc["max_range"], capacity=capacity, ops=ops)
    if m2 is not None:
        for c, r in sc["obstacles"]:
            m2.add_obstacle(np.array(c, float), r)
        for s in sc["scans"]:
            res2 = m2.update_map_arrays(data[s["key"] + "origin"], data[s["key"] + "dirs"], data[s["key"] + "hits"], s["obs_max_range"])
            assert res2["updated_voxels"] == s["updated_voxels"]
        for a, b in zip(m2.map.items(), m.map.items()):
            assert np.array_equal(a, b)
    # point queries
    q = data[k + "query_pos"]
    assert np.array_equal(m.query_occupancy_batch(q), data[k + "query_occ"])
    assert m.query_occupancy(q[0]) == data[k + "query_occ"][0]
    assert m.is_collision(q[1], 0.6) == bool(data[k + "query_occ"][1] > 0.6)
    # trajectory safety: the reference call shape, then all of them as one batch per (margin, threshold)
    P, S = data[k + "traj_P"], data[k + "traj_safe"]
    for i in (0, 1, len(P) - 1):
        assert m.is_trajectory_safe(P[i], safety_margin=S[i, 2], threshold=S[i, 3]) == (bool(S[i, 0]), int(S[i, 1]))
    for margin, thr in {(a, b) for a, b in S[:, 2:]}:
        sel = np.flatnonzero((S[:, 2] == margin) & (S[:, 3] == thr))
        safe, first = m.trajectories_safe(P[sel], margin, thr)
        assert np.array_equal(safe, S[sel, 0].astype(bool)) and np.array_equal(first, S[sel, 1].astype(np.int64))
    # packed rows (the solver's [P | V | T] layout: stride 9N) and float32 positions (vs the oracle on the rounded values)
    N = P.shape[1]
    packed = np.concatenate([P.reshape(len(P), -1), np.zeros((len(P), 6 * N))], axis=1)
    safe, first = m.trajectories_safe(packed, 1.0, 0.6, n_steps=N, stride=9 * N)
    oracle_map = mo.VoxelMap(sc["resolution"], sc["max_range"])
    kk, pp, cc = m.map.items()
    oracle_map.voxels = {tuple(int(v) for v in key): [float(p_), int(c_)] for key, p_, c_ in zip(kk, pp, cc)}
    exp = [oracle_map.is_trajectory_safe(p_, 1.0, 0.6) for p_ in P]
    assert np.array_equal(safe, [e[0] for e in exp]) and np.array_equal(first, [e[1] for e in exp])
    P32 = P.astype(np.float32)
    if hasattr(ops.be, "torch"):
        dP32 = ops.be.from_host(P32)
    else:
        dP32 = P32
    safe32, first32 = m.trajectories_safe(dP32, 1.0, 0.6)
    exp32 = [oracle_map.is_trajectory_safe(p_.astype(np.float64), 1.0, 0.6) for p_ in P32]
    assert np.array_equal(safe32, [e[0] for e in exp32]) and np.array_equal(first32, [e[1] for e in exp32])
    # local grids: the reference call (grid + occupancy) and the fused sphere list
    for g in sc["grids"]:
        n = g["num_cells"]
        if max_grid_cells is not None and n ** 3 > max_grid_cells:
            continue                                    # host emulation: ballots are slow; the GPU suite sweeps every grid
        sp = m.local_obstacle_spheres(g["centre"], g["size"], 0.6, g["target"], 1.0)
        assert np.array_equal(sp, data[g["key"] + "spheres"]), g["key"]
        sp32, cnt32 = m.map.local_spheres(g["centre"], g["size"], 0.6, g["target"], 1.0, precision="f32")
        c2 = ops.be.to_host(cnt32)
        assert int(c2[0]) == g["n_spheres"] and int(c2[1]) == g["n_occupied"]
        assert np.array_equal(np.asarray(ops.be.to_host(sp32))[:g["n_spheres"]], data[g["key"] + "spheres"].astype(np.float32))
        if n ** 3 <= 300_000:
            grid, occ = m.get_local_occupancy_grid(np.array(g["centre"], float), g["size"])
            assert grid.shape == (n, n, n, 3) and np.array_equal(grid[0, :, 0, 0], data[g["key"] + "x"])
            flat = occ.reshape(-1)
            where = np.flatnonzero(flat != 0.5)
            assert np.array_equal(where, data[g["key"] + "occ_where"]) and np.array_equal(flat[where], data[g["key"] + "occ_vals"])
    return m


def check_trace_ray_method(ops, data, meta):
    """_trace_ray itself (the method the reference's tests/test_mapper_trace_ray.py calls), every golden ray in walk
    order, and that test's own assertion (6-connected, no gaps)."""
    for r in meta["rays"]:
        m = ExplicitGeometricMapper(resolution=r["resolution"], max_range=50.0, capacity=64, ops=ops)
        vox = m._trace_ray(np.array(r["start"], float), np.array(r["direction"], float), r["distance"])
        assert np.array_equal(np.array(vox, dtype=np.int64).reshape(-1, 3), data[r["key"] + "voxels"]), r
    m = ExplicitGeometricMapper(resolution=0.5, ops=ops)
    voxels = m._trace_ray(np.array([0.0, 0.0, 0.0]), np.array([1.0, 1.0, 0.0]) / np.sqrt(2), distance=5.0)
    diffs = [np.sum(np.abs(np.subtract(v2, v1))) for v1, v2 in zip(voxels[:-1], voxels[1:])]
    assert all(d == 1 for d in diffs)


def check_trace_rays(ops, data, meta):
    """_trace_ray (mapper.py:251-312) through update_map on an empty map: one pass-through ray marks exactly the
    voxels the reference's walk returns, each once."""
    for r in meta["rays"]:
        m = ExplicitGeometricMapper(resolution=r["resolution"], max_range=1e9, capacity=64, ops=ops)
        res = m.update_map([SensorObservation(position=np.array(r["start"], float), direction=np.array(r["direction"], float),
                                              hit_distance=None, max_range=r["distance"])] if r["distance"] > 0 else [])
        if r["distance"] <= 0:
            continue                     # `hit_distance or max_range` cannot express a zero-length ray through update_map
        keys, prob, cnt = m.map.items()
        exp = np.unique(data[r["key"] + "voxels"], axis=0)
        assert res["updated_voxels"] == r["n"] and np.array_equal(keys, exp), r
        assert np.all(cnt == 1) and np.all(prob == mo.VoxelMap.bayes(0.5, False))


def check_edges(ops):
    """Empty map, empty inputs, growth, error statuses."""
    from dart_planner_amd.capi import VoxelMapDesc
    m = ExplicitGeometricMapper(resolution=0.5, max_range=10.0, capacity=64, ops=ops)
    assert len(m.map) == 0 and m.query_occupancy([1.0, 2.0, 3.0]) == 0.5
    assert m.is_trajectory_safe(np.zeros((0, 3))) == (True, -1)
    assert m.local_obstacle_spheres([0, 0, 0], 5.0).shape == (0, 4)
    assert m.local_obstacle_spheres([0, 0, 0], 0.2).shape == (0, 4)            # int(size / resolution) == 0 cells
    assert m.update_map([])["updated_voxels"] == 0
    # far outside the packable index range: unknown space, not an error
    assert m.query_occupancy([1e9, 0.0, 0.0]) == 0.5 and m.query_occupancy([np.nan, 0.0, 0.0]) == 0.5
    # growth: 5000 voxels into a 64-slot table
    ijk = np.stack(np.meshgrid(np.arange(-10, 10), np.arange(0, 25), np.arange(-5, 5), indexing="ij"), -1).reshape(-1, 3)
    m.map.insert(ijk, prob=np.linspace(0.0, 1.0, len(ijk)))
    keys, prob, _ = m.map.items()
    assert len(keys) == len(ijk) == len(m.map) and m.map.capacity * 85 >= 100 * len(ijk)
    order = np.lexsort((ijk[:, 2], ijk[:, 1], ijk[:, 0]))
    assert np.array_equal(keys, ijk[order]) and np.array_equal(prob, np.linspace(0.0, 1.0, len(ijk))[order])
    # helpers of the reference's API
    from dart_planner_amd.perception.explicit_geometric_mapper import VoxelData
    v = VoxelData()
    m._bayesian_update(v, hit=False)
    assert v.occupancy_probability == mo.VoxelMap.bayes(0.5, False)
    pts = m._get_safety_margin_positions(np.array([1.0, 2.0, 3.0]), 0.5)
    assert len(pts) == 7 and np.array_equal(pts[1], [0.5, 2.0, 3.0]) and np.array_equal(pts[6], [1.0, 2.0, 3.5])
    # C-ABI statuses
    lib = ops.lib
    assert lib.voxel_status("clear", None, 0) == -1                                           # SE3MPC_ERR_NULL
    bad = VoxelMapDesc(keys=m.map.desc.keys, prob=m.map.desc.prob, count=m.map.desc.count, capacity=100, reserved=0,
                       resolution=0.5, prior=0.5)
    assert lib.voxel_status("clear", bad, 0) == -3                                            # capacity not a power of two
    bad.capacity, bad.resolution = 128, 0.0
    assert lib.voxel_status("clear", bad, 0) == -4                                            # SE3MPC_ERR_PARAM
    assert lib.voxel_status("query_f64", m.map.desc, 0, -1, 0, 0) == -3 and lib.voxel_status("query_f64", m.map.desc, 0, 0, 0, 0) == 0
    assert lib.voxel_local_workspace(100) >= 100 ** 3 // 1024


def check_mapper_planner_loop(ops, planner_factory, steps=3):
    """The reference's integration test (tests/test_se3_mpc_with_mapper.py) restated: simulated LiDAR -> update_map ->
    local sphere list -> planner obstacles -> plan.  (Its last line reads `planner.config.dt`; in the reference, as
    here, `planner.config` is the BasePlanner dict, so the time step is taken from `se3_config`.)"""
    from dart_planner_amd.common.types import DroneState
    planner = planner_factory()
    mapper = ExplicitGeometricMapper(resolution=0.5, max_range=40.0, ops=ops)
    state = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 2.0]), velocity=np.zeros(3), attitude=np.zeros(3),
                       angular_velocity=np.zeros(3))
    goal = np.array([10.0, 0.0, 5.0])
    np.random.seed(7)
    for _ in range(steps):
        mapper.update_map(mapper.simulate_lidar_scan(state, num_rays=180))
        grid, occ = mapper.get_local_occupancy_grid(state.position, size=15.0)
        occupied = grid[occ > 0.6]
        fused = mapper.local_obstacle_spheres(state.position, size=15.0, target=10)
        ref_sel = occupied[:: max(1, len(occupied) // 10)]
        assert np.array_equal(fused[:, :3], ref_sel)
        planner.clear_obstacles()
        for p in fused:
            planner.add_obstacle(p[:3], radius=1.0)
        traj = planner.plan_trajectory(state, goal)
        assert traj is not None and len(traj.positions) > 0
        state.position = state.position + 0.3 * (traj.positions[1] - state.position)
        state.timestamp += planner.se3_config.dt
    return mapper


def check_restarts_filtered_by_map(ops, planner_factory, n_restarts=48):
    """plan_with_restarts(mapper=...): the winner is the lowest-objective restart among those the map calls safe."""
    from dart_planner_amd.common.types import DroneState
    planner = planner_factory()
    mapper = ExplicitGeometricMapper(resolution=0.5, max_range=40.0, ops=ops)
    state = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 2.0]), velocity=np.zeros(3), attitude=np.zeros(3),
                       angular_velocity=np.zeros(3))
    goal = np.array([6.0, 0.5, 2.5])
    free = planner.plan_with_restarts(state, goal, n_restarts=n_restarts, sigma=2.0, seed=1, mapper=mapper)
    assert planner.last_result["n_safe"] == n_restarts                       # empty map: everything is safe
    unfiltered = planner.plan_with_restarts(state, goal, n_restarts=n_restarts, sigma=2.0, seed=1)
    assert np.array_equal(free.positions, unfiltered.positions)
    # occupied voxels over the first planned position (the reference's objective decouples the position block from the
    # thrust block, so every restart plans the same positions): nothing passes, the lowest objective still comes back
    mapper.add_obstacle(np.array(unfiltered.positions[0], float), 1.5)
    blocked = planner.plan_with_restarts(state, goal, n_restarts=n_restarts, sigma=2.0, seed=1, mapper=mapper)
    assert planner.last_result["n_safe"] == 0 and np.array_equal(blocked.positions, unfiltered.positions)
    # the selection rule itself, against the per-restart answers of the reference-shaped call
    m2 = ExplicitGeometricMapper(resolution=0.5, max_range=40.0, ops=ops)
    m2.add_obstacle(unfiltered.positions[-1] + np.array([0.0, 0.0, 0.9]), 0.5)     # clips the unfiltered winner's last step
    res = planner.plan_batch(np.tile(state.position, (3, 1)), np.zeros((3, 3)), np.tile(goal, (3, 1)), precision="f64")
    for P in res["positions"]:
        ok, first = m2.is_trajectory_safe(P, 1.0, 0.6)
        s_b, f_b = m2.trajectories_safe(P[None], 1.0, 0.6)
        assert (ok, first) == (bool(s_b[0]), int(f_b[0]))


def check_random_scenes(ops, n_scenes=12, n_rays=150, seed=0):
    """Random maps against the oracle's dict walk, bit for bit: non-dyadic resolutions (0.1 .. 0.9 m), 3-D rays from
    random origins incl. axis-aligned and diagonal ones, hit distances from 0 to beyond the range, two scans per
    scene; then random point queries and trajectory checks on the result."""
    rng = np.random.default_rng(seed)
    for sc in range(n_scenes):
        res = float(rng.choice([0.1, 0.17, 0.2, 0.3, 0.37, 0.5, 0.9]))
        max_range = float(rng.uniform(4.0, 15.0))
        m = ExplicitGeometricMapper(resolution=res, max_range=max_range, capacity=64, ops=ops)
        o = mo.VoxelMap(res, max_range)
        for _ in range(int(rng.integers(0, 3))):
            c, r = rng.uniform(-4, 4, 3), float(rng.uniform(0.2, 1.0))
            m.add_obstacle(c, r); o.add_obstacle(c, r)
        for scan in range(2):
            origin = rng.uniform(-3, 3, 3)
            dirs = rng.normal(size=(n_rays, 3))
            special = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [1, 1, 0], [1, -1, 1], [-1, -1, -1]], float)
            dirs[:len(special)] = special
            hits = np.where(rng.random(n_rays) < 0.5, rng.uniform(0.0, 1.5 * max_range, n_rays), np.nan)
            hits[:2] = [0.0, np.nan]
            obs = [SensorObservation(position=origin.copy(), direction=d, hit_distance=(None if np.isnan(h) else float(h)),
                                     max_range=max_range * 1.2, timestamp=0.0) for d, h in zip(dirs, hits)]
            res_dev = m.update_map(obs)
            n_or = o.update_map([origin] * n_rays, dirs, [None if np.isnan(h) else float(h) for h in hits], [max_range * 1.2] * n_rays)
            assert res_dev["updated_voxels"] == n_or and res_dev["total_voxels"] == len(o.voxels), (sc, scan)
        for a, b in zip(m.map.items(), o.items()):
            assert np.array_equal(a, b), (sc, res)
        q = rng.uniform(-12, 12, (500, 3))
        assert np.array_equal(m.query_occupancy_batch(q), o.query(q))
        P = rng.uniform(-6, 6, (16, 12, 3))
        safe, first = m.trajectories_safe(P, 0.7, 0.6)
        exp = [o.is_trajectory_safe(p, 0.7, 0.6) for p in P]
        assert np.array_equal(safe, [e[0] for e in exp]) and np.array_equal(first, [e[1] for e in exp])
