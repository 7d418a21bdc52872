#!/usr/bin/env python3
"""Golden fixtures for the voxel map by RUNNING THE REFERENCE'S OWN MAPPER.

Build container only (needs /root/reference); writes tests/golden/mapper_map.npz + .json.  Same
stand-ins as make_golden.py (identity-units ``pint``); the mapper's arithmetic is the reference's own.
Observations are built here (seeded) with the structure of ``simulate_lidar_scan`` (mapper.py:361-399) plus
3-D and edge-case rays, and handed to the reference's ``update_map`` as its own ``SensorObservation``s.

Usage:  python tests/golden/make_golden_mapper.py
"""
import contextlib
import io
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.environ.get("SE3MPC_GOLDEN_OUT", HERE)      # tests/test_golden_reproducible.py writes to a scratch directory
sys.path.insert(0, HERE)
from make_golden import _install_standins  # noqa: E402


def dump_voxels(mapper):
    ks = sorted(mapper.voxels)
    return (np.array(ks, dtype=np.int64).reshape(-1, 3), np.array([mapper.voxels[k].occupancy_probability for k in ks], float),
            np.array([mapper.voxels[k].observation_count for k in ks], dtype=np.int64))


def main():
    tmp = _install_standins()
    try:
        import logging
        logging.disable(logging.CRITICAL)
        from dart_planner.perception.explicit_geometric_mapper import ExplicitGeometricMapper, SensorObservation

        def new_mapper(res, rng_):
            with contextlib.redirect_stdout(io.StringIO()):
                return ExplicitGeometricMapper(resolution=res, max_range=rng_)

        out, meta = {}, {"scenes": [], "rays": [], "numpy": np.__version__}
        rng = np.random.default_rng(20250801)

        def fan(n):
            a = np.array([2 * np.pi * i / n for i in range(n)])
            return np.stack([np.cos(a), np.sin(a), np.zeros(n)], axis=1)

        def make_obs(origin, dirs, hits, max_range):
            return [SensorObservation(position=np.array(origin, float), direction=np.array(d, float),
                                      hit_distance=(None if np.isnan(h) else float(h)), max_range=max_range, timestamp=1.0)
                    for d, h in zip(dirs, hits)]

        scenes = [
            # tag, resolution, max_range, obstacles, scans [(origin, n_rays, kind)]
            ("lidar_r05", 0.5, 40.0, [([6.0, 1.0, 2.0], 1.5)],
             [([0.0, 0.0, 2.0], 180, "fan"), ([0.4, 0.1, 2.2], 180, "fan"), ([0.9, 0.15, 2.45], 180, "fan")]),
            ("default_r02", 0.2, 50.0, [([3.0, -2.0, 1.0], 0.8), ([-2.5, 4.0, 2.0], 0.5)],
             [([0.05, -0.03, 1.5], 360, "fan"), ([0.05, -0.03, 1.5], 96, "sphere")]),
            ("edge_cases", 0.25, 12.0, [], [([-1.3, 2.6, 0.7], 24, "edges")]),
        ]
        for tag, res, mr, obstacles, scans in scenes:
            m = new_mapper(res, mr)
            for c, r in obstacles:
                m.add_obstacle(np.array(c, float), r)
            k = f"s_{tag}_"
            out[k + "obst_keys"], out[k + "obst_prob"], _ = dump_voxels(m)
            sc_meta = []
            for j, (origin, n, kind) in enumerate(scans):
                if kind == "fan":
                    dirs = fan(n)
                    hits = np.where(rng.random(n) < 0.1, rng.uniform(2.0, 20.0, n), np.nan)
                elif kind == "sphere":
                    dirs = rng.normal(size=(n, 3)) * [1.0, 1.0, 0.6]          # not normalised: update_map normalises
                    hits = np.where(rng.random(n) < 0.4, rng.uniform(0.5, 30.0, n), np.nan)
                else:
                    dirs = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                                     [1, 1, 0], [1, -1, 0], [-1, -1, 1], [1, 1, 1], [2, 0.5, -0.25], [-0.3, 0.1, 0.05]] * 2, float)
                    hits = np.array([3.0, 0.0, np.nan, 0.1, 100.0, 0.24, 5.0, np.nan, 7.5, 0.0, 11.99, 12.0,
                                     np.nan, 2.5, 0.3, np.nan, 1.0, 50.0, 0.26, 4.0, np.nan, 3.3, np.nan, 6.0])
                obs_mr = mr if kind != "edges" else 20.0                         # observation range above the map range
                res_upd = m.update_map(make_obs(origin, dirs, hits, obs_mr))
                kk = f"{k}scan{j}_"
                out[kk + "origin"] = np.tile(np.array(origin, float), (n, 1))
                out[kk + "dirs"], out[kk + "hits"] = dirs, hits
                out[kk + "keys"], out[kk + "prob"], out[kk + "count"] = dump_voxels(m)
                sc_meta.append(dict(key=kk, n_rays=n, kind=kind, obs_max_range=obs_mr, updated_voxels=int(res_upd["updated_voxels"]),
                                    total_voxels=int(res_upd["total_voxels"])))
            # point queries (mapper.py:155-183): inside, outside, negative coordinates, exact voxel boundaries
            q = np.concatenate([rng.uniform(-12, 12, (300, 3)), np.round(rng.uniform(-8, 8, (60, 3)) / res) * res,
                                np.array([[0.0, 0.0, 0.0], [-res, res, -0.0], [1e-9, -1e-9, 1.0]])])
            out[k + "query_pos"] = q
            out[k + "query_occ"] = m.query_occupancy_batch(q)
            # is_trajectory_safe (mapper.py:185-219)
            trajs, safe = [], []
            for t in range(24):
                a, b = rng.uniform(-9, 9, 3), rng.uniform(-9, 9, 3)
                n_pts = 30
                P = a + (b - a) * np.linspace(0, 1, n_pts)[:, None]
                margin, thr = (1.0, 0.6) if t % 2 == 0 else (0.5, 0.55)
                ok, first = m.is_trajectory_safe(P, safety_margin=margin, threshold=thr)
                trajs.append(P); safe.append([int(ok), int(first), margin, thr])
            out[k + "traj_P"] = np.array(trajs)
            out[k + "traj_safe"] = np.array(safe, float)
            # local grid -> spheres (mapper.py:221-248 + cloud/main_improved_threelayer.py:387-398)
            grids = []
            for gi, (centre, size, target) in enumerate([(scans[0][0], 20.0, 20), ([1.0, -1.5, 2.0], 10.0, 10), ([30.0, 30.0, 30.0], 6.0, 20)]):
                if res < 0.25 and size > 10.0:
                    size = 12.0                                                   # keep the Python dict walk to ~2e5 cells
                grid, occ = m.get_local_occupancy_grid(np.array(centre, float), size=size)
                pts = grid[occ > 0.6]
                step = max(1, pts.shape[0] // target)
                chosen = pts[::step] if pts.size else np.zeros((0, 3))
                gk = f"{k}grid{gi}_"
                out[gk + "spheres"] = np.concatenate([chosen, np.ones((len(chosen), 1))], axis=1)
                n = grid.shape[0]
                out[gk + "x"], out[gk + "y"], out[gk + "z"] = grid[0, :, 0, 0].copy(), grid[:, 0, 0, 1].copy(), grid[0, 0, :, 2].copy()
                out[gk + "occ_where"] = np.flatnonzero(occ.reshape(-1) != 0.5).astype(np.int64)
                out[gk + "occ_vals"] = occ.reshape(-1)[out[gk + "occ_where"]]
                grids.append(dict(key=gk, centre=list(map(float, centre)), size=size, target=target, num_cells=int(n),
                                  n_occupied=int(pts.shape[0]), n_spheres=int(len(chosen))))
            meta["scenes"].append(dict(key=k, tag=tag, resolution=res, max_range=mr, obstacles=obstacles, scans=sc_meta, grids=grids))
        # _trace_ray (mapper.py:251-312), incl. the reference's own test ray (tests/test_mapper_trace_ray.py:8-12)
        rays = [(0.5, [0.0, 0.0, 0.0], (np.array([1.0, 1.0, 0.0]) / np.sqrt(2)).tolist(), 5.0)]
        for _ in range(40):
            res = float(rng.choice([0.2, 0.25, 0.5]))
            rays.append((res, rng.uniform(-5, 5, 3).tolist(), rng.normal(size=3).tolist(), float(rng.uniform(0.05, 15))))
        rays += [(0.5, [0.25, 0.25, 0.25], [1.0, 0.0, 0.0], 3.0), (0.5, [0.0, 0.0, 0.0], [-1.0, -1.0, -1.0], 4.0),
                 (0.2, [1.0, 1.0, 1.0], [0.0, 0.0, 1.0], 0.0), (0.25, [-0.5, 0.5, 0.0], [1.0, -1.0, 0.0], 2.0)]
        for i, (res, start, d, dist) in enumerate(rays):
            m = new_mapper(res, 50.0)
            vox = m._trace_ray(np.array(start, float), np.array(d, float), dist)
            out[f"r{i:02d}_voxels"] = np.array(vox, dtype=np.int64).reshape(-1, 3)
            meta["rays"].append(dict(key=f"r{i:02d}_", resolution=res, start=start, direction=d, distance=dist, n=len(vox)))
        np.savez_compressed(os.path.join(OUT_DIR, "mapper_map.npz"), **out)
        with open(os.path.join(OUT_DIR, "mapper_map.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("wrote mapper_map.npz", os.path.getsize(os.path.join(OUT_DIR, "mapper_map.npz")), "bytes;", len(out), "arrays")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
