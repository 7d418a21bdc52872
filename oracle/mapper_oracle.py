"""CPU oracle for the obstacle source of the SE(3) MPC path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 restatement of the arithmetic of DART-Planner's sparse voxel map,
``src/dart_planner/perception/explicit_geometric_mapper.py`` (called ``mapper.py`` below): voxel
indexing, the DDA ray walk, the Bayesian occupancy update, point / trajectory / local-grid queries
and the grid -> sphere-list selection of ``cloud/main_improved_threelayer.py:381-398`` (SURVEY.md
section 8f-2).  Plain Python loops over a dict, like the reference: this is the checker, sized for
cases that finish in seconds.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.

Pinning: ``tests/test_mapper_oracle_golden.py`` checks every function against vectors produced by
running the reference's own mapper in the build container (``tests/golden/make_golden_mapper.py`` ->
``tests/golden/mapper_map.npz``).

Reference quirks kept as they are:
* a "miss" (ray passes through) multiplies the odds by 0.6/0.4 -- it RAISES the occupancy
  (mapper.py:319-323 with prob_miss = 0.4);
* ``hit_distance`` of 0.0 counts as "no hit distance" for the ray length (``if obs.hit_distance``,
  mapper.py:111) but as a hit for the endpoint flag (``is not None``, :121-123);
* ``add_obstacle`` measures from the voxel CORNER (index * resolution), queries use floor();
* the local grid is linspace(centre - size/2, centre + size/2, int(size / resolution)) per axis, i.e. cells are
  NOT voxel centres (mapper.py:231-239).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

PROB_HIT = 0.7        # mapper.py:81-83
PROB_MISS = 0.4
PROB_PRIOR = 0.5
Key = Tuple[int, int, int]


class VoxelMap:
    """mapper.py:50-92: sparse dict of voxels; value = [occupancy probability, observation count]."""

    def __init__(self, resolution: float = 0.2, max_range: float = 50.0):
        self.resolution = float(resolution)
        self.max_range = float(max_range)
        self.voxels: Dict[Key, List[float]] = {}

    # -------------------------------------------------------------- mapper.py:93-100
    def world_to_voxel(self, position) -> Key:
        v = np.floor(np.asarray(position, float) / self.resolution).astype(int)
        return (int(v[0]), int(v[1]), int(v[2]))

    def voxel_to_world(self, key: Key) -> np.ndarray:
        return np.array(key) * self.resolution + self.resolution / 2

    # -------------------------------------------------------------- mapper.py:251-312
    def trace_ray(self, start, direction, distance: float) -> List[Key]:
        start = np.asarray(start, float)
        direction = np.asarray(direction, float)
        direction = direction / np.linalg.norm(direction)
        end = start + direction * distance
        cur = list(self.world_to_voxel(start))
        end_voxel = self.world_to_voxel(end)
        out = [tuple(cur)]
        res = self.resolution
        step = [1 if end_voxel[i] > cur[i] else -1 if end_voxel[i] < cur[i] else 0 for i in range(3)]
        t_delta = [res / abs(direction[i]) if step[i] != 0 else math.inf for i in range(3)]
        t_max = []
        for i in range(3):
            if step[i] != 0:
                boundary = (cur[i] + (1 if step[i] > 0 else 0)) * res
                t_max.append(abs((boundary - start[i]) / direction[i]))
            else:
                t_max.append(math.inf)
        total = 0.0
        while tuple(cur) != end_voxel and total <= distance:
            axis = int(np.argmin(t_max))             # first minimum on ties
            cur[axis] += step[axis]
            total = t_max[axis]
            t_max[axis] += t_delta[axis]
            out.append(tuple(cur))
        return out

    # -------------------------------------------------------------- mapper.py:314-337
    @staticmethod
    def bayes(p: float, hit: bool) -> float:
        like = PROB_HIT if hit else 1 - PROB_MISS
        num = like * p
        den = like * p + (1 - like) * (1 - p)
        if den > 0:
            p = num / den
        return float(np.clip(p, 0.01, 0.99))

    # -------------------------------------------------------------- mapper.py:102-153
    def update_map(self, origins, directions, hit_distances: Sequence[Optional[float]], max_ranges) -> int:
        """One observation per row; hit_distances[i] None = no return.  Returns the number of voxel updates."""
        n = 0
        for o, d, h, mr in zip(origins, directions, hit_distances, max_ranges):
            dist = h if h else mr
            dist = min(dist, self.max_range)
            ray = self.trace_ray(o, d, dist)
            for i, key in enumerate(ray):
                v = self.voxels.setdefault(key, [PROB_PRIOR, 0])
                endpoint = (i == len(ray) - 1) and (h is not None)
                v[0] = self.bayes(v[0], endpoint)
                v[1] += 1
                n += 1
        return n

    # -------------------------------------------------------------- mapper.py:424-447
    def add_obstacle(self, centre, radius: float) -> None:
        centre = np.asarray(centre, float)
        vc = self.world_to_voxel(centre)
        vr = int(np.ceil(radius / self.resolution))
        for dx in range(-vr, vr + 1):
            for dy in range(-vr, vr + 1):
                for dz in range(-vr, vr + 1):
                    key = (vc[0] + dx, vc[1] + dy, vc[2] + dz)
                    if np.linalg.norm(np.array(key) * self.resolution - centre) <= radius:
                        self.voxels.setdefault(key, [PROB_PRIOR, 0])[0] = 0.9

    # -------------------------------------------------------------- mapper.py:155-183
    def query(self, positions) -> np.ndarray:
        P = np.asarray(positions, float).reshape(-1, 3)
        out = np.empty(len(P))
        for i, p in enumerate(P):
            v = self.voxels.get(self.world_to_voxel(p))
            out[i] = PROB_PRIOR if v is None else v[0]
        return out

    # -------------------------------------------------------------- mapper.py:185-219, 339-353
    def is_trajectory_safe(self, positions, safety_margin: float = 1.0, threshold: float = 0.6) -> Tuple[bool, int]:
        for i, pos in enumerate(np.asarray(positions, float).reshape(-1, 3)):
            checks = [pos]
            for axis in range(3):
                for sgn in (-1, 1):
                    off = np.zeros(3)
                    off[axis] = sgn * safety_margin
                    checks.append(pos + off)
            if np.any(self.query(np.array(checks)) > threshold):
                return False, i
        return True, -1

    # -------------------------------------------------------------- mapper.py:221-248
    def local_grid(self, centre, size: float = 20.0) -> Tuple[np.ndarray, np.ndarray]:
        grid = local_grid_positions(centre, size, self.resolution)
        return grid, self.query(grid)

    def items(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(keys (V,3) int64, probabilities (V,), observation counts (V,)), sorted by key."""
        if not self.voxels:
            return np.zeros((0, 3), np.int64), np.zeros(0), np.zeros(0, np.int64)
        ks = sorted(self.voxels)
        return (np.array(ks, dtype=np.int64), np.array([self.voxels[k][0] for k in ks]),
                np.array([self.voxels[k][1] for k in ks], dtype=np.int64))


def local_grid_positions(centre, size: float, resolution: float) -> np.ndarray:
    """mapper.py:231-239: cell positions in the reference's order, (M, 3); cell m = (iz * n + ix) * n + iy."""
    centre = np.asarray(centre, float)
    half = size / 2
    n = int(size / resolution)
    x, y, z = (np.linspace(centre[a] - half, centre[a] + half, n) for a in range(3))
    return np.array(np.meshgrid(x, y, z)).T.reshape(-1, 3)


def spheres_from_occupancy(grid, occ, threshold: float = 0.6, target: int = 20, radius: float = 1.0) -> np.ndarray:
    """cloud/main_improved_threelayer.py:387-398 (target 20) / tests/test_se3_mpc_with_mapper.py:29-33 (target 10)."""
    pts = np.asarray(grid, float).reshape(-1, 3)
    occupied = pts[np.asarray(occ, float).reshape(-1) > threshold]
    if occupied.size == 0:
        return np.zeros((0, 4))
    step = max(1, occupied.shape[0] // target)
    chosen = occupied[::step]
    return np.concatenate([chosen, np.full((len(chosen), 1), radius)], axis=1)


def lidar_directions(num_rays: int) -> np.ndarray:
    """mapper.py:372-376: the horizontal fan of simulate_lidar_scan."""
    ang = np.array([2 * np.pi * i / num_rays for i in range(num_rays)])
    return np.stack([np.cos(ang), np.sin(ang), np.zeros(num_rays)], axis=1)
