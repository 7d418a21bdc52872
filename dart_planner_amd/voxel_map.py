"""Device-resident voxel map: shape-checked calls into the ``se3mpc_voxel_*`` entry points of libse3mpc
(include/se3mpc.h, dart_planner_amd/csrc/voxel_map.hip) -- the obstacle source of the SE(3) MPC path
(SURVEY.md section 8f-2).  Counterpart of the dict inside the reference's ExplicitGeometricMapper
(src/dart_planner/perception/explicit_geometric_mapper.py:74-76); the reference-shaped class on top of it is
``dart_planner_amd.perception.explicit_geometric_mapper.ExplicitGeometricMapper``.

The table (packed keys, float64 probabilities, int32 observation counts) lives in HBM as three tensors of the
array backend; it is re-hashed into a larger one before an operation that could fill it (see ``reserve``).
There is no CPU path: without the HIP library every call raises."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from .capi import VoxelMapDesc
from .ops import Ops



class DeviceVoxelMap:
    def __init__(self, resolution: float = 0.2, prior: float = 0.5, capacity: int = 1 << 14, ops: Optional[Ops] = None):
        if not resolution > 0:
            raise ValueError("resolution must be positive")
        self.ops = ops if ops is not None else Ops()
        self.be, self.lib = self.ops.be, self.ops.lib
        self.resolution, self.prior = float(resolution), float(prior)
        self._n: Optional[int] = 0
        self._alloc(self._pow2(max(64, capacity)))
        self.clear()

    # ------------------------------------------------------------------ table
    @staticmethod
    def _pow2(n: int) -> int:
        return 1 << max(6, int(n - 1).bit_length())

    def _alloc(self, capacity: int) -> None:
        self.capacity = capacity
        self.keys = self.be.empty((capacity,), "i64")
        self.prob = self.be.empty((capacity,), "f64")
        self.count = self.be.empty((capacity,), "i32")
        self._scratch = self.be.empty((8,), "i32")
        self._slot_rows = self._zeros((capacity,), "i32")       # update_rays workspace: zero between calls
        self._row_bits = None
        self.desc = VoxelMapDesc(keys=self.be.ptr(self.keys), prob=self.be.ptr(self.prob), count=self.be.ptr(self.count),
                                 capacity=capacity, reserved=0, resolution=self.resolution, prior=self.prior)

    def clear(self) -> None:
        self.lib.voxel("clear", self.desc, self.be.stream())
        self._n = 0

    def __len__(self) -> int:
        """Number of stored voxels (synchronises when it is not known from the last operation)."""
        if self._n is None:
            self.lib.voxel("export", self.desc, 0, 0, 0, self.be.ptr(self._scratch), self.be.stream())
            self._n = int(self.be.to_host(self._scratch)[0])
        return self._n

    def items(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(indices (V,3) int64, probabilities (V,), observation counts (V,)) sorted by index -- a host snapshot."""
        ijk = self.be.empty((self.capacity, 3), "i32")
        prob = self.be.empty((self.capacity,), "f64")
        cnt = self.be.empty((self.capacity,), "i32")
        self.lib.voxel("export", self.desc, self.be.ptr(ijk), self.be.ptr(prob), self.be.ptr(cnt), self.be.ptr(self._scratch),
                       self.be.stream())
        n = self._n = int(self.be.to_host(self._scratch)[0])
        k = np.array(self.be.to_host(ijk)[:n], dtype=np.int64).reshape(-1, 3)
        order = np.lexsort((k[:, 2], k[:, 1], k[:, 0])) if n else np.zeros(0, dtype=np.int64)
        return k[order], np.array(self.be.to_host(prob)[:n])[order], np.array(self.be.to_host(cnt)[:n], dtype=np.int64)[order]

    def reserve(self, extra: int) -> None:
        """Make sure `extra` more voxels cannot overflow the table (voxels + extra <= 85 % of the capacity); when it
        has to grow, re-hash into a table twice that bound, so a steady stream of same-sized updates re-hashes rarely."""
        need = len(self) + int(extra)
        if need * 100 <= self.capacity * 85:
            return
        k, p, c = self.items()
        self._alloc(self._pow2(2 * need))
        self.lib.voxel("clear", self.desc, self.be.stream())
        self._n = 0
        if len(k):
            self._insert(k, p, None, c)

    def _insert(self, ijk: np.ndarray, prob: Optional[np.ndarray], value: Optional[float], counts: Optional[np.ndarray]) -> None:
        ijk = np.ascontiguousarray(np.asarray(ijk).reshape(-1, 3))
        if ijk.size and np.max(np.abs(ijk)) >= (1 << 20):
            raise ValueError("voxel index outside the packable range (|index| < 2**20)")
        d_ijk = self.be.from_host(ijk.astype(np.int32))
        d_prob = None if prob is None else self.be.from_host(np.asarray(prob, np.float64))
        d_cnt = None if counts is None else self.be.from_host(np.asarray(counts, np.int32))
        failed = self._scratch
        self._zero_scratch()
        self.lib.voxel("insert", self.desc, self.be.ptr(d_ijk), self.be.ptr(d_prob), float(value if value is not None else 0.0),
                       self.be.ptr(d_cnt), len(ijk), self.be.ptr(failed), self.be.stream())
        f = self.be.to_host(failed)
        if int(f[0]) != 0:
            raise RuntimeError("voxel table full")
        if self._n is not None:
            self._n += int(f[1])

    def _zero_scratch(self) -> None:
        self._fill_zero(self._scratch)

    @staticmethod
    def _fill_zero(z) -> None:
        if hasattr(z, "zero_"):
            z.zero_()
        else:
            z[...] = 0

    def _zeros(self, shape, kind):
        z = self.be.empty(shape, kind)
        self._fill_zero(z)
        return z

    def insert(self, ijk, prob=None, value: Optional[float] = None, counts=None) -> None:
        """Create-or-overwrite voxels by index: per-voxel `prob` or one `value` (what add_obstacle does with 0.9)."""
        ijk = np.asarray(ijk).reshape(-1, 3)
        self.reserve(len(ijk))
        self._insert(ijk, prob, value, counts)

    # ------------------------------------------------------------------ queries
    def _as_dev(self, a, name: str):
        """A host ndarray is uploaded (as float64, the reference's type); a backend array (f32 / f64) is used in place."""
        if isinstance(a, np.ndarray):
            if hasattr(self.be, "torch"):
                a = self.be.from_host(np.ascontiguousarray(a, dtype=np.float64))
            else:                                   # NumPy backend of the emulated library: host arrays are device arrays
                a = np.ascontiguousarray(a if a.dtype in (np.float32, np.float64) else a.astype(np.float64))
        self.be.check(a, name)
        return self.be.suffix(a), a

    def query(self, positions):
        """query_occupancy_batch (mapper.py:173-183): positions (M, 3) host or device -> device float64 (M,)."""
        suf, pos = self._as_dev(positions, "positions")
        M = int(np.prod(pos.shape)) // 3
        occ = self.be.empty((M,), "f64")
        self.lib.voxel(f"query_{suf}", self.desc, self.be.ptr(pos), M, self.be.ptr(occ), self.be.stream())
        return occ

    def trajectories_safe(self, P, n_steps: Optional[int] = None, stride: Optional[int] = None, margin: float = 1.0,
                          threshold: float = 0.6):
        """is_trajectory_safe (mapper.py:195-219) for a batch: P = (B, N, 3) host/device array, or any (B, stride)
        array whose rows start with N x 3 positions (e.g. the solver's X with stride 9N).  -> device int32 (safe, first)."""
        suf, dP = self._as_dev(P, "P")
        B = dP.shape[0]
        if n_steps is None:
            if dP.ndim != 3 or dP.shape[2] != 3:
                raise ValueError("P: expected (B, N, 3), or pass n_steps/stride for packed rows")
            n_steps, stride = dP.shape[1], 3 * dP.shape[1]
        elif stride is None:
            stride = int(np.prod(dP.shape[1:]))
        safe = self.be.empty((B,), "i32")
        first = self.be.empty((B,), "i32")
        self.lib.voxel(f"trajectory_safe_{suf}", self.desc, self.be.ptr(dP), B, int(n_steps), int(stride), float(margin),
                       float(threshold), self.be.ptr(safe), self.be.ptr(first), self.be.stream())
        return safe, first

    def local_spheres(self, centre, size: float = 20.0, threshold: float = 0.6, target: int = 20, radius: float = 1.0,
                      cap: Optional[int] = None, precision: str = "f64"):
        """get_local_occupancy_grid (mapper.py:221-248) + the sphere selection of
        cloud/main_improved_threelayer.py:387-398, fused on the device.  -> (spheres (cap, 4) device, count device int32[2]
        = [spheres written, occupied cells])."""
        n = int(size / self.resolution)
        cap = max(2 * target, 1) if cap is None else cap
        spheres = self.be.empty((cap, 4), precision)
        count = self.be.empty((2,), "i32")
        work = self.be.empty((self.lib.voxel_local_workspace(n),), "i32")
        c3 = (C.c_double * 3)(*[float(v) for v in np.asarray(centre, float).reshape(3)])
        self.lib.voxel(f"local_spheres_{precision}", self.desc, C.byref(c3), float(size), float(threshold), int(target),
                       float(radius), self.be.ptr(spheres), cap, self.be.ptr(count), self.be.ptr(work), self.be.stream())
        return spheres, count

    def trace_rays(self, origins, unit_directions, distances):
        """_trace_ray (mapper.py:251-312) for M rays on the device -> list of (len_i, 3) int64 index arrays in walk order."""
        o = np.ascontiguousarray(np.asarray(origins, np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(unit_directions, np.float64).reshape(-1, 3))
        dist = np.ascontiguousarray(np.asarray(distances, np.float64).reshape(-1))
        M = len(dist)
        if M == 0:
            return []
        max_len = 3 * int(np.ceil(float(np.max(dist)) / self.resolution)) + 8
        ray_keys = self.be.empty((M, max_len), "i64")
        ray_len = self.be.empty((M,), "i32")
        stats = self.be.empty((4,), "i32")
        d_o, d_d, d_dist = (self.be.from_host(x) for x in (o, d, dist))
        self.lib.voxel("trace_rays", self.desc, self.be.ptr(d_o), self.be.ptr(d_d), self.be.ptr(d_dist), M, self.be.ptr(ray_keys),
                       self.be.ptr(ray_len), max_len, self.be.ptr(stats), self.be.stream())
        st = self.be.to_host(stats)
        if int(st[1]) or int(st[2]):
            raise RuntimeError(f"ray walk dropped work: {int(st[1])} rays outside the index range, {int(st[2])} truncated")
        keys = np.array(self.be.to_host(ray_keys)).view(np.uint64)
        lens = np.array(self.be.to_host(ray_len))
        out = []
        for r in range(M):
            k = keys[r, :lens[r]]
            out.append(np.stack([((k >> np.uint64(42)) & np.uint64(0x1FFFFF)).astype(np.int64) - (1 << 20),
                                 ((k >> np.uint64(21)) & np.uint64(0x1FFFFF)).astype(np.int64) - (1 << 20),
                                 (k & np.uint64(0x1FFFFF)).astype(np.int64) - (1 << 20)], axis=1))
        return out

    # ------------------------------------------------------------------ update_map
    def update_rays(self, origins, unit_directions, distances, hits, like_hit: float = 0.7, like_miss: float = 0.6):
        """update_map (mapper.py:102-153) for M observations in order (chunks of at most 1024 rays per launch group).
        Returns (voxel updates, total voxels)."""
        o = np.ascontiguousarray(np.asarray(origins, np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(unit_directions, np.float64).reshape(-1, 3))
        dist = np.ascontiguousarray(np.asarray(distances, np.float64).reshape(-1))
        hit = np.ascontiguousarray(np.asarray(hits, np.int32).reshape(-1))
        M = len(dist)
        if not (len(o) == len(d) == len(hit) == M):
            raise ValueError("origins, directions, distances and hits must have one row per observation")
        if M == 0:
            return 0, len(self)
        max_len = 3 * int(np.ceil(float(np.max(dist)) / self.resolution)) + 8
        # Validate BEFORE the first launch, so a scan is applied whole or not at all (the reference's dict never fails): every voxel a
        # ray can walk lies within one cell of the box spanned by its end points, which must stay inside the packable index range.
        ends = np.concatenate([o, o + d * dist[:, None]]) / self.resolution
        if not np.all(np.isfinite(ends)) or np.max(np.abs(ends)) >= (1 << 20) - 2:
            raise ValueError("a ray leaves the packable voxel index range (|index| < 2^20); nothing was applied")
        per_ray = 3 * np.ceil(dist / self.resolution) + 8                        # every walked voxel could be new
        chunk = min(M, 1024)
        while chunk > 64 and self.lib.voxel_update_row_words(chunk, max_len) * 8 > (256 << 20):
            chunk //= 2                                                            # keep the bit-set workspace under 256 MiB
        words = self.lib.voxel_update_row_words(chunk, max_len)
        ray_keys = self.be.empty((chunk, max_len), "i64")
        ray_len = self.be.empty((chunk,), "i32")
        stats = self.be.empty((4,), "i32")
        updates = 0
        for lo in range(0, M, chunk):
            hi = min(M, lo + chunk)
            n = hi - lo
            self.reserve(int(np.sum(per_ray[lo:hi])))                            # per chunk: a long scan does not size the table for all its rays at once
            if self._row_bits is None or int(np.prod(self._row_bits.shape)) < words:   # (a re-hash drops the workspace)
                self._row_bits = self._zeros((words,), "i64")
            # one upload per chunk: [origin 3n | direction 3n | distance n] float64, then hit n int32
            packed = np.empty(7 * n * 8 + n * 4, dtype=np.uint8)
            f64 = packed[:7 * n * 8].view(np.float64)
            f64[:3 * n] = o[lo:hi].ravel(); f64[3 * n:6 * n] = d[lo:hi].ravel(); f64[6 * n:] = dist[lo:hi]
            packed[7 * n * 8:].view(np.int32)[:] = hit[lo:hi]
            d_in = self.be.from_host(packed)
            base = self.be.ptr(d_in)
            self.lib.voxel("update_rays", self.desc, base, base + 3 * n * 8, base + 6 * n * 8, base + 7 * n * 8,
                           n, float(like_hit), float(like_miss), self.be.ptr(ray_keys), self.be.ptr(ray_len), max_len,
                           self.be.ptr(self._slot_rows), self.be.ptr(self._row_bits), self.be.ptr(stats), self.be.stream())
            st = self.be.to_host(stats)
            if int(st[1]) or int(st[2]):
                self._n = None                                                    # recount from the device on the next len()
                raise RuntimeError(f"voxel update dropped work: {int(st[1])} voxels not stored, {int(st[2])} rays truncated "
                                   f"(chunk {lo}:{hi}; earlier chunks are applied)")
            updates += int(st[0])
            if self._n is not None:
                self._n += int(st[3])
        return updates, len(self)
