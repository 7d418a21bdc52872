// voxel_map.hip -- device-resident voxel map: the obstacle source of the SE(3) MPC path.
//
// Counterpart of ExplicitGeometricMapper (reference src/dart_planner/perception/explicit_geometric_mapper.py,
// "mapper.py" below).  The reference keeps a Python dict {(ix, iy, iz): VoxelData} and answers every query with a dict
// lookup (one million of them for the default 20 m / 0.2 m local grid of get_local_occupancy_grid, each planning
// cycle of cloud/main_improved_threelayer.py:381-398).  Here the dict is an open-addressing hash table in HBM
// (64-bit packed key, linear probing, multiplicative hash); it is tens of kilobytes to a few megabytes, so it lives
// in L2 / Infinity Cache and every kernel below is bound by hash probes and integer/f64 index arithmetic, not by HBM
// streams.  Voxel indices and probabilities follow the reference's float64 arithmetic operation by operation
// (floor(p / resolution), numpy.linspace's i * step + start with the last point pinned, no FMA contraction),
// so results are bit-identical to the reference's (tests/golden/mapper_map.npz).
#include <hip/hip_runtime.h>

#include <cmath>

#include "se3mpc_common.hpp"
#include <se3mpc_wave_ops.hpp>

static_assert(sizeof(se3mpc_voxel_map) == 48, "se3mpc_voxel_map is part of the C ABI (dart_planner_amd/capi.py mirrors it)");

namespace se3mpc {

struct VoxDev {
  unsigned long long* keys;
  double* prob;
  int32_t* count;
  uint32_t mask;
  int shift;          // 64 - log2(capacity)
  double res, prior;
};

constexpr long long kVoxBias = 1ll << 20;
constexpr int kVoxOut = INT32_MIN;             // axis index outside the packable range
constexpr unsigned long long kVoxEmpty = SE3MPC_VOXEL_EMPTY;

static int make_vox_dev(const se3mpc_voxel_map* m, VoxDev& d) {
  if (m == nullptr || m->keys == nullptr || m->prob == nullptr || m->count == nullptr) return SE3MPC_ERR_NULL;
  const int cap = m->capacity;
  if (cap < 64 || (cap & (cap - 1)) != 0) return SE3MPC_ERR_SHAPE;
  if (!(m->resolution > 0.0) || !std::isfinite(m->resolution) || !std::isfinite(m->prior)) return SE3MPC_ERR_PARAM;
  int lg = 0;
  while ((1 << lg) < cap) ++lg;
  d.keys = reinterpret_cast<unsigned long long*>(m->keys);
  d.prob = m->prob;
  d.count = m->count;
  d.mask = (uint32_t)cap - 1u;
  d.shift = 64 - lg;
  d.res = m->resolution;
  d.prior = m->prior;
  return SE3MPC_OK;
}

// world_to_voxel, one axis (mapper.py:93-96): floor(p / resolution) in float64
__device__ __forceinline__ int vox_axis(double p, double res) {
  const double v = floor(p / res);
  return (fabs(v) < (double)kVoxBias) ? (int)v : kVoxOut;      // NaN fails the comparison too
}

__device__ __forceinline__ bool vox_pack(int ix, int iy, int iz, unsigned long long& key) {
  if (ix == kVoxOut || iy == kVoxOut || iz == kVoxOut) return false;
  const long long lim = kVoxBias;
  if (ix < -lim || ix >= lim || iy < -lim || iy >= lim || iz < -lim || iz >= lim) return false;
  key = ((unsigned long long)(ix + kVoxBias) << 42) | ((unsigned long long)(iy + kVoxBias) << 21) |
        (unsigned long long)(iz + kVoxBias);
  return true;
}

__device__ __forceinline__ void vox_unpack(unsigned long long key, int& ix, int& iy, int& iz) {
  ix = (int)((long long)((key >> 42) & 0x1FFFFFull) - kVoxBias);
  iy = (int)((long long)((key >> 21) & 0x1FFFFFull) - kVoxBias);
  iz = (int)((long long)(key & 0x1FFFFFull) - kVoxBias);
}

__device__ __forceinline__ uint32_t vox_hash(const VoxDev& d, unsigned long long key) {
  return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> d.shift) & d.mask;
}

// slot of `key`, or -1 (the table always keeps at least one empty slot per probe run at sane load factors; the
// loop is bounded by the capacity regardless)
__device__ __forceinline__ int vox_find(const VoxDev& d, unsigned long long key) {
  uint32_t h = vox_hash(d, key);
  for (uint32_t n = 0; n <= d.mask; ++n) {
    const unsigned long long k = d.keys[h];
    if (k == key) return (int)h;
    if (k == kVoxEmpty) return -1;
    h = (h + 1u) & d.mask;
  }
  return -1;
}

// slot of `key`, claiming a free one if it is new (`created` is then set); -1 = table full
__device__ __forceinline__ int vox_find_or_claim(const VoxDev& d, unsigned long long key, bool& created) {
  uint32_t h = vox_hash(d, key);
  created = false;
  for (uint32_t n = 0; n <= d.mask; ++n) {
    unsigned long long k = d.keys[h];
    if (k == kVoxEmpty) {
      k = atomicCAS(&d.keys[h], kVoxEmpty, key);
      if (k == kVoxEmpty) { created = true; return (int)h; }
    }
    if (k == key) return (int)h;
    h = (h + 1u) & d.mask;
  }
  return -1;
}

// query_occupancy (mapper.py:155-171) by voxel index
__device__ __forceinline__ double vox_occupancy_idx(const VoxDev& d, int ix, int iy, int iz) {
  unsigned long long key;
  if (!vox_pack(ix, iy, iz, key)) return d.prior;
  const int s = vox_find(d, key);
  return s < 0 ? d.prior : d.prob[s];
}

__device__ __forceinline__ double vox_occupancy(const VoxDev& d, double x, double y, double z) {
  return vox_occupancy_idx(d, vox_axis(x, d.res), vox_axis(y, d.res), vox_axis(z, d.res));
}

// ------------------------------------------------------------------------------------------ table maintenance
__global__ void __launch_bounds__(256) voxel_clear_kernel(VoxDev d) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= d.mask; i += gridDim.x * blockDim.x) {
    d.keys[i] = kVoxEmpty;
    d.prob[i] = d.prior;
    d.count[i] = 0;
  }
}

__global__ void __launch_bounds__(256)
voxel_insert_kernel(VoxDev d, const int32_t* __restrict__ ijk, const double* __restrict__ prob, double value,
                    const int32_t* __restrict__ count_in, int M, int32_t* __restrict__ failed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  unsigned long long key;
  int s = -1;
  bool created = false;
  if (vox_pack(ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2], key)) s = vox_find_or_claim(d, key, created);
  if (s >= 0) {
    d.prob[s] = prob != nullptr ? prob[i] : value;
    if (count_in != nullptr) d.count[s] = count_in[i];
    if (created && failed != nullptr) atomicAdd(&failed[1], 1);
  } else if (failed != nullptr) {
    atomicAdd(&failed[0], 1);
  }
}

__global__ void __launch_bounds__(256)
voxel_export_kernel(VoxDev d, int32_t* __restrict__ ijk_out, double* __restrict__ prob_out, int32_t* __restrict__ count_out,
                    int32_t* __restrict__ n_out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= d.mask; i += gridDim.x * blockDim.x) {
    const unsigned long long k = d.keys[i];
    if (k == kVoxEmpty) continue;
    const int o = atomicAdd(n_out, 1);
    int ix, iy, iz;
    vox_unpack(k, ix, iy, iz);
    if (ijk_out != nullptr) { ijk_out[3 * o] = ix; ijk_out[3 * o + 1] = iy; ijk_out[3 * o + 2] = iz; }
    if (prob_out != nullptr) prob_out[o] = d.prob[i];
    if (count_out != nullptr) count_out[o] = d.count[i];
  }
}

// ------------------------------------------------------------------------------------------ queries
// query_occupancy_batch (mapper.py:173-183): one position per lane
template <typename R>
__global__ void __launch_bounds__(256)
voxel_query_kernel(VoxDev d, const R* __restrict__ pos, int M, double* __restrict__ occ) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  occ[i] = vox_occupancy(d, (double)pos[(size_t)3 * i], (double)pos[(size_t)3 * i + 1], (double)pos[(size_t)3 * i + 2]);
}

// is_trajectory_safe (mapper.py:195-219) with the seven check positions of _get_safety_margin_positions
// (:339-353): one wavefront per trajectory, lane = step; the first colliding step comes out of one ballot.
template <typename R>
__global__ void __launch_bounds__(256)
voxel_trajectory_safe_kernel(VoxDev d, const R* __restrict__ P, int B, int N, long long stride, double margin, double thr,
                             int32_t* __restrict__ safe, int32_t* __restrict__ first) {
  const int lane = lane_id();
  const int b = blockIdx.x * (blockDim.x / kWave) + (int)(threadIdx.x / kWave);
  if (b >= B) return;                                          // whole wavefront
  const R* Pb = P + (size_t)b * (size_t)stride;
  int found = -1;
  for (int k0 = 0; k0 < N; k0 += kWave) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < N) {
      const double c[3] = {(double)Pb[3 * k], (double)Pb[3 * k + 1], (double)Pb[3 * k + 2]};
      hit = vox_occupancy(d, c[0], c[1], c[2]) > thr;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int sg = -1; sg <= 1; sg += 2) {
          double q[3] = {c[0], c[1], c[2]};                     // centre + offset, offset = 0 off the axis (mapper.py:347-351)
          q[a] = c[a] + (double)sg * margin;
          hit = hit || (vox_occupancy(d, q[0], q[1], q[2]) > thr);
        }
      }
    }
    const uint64_t m = wave_ballot(hit);
    if (m != 0ull) { found = k0 + first_lane(m); break; }
  }
  if (lane == 0) {
    if (safe != nullptr) safe[b] = found < 0 ? 1 : 0;
    if (first != nullptr) first[b] = found;
  }
}

// ------------------------------------------------------------------------------------------ local grid -> spheres
struct GridDev {
  double start[3], stop[3], step[3];
  int n;                // cells per axis
  long long M;          // n^3
  double thr, radius;
  int target, cap;
};

// numpy.linspace(start, stop, n)[i]: i * step + start (two roundings), the last point pinned to stop
__device__ __forceinline__ double grid_axis_value(const GridDev& g, int a, int i) {
#pragma clang fp contract(off)
  if (g.n > 1 && i == g.n - 1) return g.stop[a];
  const double t = (double)i * g.step[a];
  return t + g.start[a];
}

constexpr int kGridRun = 256;    // consecutive cells owned by one wavefront (4 rows of 64: enough wavefronts to hide the probe latency)

// Axis tables in LDS: voxel index of every linspace point, per axis -- 3n divisions per workgroup instead of 3 per cell.
__device__ __forceinline__ void grid_stage_axes(const VoxDev& d, const GridDev& g, int* ax) {
  for (int i = threadIdx.x; i < 3 * g.n; i += blockDim.x) {
    const int a = i / g.n, j = i - a * g.n;
    ax[i] = vox_axis(grid_axis_value(g, a, j), d.res);
  }
  __syncthreads();
}

// cell m of the reference's flattened grid = (iz, ix, iy), iy fastest (np.array(np.meshgrid(x, y, z)).T.reshape(-1, 3))
__device__ __forceinline__ bool grid_cell_occupied(const VoxDev& d, const GridDev& g, const int* ax, long long m, int& ix,
                                                   int& iy, int& iz) {
  const long long n = g.n;
  iy = (int)(m % n);
  ix = (int)((m / n) % n);
  iz = (int)(m / (n * n));
  return vox_occupancy_idx(d, ax[ix], ax[g.n + iy], ax[2 * g.n + iz]) > g.thr;
}

__global__ void __launch_bounds__(256)
voxel_grid_count_kernel(VoxDev d, GridDev g, int32_t* __restrict__ run_counts, unsigned long long* __restrict__ masks) {
  HIP_DYNAMIC_SHARED(int, ax)
  grid_stage_axes(d, g, ax);
  const int lane = lane_id();
  const long long run = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const long long base = run * kGridRun;
  if (base >= g.M) return;
  int cnt = 0;
  for (int r = 0; r < kGridRun / kWave; ++r) {
    const long long m = base + r * kWave + lane;
    int ix, iy, iz;
    const bool o = m < g.M && grid_cell_occupied(d, g, ax, m, ix, iy, iz);
    const uint64_t mask = wave_ballot(o);
    if (lane == 0) masks[run * (kGridRun / kWave) + r] = mask;          // the second pass reads the answer instead of probing again
    cnt += __builtin_popcountll(mask);
  }
  if (lane == 0) run_counts[run] = cnt;
}

// Exclusive prefix sum of the per-run counts (one workgroup; offsets[nruns] = total): every thread sums a contiguous
// chunk, the 1024 chunk sums are scanned with wave shuffles + the 16 wave totals in LDS, then the chunk is rewritten.
__global__ void __launch_bounds__(1024)
voxel_grid_scan_kernel(const int32_t* __restrict__ counts, int nruns, int32_t* __restrict__ offsets) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int chunk = (nruns + 1023) / 1024;
  const int lo = tid * chunk, hi = (lo + chunk < nruns) ? lo + chunk : nruns;
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += counts[i];
  int incl = sum;                                            // inclusive scan across the wavefront
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl(incl, lane - off >= 0 ? lane - off : lane, kWave);
    if (lane >= off) incl += up;
  }
  if (lane == kWave - 1) wave_tot[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < 16; ++w) { if (w < wave) base += wave_tot[w]; total += wave_tot[w]; }
  int run = base + incl - sum;
  for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
  if (tid == 0) offsets[nruns] = total;
}

template <typename R>
__global__ void __launch_bounds__(256)
voxel_grid_select_kernel(GridDev g, const int32_t* __restrict__ offsets, const unsigned long long* __restrict__ masks,
                         int nruns, R* __restrict__ spheres, int32_t* __restrict__ count) {
  const int lane = lane_id();
  const long long run = (long long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const long long base = run * kGridRun;
  if (base >= g.M) return;
  // rank of this run's first occupied cell and the grand total, from the scanned per-run counts
  const int before = offsets[run], total = offsets[nruns];
  const int step = (g.target > 0 && total / g.target > 1) ? total / g.target : 1;   // max(1, n_occupied // target)
  int rank = before;
  const long long n = g.n;
  for (int r = 0; r < kGridRun / kWave; ++r) {
    const uint64_t mask = masks[run * (kGridRun / kWave) + r];
    if ((mask >> lane) & 1ull) {
      const int rk = rank + __builtin_popcountll(mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      if (rk % step == 0 && rk / step < g.cap) {
        const long long m = base + r * kWave + lane;
        const int iy = (int)(m % n), ix = (int)((m / n) % n), iz = (int)(m / (n * n));   // (iz, ix, iy), iy fastest
        R* s4 = spheres + (size_t)4 * (rk / step);
        s4[0] = (R)grid_axis_value(g, 0, ix); s4[1] = (R)grid_axis_value(g, 1, iy); s4[2] = (R)grid_axis_value(g, 2, iz);
        s4[3] = (R)g.radius;
      }
    }
    rank += __builtin_popcountll(mask);
  }
  if (run == 0 && lane == 0) {
    const int k = total == 0 ? 0 : (total + step - 1) / step;
    count[0] = k < g.cap ? k : g.cap;
    count[1] = total;
  }
}

// ------------------------------------------------------------------------------------------ update_map
// _trace_ray (mapper.py:251-312): the reference's DDA, one ray per lane, float64 operation by operation (no FMA
// contraction: `end = start + direction * distance` and the boundary arithmetic round twice, as NumPy does).
// Emits packed voxel keys (EMPTY for a voxel outside the packable range) into ray_keys[ray][0..len).
__global__ void __launch_bounds__(64)
voxel_trace_kernel(VoxDev d, const double* __restrict__ origin, const double* __restrict__ direction,
                   const double* __restrict__ distance, int M, unsigned long long* __restrict__ ray_keys,
                   int32_t* __restrict__ ray_len, int max_len, int32_t* __restrict__ stats) {
#pragma clang fp contract(off)
  const int ray = blockIdx.x * blockDim.x + threadIdx.x;
  if (ray >= M) return;
  const double res = d.res, dist = distance[ray];
  // per-axis state in named scalars: a dynamically indexed private array would live in scratch memory
  const double s0 = origin[3 * ray], s1 = origin[3 * ray + 1], s2 = origin[3 * ray + 2];
  const double d0 = direction[3 * ray], d1 = direction[3 * ray + 1], d2 = direction[3 * ray + 2];
  const double f0 = floor(s0 / res), f1 = floor(s1 / res), f2 = floor(s2 / res);
  const double g0 = floor((s0 + d0 * dist) / res), g1 = floor((s1 + d1 * dist) / res), g2 = floor((s2 + d2 * dist) / res);   // :268-270
  // both end voxels well inside the packable range => every voxel of the walk is: it stays in their bounding box,
  // except that rounding may carry it a step or two past the end voxel before `total > distance` stops it (hence the
  // margin).  A ray that leaves the range cannot be stored and is dropped as a whole (counted in stats[1]).
  const double lim = (double)(kVoxBias - 16);
  const bool ok = fabs(f0) <= lim && fabs(f1) <= lim && fabs(f2) <= lim && fabs(g0) <= lim && fabs(g1) <= lim && fabs(g2) <= lim;
  unsigned long long* out = ray_keys + (size_t)ray * (size_t)max_len;
  int len = 0;
  bool truncated = false;
  if (ok) {
    const int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
    const int e0 = (int)g0, e1 = (int)g1, e2 = (int)g2;
    const int st0 = e0 > c0 ? 1 : (e0 < c0 ? -1 : 0), st1 = e1 > c1 ? 1 : (e1 < c1 ? -1 : 0), st2 = e2 > c2 ? 1 : (e2 < c2 ? -1 : 0);
    // t_delta (:285-291), first boundary (:294-296), t_max (:298-304)
    const double td0 = st0 ? res / fabs(d0) : INFINITY, td1 = st1 ? res / fabs(d1) : INFINITY, td2 = st2 ? res / fabs(d2) : INFINITY;
    double tm0 = st0 ? fabs(((double)(c0 + (st0 > 0 ? 1 : 0)) * res - s0) / d0) : INFINITY;
    double tm1 = st1 ? fabs(((double)(c1 + (st1 > 0 ? 1 : 0)) * res - s1) / d1) : INFINITY;
    double tm2 = st2 ? fabs(((double)(c2 + (st2 > 0 ? 1 : 0)) * res - s2) / d2) : INFINITY;
    // The walk is kept as (a) the packed key, advanced by one signed per-axis increment per step, and (b) the signed
    // step counts still to go per axis, r_a = (end_a - cur_a) * step_a: `cur != end_voxel` (:307) is r0|r1|r2 != 0,
    // also after an overshoot (r_a < 0), exactly as the reference's tuple comparison.
    unsigned long long key = 0;
    vox_pack(c0, c1, c2, key);
    const unsigned long long k0 = (unsigned long long)((long long)st0 << 42), k1 = (unsigned long long)((long long)st1 << 21),
                             k2 = (unsigned long long)(long long)st2;
    int r0 = (e0 - c0) * st0, r1 = (e1 - c1) * st1, r2 = (e2 - c2) * st2;
    out[len++] = key;
    double total = 0.0;
    while ((r0 | r1 | r2) != 0 && total <= dist) {
      if (len >= max_len) { truncated = true; break; }
      if (tm0 <= tm1 && tm0 <= tm2) { total = tm0; tm0 += td0; key += k0; --r0; }           // np.argmin: the first minimum
      else if (tm1 <= tm2) { total = tm1; tm1 += td1; key += k1; --r1; }
      else { total = tm2; tm2 += td2; key += k2; --r2; }
      out[len++] = key;
    }
  } else if (stats != nullptr) {
    atomicAdd(&stats[1], 1);
  }
  ray_len[ray] = len;
  if (stats != nullptr) {
    if (len) atomicAdd(&stats[0], len);                      // walked voxels (those that cannot be stored are in stats[1])
    if (truncated) atomicAdd(&stats[2], 1);
  }
}

// _bayesian_update (mapper.py:314-337)
__device__ __forceinline__ double vox_bayes(double p, double like) {
#pragma clang fp contract(off)
  const double num = like * p;
  const double den = like * p + (1.0 - like) * (1.0 - p);
  if (den > 0.0) p = num / den;
  return fmin(fmax(p, 0.01), 0.99);
}

// Slot resolution for every walked voxel, fully parallel (creating a voxel does not depend on the order).  Record
// j = (ray, i) finds or claims its table slot (written back in place of the key) and files its ray into the slot's
// "row": two bit sets over the rays of this call -- which rays touch the voxel, and which of those end in it with a
// hit.  The row of a slot is the one of the first record that reaches it (slot_rows[slot] = j + 1 by atomicCAS: no
// counter, no waiting).  Rows and slot_rows are all-zero between calls.
__global__ void __launch_bounds__(256)
voxel_resolve_kernel(VoxDev d, unsigned long long* __restrict__ ray_keys, const int32_t* __restrict__ ray_len,
                     const int32_t* __restrict__ hit, int M, int max_len, int32_t* __restrict__ slot_rows,
                     unsigned long long* __restrict__ row_bits, int W, int32_t* __restrict__ stats) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = j < (long long)M * max_len;
  const int ray = in_range ? (int)(j / max_len) : 0, i = in_range ? (int)(j - (long long)ray * max_len) : 0;
  const int len = in_range ? ray_len[ray] : 0;
  const bool live = in_range && i < len;
  bool created = false, lost = false;
  if (live) {
    const unsigned long long key = ray_keys[j];
    const int s = key == kVoxEmpty ? -1 : vox_find_or_claim(d, key, created);
    ray_keys[j] = s < 0 ? kVoxEmpty : (unsigned long long)s;
    lost = s < 0;
    if (s >= 0) {
      int row = atomicCAS(&slot_rows[s], 0, (int)j + 1);
      if (row == 0) row = (int)j + 1;
      unsigned long long* bits = row_bits + (size_t)(row - 1) * (size_t)(2 * W);
      const unsigned long long bit = 1ull << (ray & 63);
      atomicOr(&bits[ray >> 6], bit);
      if (i == len - 1 && hit[ray] != 0) atomicOr(&bits[W + (ray >> 6)], bit);        // mapper.py:121-123
    }
  }
  // one atomic per wavefront for the counters (same-address atomics serialise)
  const int n_created = __builtin_popcountll(wave_ballot(created)), n_lost = __builtin_popcountll(wave_ballot(lost));
  if (stats != nullptr && lane_id() == 0) {
    if (n_created) atomicAdd(&stats[3], n_created);
    if (n_lost) atomicAdd(&stats[1], n_lost);
  }
}

// The updates of update_map (mapper.py:114-141): the clamped Bayesian update does not commute, so every voxel must
// see its observations in observation order -- but voxels are independent of each other.  The record that owns a
// voxel's row walks the row's bits in ray order (a ray visits a voxel at most once) and applies hit / pass-through
// updates one after another; all touched voxels proceed in parallel, the longest chain is the sensor's own voxel
// (every ray).  The row and the slot's row id are cleared on the way out.
__global__ void __launch_bounds__(256)
voxel_apply_kernel(VoxDev d, const unsigned long long* __restrict__ ray_slots, const int32_t* __restrict__ ray_len, int M,
                   int max_len, int32_t* __restrict__ slot_rows, unsigned long long* __restrict__ row_bits, int W,
                   double like_hit, double like_miss) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= (long long)M * max_len) return;
  const int ray = (int)(j / max_len), i = (int)(j - (long long)ray * max_len);
  if (i >= ray_len[ray]) return;
  const unsigned long long sl = ray_slots[j];
  if (sl == kVoxEmpty || slot_rows[sl] != (int)j + 1) return;            // not stored, or another record owns the row
  unsigned long long* bits = row_bits + (size_t)j * (size_t)(2 * W);
  double p = d.prob[sl];
  int n = 0;
  // The update is a pure function of (p, kind): once a kind of update leaves p unchanged it will do so again at this
  // p (the clamp makes 0.99 / 0.01 such fixed points -- the sensor's own voxel saturates after a dozen pass-throughs and
  // is then touched by every remaining ray).  A kind known to be idle is skipped; when both are, the rest is a popcount.
  bool idle_hit = false, idle_miss = false;
  for (int w = 0; w < W; ++w) {
    unsigned long long touched = bits[w];
    const unsigned long long hits = bits[W + w];
    bits[w] = 0ull;
    bits[W + w] = 0ull;
    while (touched != 0ull) {
      const unsigned long long lowest = touched & (0ull - touched);
      const bool is_hit = (hits & lowest) != 0ull;
      if (is_hit ? idle_hit : idle_miss) {
        // an idle kind: take the whole run of it, up to the next update of the other kind, in one step
        const unsigned long long other = is_hit ? (touched & ~hits) : (touched & hits);
        const unsigned long long run = other != 0ull ? (touched & ((other & (0ull - other)) - 1ull)) : touched;
        n += __builtin_popcountll(run);
        touched ^= run;
        continue;
      }
      const double pn = vox_bayes(p, is_hit ? like_hit : like_miss);
      if (pn == p) { if (is_hit) idle_hit = true; else idle_miss = true; }
      else { idle_hit = false; idle_miss = false; p = pn; }
      ++n;
      touched ^= lowest;
    }
  }
  d.prob[sl] = p;
  d.count[sl] += n;
  slot_rows[sl] = 0;
}

// ------------------------------------------------------------------------------------------ host side
template <typename R>
int voxel_query_impl(const se3mpc_voxel_map* m, const R* pos, int M, double* occ, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (M < 0) return SE3MPC_ERR_SHAPE;
  if (M == 0) return SE3MPC_OK;
  if (!pos || !occ) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(voxel_query_kernel<R>, dim3(grid_for(M, 256)), dim3(256), 0, (hipStream_t)stream, d, pos, M, occ);
  return launch_status("se3mpc_voxel_query");
}

template <typename R>
int voxel_trajectory_safe_impl(const se3mpc_voxel_map* m, const R* P, int B, int N, long long stride, double margin,
                               double threshold, int32_t* safe, int32_t* first, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (B < 0 || N < 0 || stride < (long long)3 * N) return SE3MPC_ERR_SHAPE;
  if (!std::isfinite(margin) || !std::isfinite(threshold)) return SE3MPC_ERR_PARAM;
  if (B == 0) return SE3MPC_OK;
  if ((N > 0 && !P) || (!safe && !first)) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(voxel_trajectory_safe_kernel<R>, dim3(grid_for(B, 4)), dim3(256), 0, (hipStream_t)stream, d, P, B, N,
                     stride, margin, threshold, safe, first);
  return launch_status("se3mpc_voxel_trajectory_safe");
}

static int grid_runs(long long M) { return (int)((M + kGridRun - 1) / kGridRun); }

template <typename R>
int voxel_local_spheres_impl(const se3mpc_voxel_map* m, const double* centre, double size, double threshold, int target,
                             double radius, R* spheres, int cap, int32_t* count, int32_t* workspace, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (!centre || !count) return SE3MPC_ERR_NULL;
  if (cap < 0 || target < 1) return SE3MPC_ERR_SHAPE;
  if (!std::isfinite(size) || !(size >= 0.0) || !std::isfinite(threshold) || !std::isfinite(radius)) return SE3MPC_ERR_PARAM;
  hipStream_t s = (hipStream_t)stream;
  GridDev g;
  const double half = size / 2;                              // mapper.py:231-233
  g.n = (int)(size / m->resolution);                         // num_cells = int(size / resolution)   (:236)
  if (g.n > SE3MPC_VOXEL_MAX_CELLS) return SE3MPC_ERR_SHAPE;
  if (g.n < 1) {                                             // linspace(..., 0) is empty: no cells, no spheres
    if (hipMemsetAsync(count, 0, 2 * sizeof(int32_t), s) != hipSuccess) return launch_status("se3mpc_voxel_local_spheres(memset)");
    return SE3MPC_OK;
  }
  if ((cap > 0 && !spheres) || !workspace) return SE3MPC_ERR_NULL;
  for (int a = 0; a < 3; ++a) {
    if (!std::isfinite(centre[a])) return SE3MPC_ERR_PARAM;
    g.start[a] = centre[a] - half;
    g.stop[a] = centre[a] + half;
    const double delta = g.stop[a] - g.start[a];
    g.step[a] = g.n > 1 ? delta / (double)(g.n - 1) : 0.0;   // numpy.linspace: step = delta / div
  }
  g.M = (long long)g.n * g.n * g.n;
  g.thr = threshold; g.radius = radius; g.target = target; g.cap = cap;
  const int nruns = grid_runs(g.M);
  const int nblk = grid_for(nruns, 4);
  const size_t lds = (size_t)3 * g.n * sizeof(int);
  // workspace: [nruns run counts | nruns + 1 offsets | pad to 8 B | one 64-bit occupancy ballot per 64 cells]
  int32_t* offsets = workspace + nruns;
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(workspace + ((2 * nruns + 2) & ~1));
  hipLaunchKernelGGL(voxel_grid_count_kernel, dim3(nblk), dim3(256), lds, s, d, g, workspace, masks);
  rc = launch_status("se3mpc_voxel_local_spheres(count)");
  if (rc) return rc;
  hipLaunchKernelGGL(voxel_grid_scan_kernel, dim3(1), dim3(1024), 0, s, workspace, nruns, offsets);
  rc = launch_status("se3mpc_voxel_local_spheres(scan)");
  if (rc) return rc;
  hipLaunchKernelGGL(voxel_grid_select_kernel<R>, dim3(nblk), dim3(256), 0, s, g, offsets, masks, nruns, spheres, count);
  return launch_status("se3mpc_voxel_local_spheres(select)");
}

}  // namespace se3mpc

using namespace se3mpc;

extern "C" int se3mpc_voxel_clear(const se3mpc_voxel_map* m, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  int grid = grid_for(m->capacity, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(voxel_clear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d);
  return launch_status("se3mpc_voxel_clear");
}

extern "C" int se3mpc_voxel_insert(const se3mpc_voxel_map* m, const int32_t* ijk, const double* prob, double value,
                                   const int32_t* count_in, int M, int32_t* failed, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (M < 0) return SE3MPC_ERR_SHAPE;
  if (M == 0) return SE3MPC_OK;
  if (!ijk) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(voxel_insert_kernel, dim3(grid_for(M, 256)), dim3(256), 0, (hipStream_t)stream, d, ijk, prob, value, count_in, M,
                     failed);
  return launch_status("se3mpc_voxel_insert");
}

extern "C" int se3mpc_voxel_export(const se3mpc_voxel_map* m, int32_t* ijk_out, double* prob_out, int32_t* count_out,
                                   int32_t* n_out, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (!n_out) return SE3MPC_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(n_out, 0, sizeof(int32_t), s) != hipSuccess) return launch_status("se3mpc_voxel_export(memset)");
  int grid = grid_for(m->capacity, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(voxel_export_kernel, dim3(grid), dim3(256), 0, s, d, ijk_out, prob_out, count_out, n_out);
  return launch_status("se3mpc_voxel_export");
}

extern "C" int se3mpc_voxel_update_rays(const se3mpc_voxel_map* m, const double* origin, const double* direction,
                                        const double* distance, const int32_t* hit, int M, double like_hit, double like_miss,
                                        uint64_t* ray_keys, int32_t* ray_len, int max_len, int32_t* slot_rows,
                                        uint64_t* row_bits, int32_t* stats, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (M < 0 || M > SE3MPC_VOXEL_MAX_RAYS || max_len < 1 || (long long)M * max_len >= (1ll << 31)) return SE3MPC_ERR_SHAPE;
  if (!(like_hit > 0.0 && like_hit < 1.0 && like_miss > 0.0 && like_miss < 1.0)) return SE3MPC_ERR_PARAM;
  hipStream_t s = (hipStream_t)stream;
  if (stats != nullptr && hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), s) != hipSuccess)
    return launch_status("se3mpc_voxel_update_rays(memset)");
  if (M == 0) return SE3MPC_OK;
  if (!origin || !direction || !distance || !hit || !ray_keys || !ray_len || !slot_rows || !row_bits) return SE3MPC_ERR_NULL;
  const int W = (M + 63) / 64;
  hipLaunchKernelGGL(voxel_trace_kernel, dim3(grid_for(M, 64)), dim3(64), 0, s, d, origin, direction, distance, M,
                     reinterpret_cast<unsigned long long*>(ray_keys), ray_len, max_len, stats);
  rc = launch_status("se3mpc_voxel_update_rays(trace)");
  if (rc) return rc;
  const long long recs = (long long)M * max_len;
  const unsigned nblk = (unsigned)((recs + 255) / 256);
  hipLaunchKernelGGL(voxel_resolve_kernel, dim3(nblk), dim3(256), 0, s, d, reinterpret_cast<unsigned long long*>(ray_keys),
                     ray_len, hit, M, max_len, slot_rows, reinterpret_cast<unsigned long long*>(row_bits), W, stats);
  rc = launch_status("se3mpc_voxel_update_rays(resolve)");
  if (rc) return rc;
  hipLaunchKernelGGL(voxel_apply_kernel, dim3(nblk), dim3(256), 0, s, d, reinterpret_cast<const unsigned long long*>(ray_keys),
                     ray_len, M, max_len, slot_rows, reinterpret_cast<unsigned long long*>(row_bits), W, like_hit, like_miss);
  return launch_status("se3mpc_voxel_update_rays(apply)");
}

extern "C" int se3mpc_voxel_trace_rays(const se3mpc_voxel_map* m, const double* origin, const double* direction,
                                       const double* distance, int M, uint64_t* ray_keys, int32_t* ray_len, int max_len,
                                       int32_t* stats, void* stream) {
  VoxDev d;
  int rc = make_vox_dev(m, d);
  if (rc) return rc;
  if (M < 0 || max_len < 1 || (long long)M * max_len >= (1ll << 31)) return SE3MPC_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (stats != nullptr && hipMemsetAsync(stats, 0, 4 * sizeof(int32_t), s) != hipSuccess)
    return launch_status("se3mpc_voxel_trace_rays(memset)");
  if (M == 0) return SE3MPC_OK;
  if (!origin || !direction || !distance || !ray_keys || !ray_len) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(voxel_trace_kernel, dim3(grid_for(M, 64)), dim3(64), 0, s, d, origin, direction, distance, M,
                     reinterpret_cast<unsigned long long*>(ray_keys), ray_len, max_len, stats);
  return launch_status("se3mpc_voxel_trace_rays");
}

extern "C" long long se3mpc_voxel_update_row_words(int M, int max_len) {
  if (M < 1 || max_len < 1) return 0;
  return (long long)M * max_len * 2 * ((M + 63) / 64);
}

extern "C" int se3mpc_voxel_local_workspace(int cells_per_axis) {
  if (cells_per_axis < 1) return 1;
  const long long n = cells_per_axis;
  const int nruns = grid_runs(n * n * n);
  return ((2 * nruns + 2) & ~1) + 2 * nruns * (kGridRun / kWave);        // run counts, offsets, one 64-bit mask per 64 cells
}

#define SE3MPC_DEFINE_VOXEL_API(SUF, R)                                                                                    \
  extern "C" int se3mpc_voxel_query_##SUF(const se3mpc_voxel_map* m, const R* positions, int M, double* occupancy,          \
                                          void* stream) {                                                                   \
    return voxel_query_impl<R>(m, positions, M, occupancy, stream);                                                         \
  }                                                                                                                         \
  extern "C" int se3mpc_voxel_trajectory_safe_##SUF(const se3mpc_voxel_map* m, const R* P, int B, int N, long long stride,  \
                                                    double margin, double threshold, int32_t* safe, int32_t* first,         \
                                                    void* stream) {                                                         \
    return voxel_trajectory_safe_impl<R>(m, P, B, N, stride, margin, threshold, safe, first, stream);                       \
  }                                                                                                                         \
  extern "C" int se3mpc_voxel_local_spheres_##SUF(const se3mpc_voxel_map* m, const double* centre, double size,             \
                                                  double threshold, int target, double radius, R* spheres, int cap,         \
                                                  int32_t* count, int32_t* workspace, void* stream) {                       \
    return voxel_local_spheres_impl<R>(m, centre, size, threshold, target, radius, spheres, cap, count, workspace, stream); \
  }

SE3MPC_DEFINE_VOXEL_API(f32, float)
SE3MPC_DEFINE_VOXEL_API(f64, double)
