// se3mpc_wave_ops.hpp -- 64-lane wavefront reductions on CDNA4 with DPP (no LDS traffic).
//
// A reduction is 4 in-row butterfly steps (quad_perm xor1, xor2, row_half_mirror, row_mirror:
// afterwards every lane of a 16-lane row holds the row result), then row_bcast:15 (each row's last
// lane into the next row) and row_bcast:31 (lane 31 into rows 2,3), after which lane 63 holds the
// wave result, which v_readlane broadcasts through an SGPR.  64-bit values move as two 32-bit DPP movs per step.
// (tests/emu shadows this header with a host implementation of the same functions.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace se3mpc {

constexpr int kDppQuadXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;

// One DPP move without a tied `old` operand: every lane whose source exists takes it, a lane without one (rows 0 / 0-1 of the two
// row_bcast steps) reads 0 (bound_ctrl).  Only lane 63 is read at the end, and each of the six steps gives lane 63 a real source, so
// what the other lanes accumulate does not matter -- and the compiler no longer has to initialise a destination (the identity of the
// reduction) before each of the twelve moves of a 64-bit reduction, as it did with `update_dpp(identity, v, ...)` and row masks.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = dpp_u32<CTRL>((uint32_t)x);
  const uint32_t hi = dpp_u32<CTRL>((uint32_t)(x >> 32));
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double readlane63_f64(double v) {
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), 63);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// Sum over the 64 lanes, result in every lane.  All 64 lanes must be active.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<kDppQuadXor1>(v);
  v += dpp_f64<kDppQuadXor2>(v);
  v += dpp_f64<kDppRowHalfMirror>(v);
  v += dpp_f64<kDppRowMirror>(v);
  v += dpp_f64<kDppRowBcast15>(v);
  v += dpp_f64<kDppRowBcast31>(v);
  return readlane63_f64(v);
}

// K sums at once: the K butterfly chains are independent, so their DPP moves and adds interleave and the latency of one chain
// (6 dependent steps of 2 DPP moves + 1 f64 add) is paid once instead of K times.  Same operation order per value as wave_sum.
template <int K>
__device__ __forceinline__ void wave_sum_n(double (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppQuadXor1>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppQuadXor2>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowHalfMirror>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowMirror>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowBcast15>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowBcast31>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = readlane63_f64(v[k]);
}

__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<kDppQuadXor1>(v));
  v = fmax(v, dpp_f64<kDppQuadXor2>(v));
  v = fmax(v, dpp_f64<kDppRowHalfMirror>(v));
  v = fmax(v, dpp_f64<kDppRowMirror>(v));
  v = fmax(v, dpp_f64<kDppRowBcast15>(v));
  v = fmax(v, dpp_f64<kDppRowBcast31>(v));
  return readlane63_f64(v);
}

__device__ __forceinline__ double wave_min(double v) {
  v = fmin(v, dpp_f64<kDppQuadXor1>(v));
  v = fmin(v, dpp_f64<kDppQuadXor2>(v));
  v = fmin(v, dpp_f64<kDppRowHalfMirror>(v));
  v = fmin(v, dpp_f64<kDppRowMirror>(v));
  v = fmin(v, dpp_f64<kDppRowBcast15>(v));
  v = fmin(v, dpp_f64<kDppRowBcast31>(v));
  return readlane63_f64(v);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
  v = mn(v, dpp_u32<kDppQuadXor1>(v));
  v = mn(v, dpp_u32<kDppQuadXor2>(v));
  v = mn(v, dpp_u32<kDppRowHalfMirror>(v));
  v = mn(v, dpp_u32<kDppRowMirror>(v));
  v = mn(v, dpp_u32<kDppRowBcast15>(v));
  v = mn(v, dpp_u32<kDppRowBcast31>(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int wave_sum_i32(int v) {
  v += (int)dpp_u32<kDppQuadXor1>((uint32_t)v);
  v += (int)dpp_u32<kDppQuadXor2>((uint32_t)v);
  v += (int)dpp_u32<kDppRowHalfMirror>((uint32_t)v);
  v += (int)dpp_u32<kDppRowMirror>((uint32_t)v);
  v += (int)dpp_u32<kDppRowBcast15>((uint32_t)v);
  v += (int)dpp_u32<kDppRowBcast31>((uint32_t)v);
  return __builtin_amdgcn_readlane(v, 63);
}

// 64-bit mask of lanes whose predicate is true (s_cmp -> SGPR pair), and its lowest set lane.
__device__ __forceinline__ uint64_t wave_ballot(bool pred) { return __ballot(pred); }
__device__ __forceinline__ int first_lane(uint64_t mask) { return mask ? __builtin_ctzll(mask) : -1; }

// Value of lane `src` (wave-uniform src) in every lane.
__device__ __forceinline__ double wave_bcast(double v, int src) {
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ float wave_bcast(float v, int src) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), src));
}
__device__ __forceinline__ int wave_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// ---- sub-wavefront groups: G = 8, 16, 32 or 64 consecutive lanes (aligned) work on one problem, 64 / G problems per wavefront ----
// Every group operation below reads lanes of the caller's own group only, so groups may sit in different branches of divergent control
// flow (each group's lanes are all active or all inactive together).  G <= 16: in-row DPP butterfly (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror), after which every lane of the group holds the result.  G = 32: one more step across the two rows of a
// half with gfx950's v_permlane16_swap_b32 (odd rows of one operand <-> even rows of the other: with both operands = v every lane ends
// up with row0's value in one register and row1's in the other, in the same order for both rows, so both rows form bit-identical
// results).  G = 64 is the whole-wavefront form above (v_readlane: the result is wave-uniform and lives in SGPRs).
__device__ __forceinline__ void swap16_pair(uint32_t v, uint32_t& even_row, uint32_t& odd_row) {
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  even_row = r[0]; odd_row = r[1];
}
// (a, b) = (value of the even row, value of the odd row) of the caller's 32-lane half, in every lane of the half
__device__ __forceinline__ void swap16_f64(double v, double& a, double& b) {
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  uint32_t alo, blo, ahi, bhi;
  swap16_pair((uint32_t)x, alo, blo);
  swap16_pair((uint32_t)(x >> 32), ahi, bhi);
  a = __longlong_as_double((long long)(((uint64_t)ahi << 32) | alo));
  b = __longlong_as_double((long long)(((uint64_t)bhi << 32) | blo));
}

template <int G>
__device__ __forceinline__ double group_sum(double v) {
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group size");
  if constexpr (G == 64) return wave_sum(v);
  v += dpp_f64<kDppQuadXor1>(v);
  v += dpp_f64<kDppQuadXor2>(v);
  v += dpp_f64<kDppRowHalfMirror>(v);
  if constexpr (G >= 16) v += dpp_f64<kDppRowMirror>(v);
  if constexpr (G >= 32) { double a, b; swap16_f64(v, a, b); v = a + b; }
  return v;
}
template <int G, int K>
__device__ __forceinline__ void group_sum_n(double (&v)[K]) {
  if constexpr (G == 64) { wave_sum_n<K>(v); return; }
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppQuadXor1>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppQuadXor2>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowHalfMirror>(v[k]);
  if constexpr (G >= 16) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f64<kDppRowMirror>(v[k]);
  }
  if constexpr (G >= 32) {
#pragma unroll
    for (int k = 0; k < K; ++k) { double a, b; swap16_f64(v[k], a, b); v[k] = a + b; }
  }
}
template <int G>
__device__ __forceinline__ double group_max(double v) {
  if constexpr (G == 64) return wave_max(v);
  v = fmax(v, dpp_f64<kDppQuadXor1>(v));
  v = fmax(v, dpp_f64<kDppQuadXor2>(v));
  v = fmax(v, dpp_f64<kDppRowHalfMirror>(v));
  if constexpr (G >= 16) v = fmax(v, dpp_f64<kDppRowMirror>(v));
  if constexpr (G >= 32) { double a, b; swap16_f64(v, a, b); v = fmax(a, b); }
  return v;
}
template <int G>
__device__ __forceinline__ double group_min(double v) {
  if constexpr (G == 64) return wave_min(v);
  v = fmin(v, dpp_f64<kDppQuadXor1>(v));
  v = fmin(v, dpp_f64<kDppQuadXor2>(v));
  v = fmin(v, dpp_f64<kDppRowHalfMirror>(v));
  if constexpr (G >= 16) v = fmin(v, dpp_f64<kDppRowMirror>(v));
  if constexpr (G >= 32) { double a, b; swap16_f64(v, a, b); v = fmin(a, b); }
  return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
  if constexpr (G == 64) return wave_min_u32(v);
  auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
  v = mn(v, dpp_u32<kDppQuadXor1>(v));
  v = mn(v, dpp_u32<kDppQuadXor2>(v));
  v = mn(v, dpp_u32<kDppRowHalfMirror>(v));
  if constexpr (G >= 16) v = mn(v, dpp_u32<kDppRowMirror>(v));
  if constexpr (G >= 32) { uint32_t a, b; swap16_pair(v, a, b); v = mn(a, b); }
  return v;
}
template <int G>
__device__ __forceinline__ int group_sum_i32(int v) {
  if constexpr (G == 64) return wave_sum_i32(v);
  v += (int)dpp_u32<kDppQuadXor1>((uint32_t)v);
  v += (int)dpp_u32<kDppQuadXor2>((uint32_t)v);
  v += (int)dpp_u32<kDppRowHalfMirror>((uint32_t)v);
  if constexpr (G >= 16) v += (int)dpp_u32<kDppRowMirror>((uint32_t)v);
  if constexpr (G >= 32) { uint32_t a, b; swap16_pair((uint32_t)v, a, b); v = (int)(a + b); }
  return v;
}
// bit i = predicate of lane i of the caller's group
template <int G>
__device__ __forceinline__ uint64_t group_ballot(bool pred) {
  const uint64_t m = __ballot(pred);
  if constexpr (G == 64) return m;
  return (m >> (threadIdx.x & 63u & ~(unsigned)(G - 1))) & ((1ull << G) - 1ull);
}
// value of lane `src` of the caller's group (src identical in the group's lanes) in every lane of the group
template <int G>
__device__ __forceinline__ int group_bcast(int v, int src) {
  if constexpr (G == 64) return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
  return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63u & ~(unsigned)(G - 1)) + (unsigned)src) << 2), v);
}
template <int G>
__device__ __forceinline__ double group_bcast(double v, int src) {
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)group_bcast<G>((int)(uint32_t)x, src);
  const uint32_t hi = (uint32_t)group_bcast<G>((int)(uint32_t)(x >> 32), src);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// value of lane `src` of the caller's group, src chosen per lane (ds_bpermute_b32: the LDS crossbar, no LDS memory).  Every lane of the
// group must call it (a lane that is switched off hands out 0).
template <int G>
__device__ __forceinline__ double group_gather(double v, int src) {
  const int addr = (int)(((threadIdx.x & 63u & ~(unsigned)(G - 1)) + (unsigned)src) << 2);
  const uint64_t x = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(x >> 32));
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// LDS hand-off inside one wavefront (one lane of a group writes, the group's other lanes read): DS operations of a wavefront execute in
// order, so all that is needed is that the compiler keeps the order (and waits for outstanding returns).  Unlike __syncthreads() this
// may be called by groups in different branches.
template <int G>
__device__ __forceinline__ void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// ---- lane-layout memory access through a buffer resource (SRSRC) ---------------------------
// address = base (descriptor, 4 SGPRs) + soff (SGPR: row * ld * sizeof(R), wave-uniform) + voff
// (ONE VGPR: lane * sizeof(R)) -> `buffer_load_dword v, v_off, s[rsrc], s_row offen`.  No 64-bit
// vector address arithmetic and no per-row address VGPRs, so all N row loads of a sweep can be
// in flight at once.  Offsets are 32-bit: rows*ld*sizeof(R) < 4 GiB is checked on the host.
template <typename R>
struct LaneBuf {
  __amdgpu_buffer_rsrc_t rsrc;
};
template <typename R>
__device__ __forceinline__ LaneBuf<R> lane_buf(const R* base) {
  LaneBuf<R> b;
  // raw buffer (stride 0), num_records = max, DWORD3 = 0x00020000 (gfx9 family: 32-bit data format)
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<R*>(base), 0, 0xFFFFFFFF, 0x00020000);
  return b;
}
// AUX: cache-policy bits of the buffer instruction (0 = default, 2 = nt: streamed once, do not keep).
template <int AUX = 0>
__device__ __forceinline__ float lane_ld(const LaneBuf<float>& b, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.rsrc, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ double lane_ld(const LaneBuf<double>& b, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void lane_st(const LaneBuf<float>& b, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.rsrc, voff, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void lane_st(const LaneBuf<double>& b, unsigned voff, unsigned soff, double v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), b.rsrc, voff, soff, AUX);
}

// ---- 16 bytes per lane: four consecutive trajectories of one row (streaming kernels at saturating batches, eval_kernels.hip)
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vf4 lane_ld4(const vf4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void lane_st4(vf4* p, vf4 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ vf4 splat4(float x) { return (vf4)(x); }

// 1 / x: one v_rcp_f32 (1 ulp) for float where a kernel's float32 tolerance allows it, the IEEE quotient for double
// ---- matrix core: v_mfma_f32_16x16x4_f32, D(16x16) = A(16x4) B(4x16) + C, full float32 multiply-adds.  Lane l supplies a = A[l % 16][l / 16] and
// b = B[l / 16][l % 16]; register r of c / d is C / D[4 (l / 16) + r][l % 16].  Every lane of the wavefront must be active.
__device__ __forceinline__ vf4 mfma_16x16x4_f32(float a, float b, vf4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// value of lane (l ^ mask) of the same wavefront
__device__ __forceinline__ float wave_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

__device__ __forceinline__ float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_approx(double x) { return 1.0 / x; }

// Tell the compiler a value is the same in every lane (moves it to an SGPR).
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

}  // namespace se3mpc
