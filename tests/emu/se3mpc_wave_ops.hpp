// TEST-ONLY host stand-in for dart_planner_amd/csrc/se3mpc_wave_ops.hpp (same API, lanes are
// std::threads that rendezvous in emu::exchange).  The DPP code itself is validated on the GPU by
// se3mpc_selftest (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace se3mpc {

template <typename T, typename OP>
inline T emu_wave_reduce(T v, OP op) {
  // gather all 64 lanes in lane order so the summation order is deterministic
  T acc{};
  const int base = (int)threadIdx.x - (int)(threadIdx.x % 64);
  for (int l = 0; l < 64; ++l) {
    T x = ::emu::exchange(v, base + l);
    acc = (l == 0) ? x : op(acc, x);
  }
  return acc;
}
inline double wave_sum(double v) { return emu_wave_reduce(v, [](double a, double b) { return a + b; }); }
template <int K>
inline void wave_sum_n(double (&v)[K]) { for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]); }
inline double wave_max(double v) { return emu_wave_reduce(v, [](double a, double b) { return a > b ? a : b; }); }
inline double wave_min(double v) { return emu_wave_reduce(v, [](double a, double b) { return a < b ? a : b; }); }
inline uint32_t wave_min_u32(uint32_t v) { return emu_wave_reduce(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
inline int wave_sum_i32(int v) { return emu_wave_reduce(v, [](int a, int b) { return a + b; }); }
inline uint64_t wave_ballot(bool pred) {
  uint64_t m = 0;
  const int base = (int)threadIdx.x - (int)(threadIdx.x % 64);
  for (int l = 0; l < 64; ++l) m |= (uint64_t)(::emu::exchange<int>(pred ? 1 : 0, base + l) & 1) << l;
  return m;
}
inline int first_lane(uint64_t mask) { return mask ? __builtin_ctzll(mask) : -1; }
inline double wave_bcast(double v, int src) { return ::emu::exchange(v, (int)threadIdx.x - (int)(threadIdx.x % 64) + src); }
inline float wave_bcast(float v, int src) { return ::emu::exchange(v, (int)threadIdx.x - (int)(threadIdx.x % 64) + src); }
inline int wave_bcast(int v, int src) { return ::emu::exchange(v, (int)threadIdx.x - (int)(threadIdx.x % 64) + src); }
// ---- sub-wavefront groups (product header: DPP butterflies / v_permlane16_swap; here: rendezvous of the group's fibers only)
template <int G, typename T, typename OP>
inline T emu_group_reduce(T v, OP op) {
  const int base = (int)threadIdx.x - (int)(threadIdx.x % G);
  T acc{};
  for (int l = 0; l < G; ++l) {
    T x = ::emu::exchange_group(v, base + l, base, G);
    acc = (l == 0) ? x : op(acc, x);
  }
  return acc;
}
template <int G> inline double group_sum(double v) { return emu_group_reduce<G>(v, [](double a, double b) { return a + b; }); }
template <int G, int K> inline void group_sum_n(double (&v)[K]) { for (int k = 0; k < K; ++k) v[k] = group_sum<G>(v[k]); }
template <int G> inline double group_max(double v) { return emu_group_reduce<G>(v, [](double a, double b) { return a > b ? a : b; }); }
template <int G> inline double group_min(double v) { return emu_group_reduce<G>(v, [](double a, double b) { return a < b ? a : b; }); }
template <int G> inline uint32_t group_min_u32(uint32_t v) { return emu_group_reduce<G>(v, [](uint32_t a, uint32_t b) { return a < b ? a : b; }); }
template <int G> inline int group_sum_i32(int v) { return emu_group_reduce<G>(v, [](int a, int b) { return a + b; }); }
template <int G>
inline uint64_t group_ballot(bool pred) {
  uint64_t m = 0;
  const int base = (int)threadIdx.x - (int)(threadIdx.x % G);
  for (int l = 0; l < G; ++l) m |= (uint64_t)(::emu::exchange_group<int>(pred ? 1 : 0, base + l, base, G) & 1) << l;
  return m;
}
template <int G> inline int group_bcast(int v, int src) { const int base = (int)threadIdx.x - (int)(threadIdx.x % G); return ::emu::exchange_group(v, base + src, base, G); }
template <int G> inline double group_bcast(double v, int src) { const int base = (int)threadIdx.x - (int)(threadIdx.x % G); return ::emu::exchange_group(v, base + src, base, G); }
template <int G> inline double group_gather(double v, int src) { const int base = (int)threadIdx.x - (int)(threadIdx.x % G); return ::emu::exchange_group(v, base + src, base, G); }
template <int G> inline void group_sync() { ::emu::sync_group((int)threadIdx.x - (int)(threadIdx.x % G), G); }

template <typename R>
struct LaneBuf { const char* base; };
template <typename R>
inline LaneBuf<R> lane_buf(const R* base) { return LaneBuf<R>{reinterpret_cast<const char*>(base)}; }
template <int AUX = 0, typename R>
inline R lane_ld(const LaneBuf<R>& b, unsigned voff, unsigned soff) {
  return *reinterpret_cast<const R*>(b.base + (size_t)soff + voff);
}
template <int AUX = 0, typename R>
inline void lane_st(const LaneBuf<R>& b, unsigned voff, unsigned soff, R v) {
  *reinterpret_cast<R*>(const_cast<char*>(b.base) + (size_t)soff + voff) = v;
}
inline int wave_uniform(int v) { return v; }
inline int lane_id() { return (int)(threadIdx.x & 63u); }


// ---- 16 bytes per lane (product header: a clang ext_vector_type + non-temporal builtins)
struct vf4 {
  float v[4];
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};
#define SE3MPC_EMU_VF4_OP(OP)                                                                                         \
  inline vf4 operator OP(const vf4& a, const vf4& b) { vf4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] OP b.v[i]; return r; } \
  inline vf4 operator OP(float a, const vf4& b) { vf4 r; for (int i = 0; i < 4; ++i) r.v[i] = a OP b.v[i]; return r; }        \
  inline vf4 operator OP(const vf4& a, float b) { vf4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] OP b; return r; }
SE3MPC_EMU_VF4_OP(+)
SE3MPC_EMU_VF4_OP(-)
SE3MPC_EMU_VF4_OP(*)
SE3MPC_EMU_VF4_OP(/)
#undef SE3MPC_EMU_VF4_OP
// matrix core (product header: v_mfma_f32_16x16x4_f32): the wavefront's 64 lanes deposit their operands, rendezvous, and every lane forms its four outputs
inline float g_mfma_a[1024], g_mfma_b[1024];
inline vf4 mfma_16x16x4_f32(float a, float b, vf4 c) {
  const int me = (int)threadIdx.x, l = me % 64, base = me - l;
  g_mfma_a[me] = a; g_mfma_b[me] = b;
  ::emu::sync_group(base, 64);
  vf4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l / 16) + r, j = l % 16;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = __builtin_fmaf(g_mfma_a[base + 16 * k + i], g_mfma_b[base + 16 * k + j], acc);
    d[r] = acc;
  }
  ::emu::sync_group(base, 64);
  return d;
}
inline float wave_xor(float v, int mask) { const int me = (int)threadIdx.x, l = me % 64, base = me - l; return ::emu::exchange_group(v, base + (l ^ mask), base, 64); }
inline float rcp_approx(float x) { return 1.0f / x; }
inline double rcp_approx(double x) { return 1.0 / x; }
inline vf4 lane_ld4(const vf4* p) { return *p; }
inline void lane_st4(vf4* p, vf4 v) { *p = v; }
inline vf4 splat4(float x) { return vf4{{x, x, x, x}}; }
}  // namespace se3mpc
