// tests/emu/hip/hip_runtime.h -- TEST-ONLY host emulation of the tiny slice of HIP that
// dart_planner_amd/csrc/*.hip uses.  It shadows <hip/hip_runtime.h> when tests/emu/build_emu.py
// compiles the UNMODIFIED product sources with g++ into tests/emu/libse3mpc_emu.so, so that the
// kernels' arithmetic and the C-ABI argument handling can be checked against the oracle on a
// machine without a GPU (`pytest -m "not gpu"`).  One user-level fiber (ucontext) per lane of a
// workgroup, workgroups run one after another; __syncthreads()/__shfl*() are rendezvous points.
// Never shipped, never loaded by the product (dart_planner_amd loads libse3mpc.so only).
#pragma once
#include <algorithm>
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::emu::g_dyn_lds);

using std::asin; using std::atan2; using std::fabs; using std::fmax; using std::fmin;
using std::isinf; using std::isnan; using std::sqrt; using std::sin; using std::cos; using std::acos; using std::ldexp;
// v_med3_f32: the median of three
inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
struct hipDeviceProp_t { char gcnArchName[256]; };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->gcnArchName[0] = 0; return hipSuccess; }
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }

namespace emu {
// Lanes are user-level fibers (ucontext) scheduled round-robin on the calling OS thread: a lane runs
// until it reaches a rendezvous (__syncthreads / a cross-lane op) or returns, then yields.  No OS
// scheduling is involved, so a 64-lane workgroup with thousands of rendezvous runs in milliseconds.
struct Fiber {
  ucontext_t ctx;
  dim3 tid;
  bool done = false;
  std::vector<unsigned char> stack;
};
inline std::vector<Fiber> g_fibers;
inline ucontext_t g_main;
inline int g_cur = 0, g_live = 0;
inline unsigned g_arrived = 0, g_generation = 0;
inline dim3 g_blockIdx, g_blockDim, g_gridDim;
inline std::function<void()> g_body;
inline unsigned char g_dyn_lds[163840] __attribute__((aligned(64)));
inline unsigned long long g_slots[1024];

inline void yield_to_next() {
  const int n = (int)g_fibers.size();
  int nxt = g_cur;
  for (int i = 1; i <= n; ++i) {
    const int c = (g_cur + i) % n;
    if (!g_fibers[c].done) { nxt = c; break; }
  }
  if (nxt == g_cur) return;
  const int prev = g_cur;
  g_cur = nxt;
  swapcontext(&g_fibers[prev].ctx, &g_fibers[nxt].ctx);
}

// barrier over the lanes that are still running (a lane that returned has dropped out, as
// std::barrier::arrive_and_drop would)
inline void sync() {
  const unsigned gen = g_generation;
  if (++g_arrived >= (unsigned)g_live) { g_arrived = 0; ++g_generation; return; }
  while (g_generation == gen) yield_to_next();
}

inline void fiber_entry() {
  g_body();
  Fiber& f = g_fibers[g_cur];
  f.done = true;
  --g_live;
  if (g_live > 0 && g_arrived >= (unsigned)g_live) { g_arrived = 0; ++g_generation; }   // release waiters
  if (g_live == 0) { swapcontext(&f.ctx, &g_main); return; }
  const int n = (int)g_fibers.size();
  for (int i = 1; i <= n; ++i) {
    const int c = (g_cur + i) % n;
    if (!g_fibers[c].done) { g_cur = c; setcontext(&g_fibers[c].ctx); }
  }
}

template <typename T>
inline T exchange(T v, int src_lane_in_block) {
  static_assert(sizeof(T) <= 8, "emu shuffle: <= 8 bytes");
  const unsigned me = g_fibers[g_cur].tid.x;
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  g_slots[me] = raw;
  sync();
  unsigned long long got = (src_lane_in_block >= 0 && src_lane_in_block < (int)g_blockDim.x) ? g_slots[src_lane_in_block] : raw;
  sync();
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}

// rendezvous of the `size` lanes [base, base + size) only: groups of a wavefront that sit in different branches of divergent control
// flow (the packed solver: one problem per aligned group of 8 / 16 / 32 lanes).  All lanes of a group run the same control flow and
// return together, so a group's count is simply `size`.
inline unsigned g_group_arrived[1024], g_group_generation[1024];
inline void sync_group(int base, int size) {
  if (size >= (int)g_blockDim.x && base == 0 && g_live == (int)g_blockDim.x) { sync(); return; }
  const unsigned gen = g_group_generation[base];
  if (++g_group_arrived[base] >= (unsigned)size) { g_group_arrived[base] = 0; ++g_group_generation[base]; return; }
  while (g_group_generation[base] == gen) yield_to_next();
}
template <typename T>
inline T exchange_group(T v, int src_lane_in_block, int base, int size) {
  static_assert(sizeof(T) <= 8, "emu shuffle: <= 8 bytes");
  const unsigned me = g_fibers[g_cur].tid.x;
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  g_slots[me] = raw;
  sync_group(base, size);
  unsigned long long got = (src_lane_in_block >= base && src_lane_in_block < base + size) ? g_slots[src_lane_in_block] : raw;
  sync_group(base, size);
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}

// Dynamic LDS of a launch: the requested bytes are poisoned before every workgroup (a read of LDS nobody wrote shows up as garbage, as on
// the device) and a canary page behind them must come back untouched (a store past the size the launcher asked for aborts the test run).
constexpr size_t kLdsCanary = 4096;
inline void lds_arm(size_t shmem) {
  if (shmem + kLdsCanary > sizeof(g_dyn_lds)) shmem = sizeof(g_dyn_lds) - kLdsCanary;
  std::memset(g_dyn_lds, 0x7B, shmem);
  std::memset(g_dyn_lds + shmem, 0xA5, kLdsCanary);
}
inline void lds_check(size_t shmem, const char* what) {
  if (shmem + kLdsCanary > sizeof(g_dyn_lds)) shmem = sizeof(g_dyn_lds) - kLdsCanary;
  for (size_t i = 0; i < kLdsCanary; ++i) {
    if (g_dyn_lds[shmem + i] != 0xA5) {
      std::fprintf(stderr, "emu: %s stored %zu bytes past its %zu bytes of dynamic LDS\n", what, i + 1, shmem);
      std::abort();
    }
  }
}

template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
  constexpr size_t kStack = 1 << 20;
  g_blockDim = block; g_gridDim = grid;
  g_body = [=]() { kernel(args...); };
  if (g_fibers.size() != block.x) g_fibers.assign(block.x, Fiber());
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      g_blockIdx = dim3(bx, by);
      g_arrived = 0; g_generation = 0; g_live = (int)block.x;
      std::memset(g_group_arrived, 0, sizeof(g_group_arrived)); std::memset(g_group_generation, 0, sizeof(g_group_generation));
      for (unsigned t = 0; t < block.x; ++t) {
        Fiber& f = g_fibers[t];
        f.tid = dim3(t); f.done = false;
        if (f.stack.size() != kStack) f.stack.resize(kStack);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &g_main;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
      }
      g_cur = 0;
      lds_arm(shmem);
      swapcontext(&g_main, &g_fibers[0].ctx);
      lds_check(shmem, __PRETTY_FUNCTION__);
    }
}
}  // namespace emu

#define threadIdx (::emu::g_fibers[::emu::g_cur].tid)
#define blockIdx (::emu::g_blockIdx)
#define blockDim (::emu::g_blockDim)
#define gridDim (::emu::g_gridDim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) ::emu::launch(kernel, grid, block, (size_t)(shmem), __VA_ARGS__)

inline void __syncthreads() { ::emu::sync(); }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

template <typename T>
inline T __shfl_down(T v, int off, int width = 64) {
  const int me = (int)threadIdx.x, lane = me % width;
  return ::emu::exchange(v, (lane + off < width) ? me + off : me);
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  const int me = (int)threadIdx.x, lane = me % width;
  return ::emu::exchange(v, me - lane + ((lane ^ mask) % width));
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  const int me = (int)threadIdx.x, lane = me % width;
  return ::emu::exchange(v, me - lane + (src % width));
}
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
  unsigned long long old = *p;             // fibers are cooperative: no preemption inside this function
  if (v < old) *p = v;
  return old;
}
inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long expect, unsigned long long desired) {
  unsigned long long old = *p;
  if (old == expect) *p = desired;
  return old;
}
inline int atomicAdd(int* p, int v) { int old = *p; *p = old + v; return old; }
inline int atomicCAS(int* p, int expect, int desired) { int old = *p; if (old == expect) *p = desired; return old; }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long old = *p; *p = old | v; return old; }
inline void __threadfence() {}
inline void __threadfence_system() {}
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
