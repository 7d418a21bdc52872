// tests/emu/hip/hip_runtime.h -- TEST-ONLY host emulation of the tiny slice of HIP that
// dart_planner_amd/csrc/*.hip uses.  It shadows <hip/hip_runtime.h> when tests/emu/build_emu.py
// compiles the UNMODIFIED product sources with g++ into tests/emu/libse3mpc_emu.so, so that the
// kernels' arithmetic and the C-ABI argument handling can be checked against the oracle on a
// machine without a GPU (`pytest -m "not gpu"`).  One std::thread per lane of a workgroup,
// workgroups run one after another; __syncthreads()/__shfl*() rendezvous on a std::barrier.
// Never shipped, never loaded by the product (dart_planner_amd loads libse3mpc.so only).
#pragma once
#include <algorithm>
#include <barrier>
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
using std::isinf; using std::isnan; using std::sqrt;

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
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }

namespace emu {
inline thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
inline std::barrier<>* g_barrier = nullptr;
inline unsigned char g_dyn_lds[163840] __attribute__((aligned(64)));
inline unsigned long long g_slots[1024];

inline void sync() { g_barrier->arrive_and_wait(); }

template <typename T>
inline T exchange(T v, int src_lane_in_block) {
  static_assert(sizeof(T) <= 8, "emu shuffle: <= 8 bytes");
  const unsigned me = t_threadIdx.x;
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  g_slots[me] = raw;
  sync();
  unsigned long long got = (src_lane_in_block >= 0 && src_lane_in_block < (int)t_blockDim.x) ? g_slots[src_lane_in_block] : raw;
  sync();
  T out;
  std::memcpy(&out, &got, sizeof(T));
  return out;
}

template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args) {
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      std::barrier<> bar(block.x);
      g_barrier = &bar;
      std::vector<std::thread> th;
      th.reserve(block.x);
      for (unsigned t = 0; t < block.x; ++t)
        th.emplace_back([=, &bar]() {
          t_threadIdx = dim3(t); t_blockIdx = dim3(bx, by); t_blockDim = block; t_gridDim = grid;
          kernel(args...);
          bar.arrive_and_drop();
        });
      for (auto& x : th) x.join();
    }
  g_barrier = nullptr;
}
}  // namespace emu

#define threadIdx (::emu::t_threadIdx)
#define blockIdx (::emu::t_blockIdx)
#define blockDim (::emu::t_blockDim)
#define gridDim (::emu::t_gridDim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) ::emu::launch(kernel, grid, block, __VA_ARGS__)

inline void __syncthreads() { ::emu::sync(); }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

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
  static std::mutex* m = new std::mutex;   // blocks run sequentially; lanes of a block may race
  std::lock_guard<std::mutex> g(*m);
  unsigned long long old = *p;
  if (v < old) *p = v;
  return old;
}
