// Developer probe (GPU box): what limits write-heavy lane-layout streams?  [rows][B] float matrix, one column per lane;
// a wavefront writes (or copies) RW consecutive rows of its 64 columns; grid = (column blocks, row chunks).
// Usage: probe_rows [rows=270] [log2B=20] [ld_pad=0]      (leading dimension = B + ld_pad columns: does the 4 MB row stride of B = 1 M alias channels / pages?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 write-only, 1 copy, 2 read-only
__global__ void __launch_bounds__(64) rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, size_t ld, int rw, float* sink) {
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  const size_t col = (size_t)blk * 64 + threadIdx.x;
  const int r0 = blockIdx.y * rw;
  float acc = 0.f;
#pragma unroll 8
  for (int r = r0; r < r0 + rw && r < rows; ++r) {
    if (MODE == 0) __builtin_nontemporal_store((float)r, dst + (size_t)r * ld + col);
    else if (MODE == 1) __builtin_nontemporal_store(__builtin_nontemporal_load(src + (size_t)r * ld + col) + 1.f, dst + (size_t)r * ld + col);
    else acc += __builtin_nontemporal_load(src + (size_t)r * ld + col);
  }
  if (MODE == 2 && acc == 123.456f) *sink = acc;
}

template <int MODE>   // vf4 per lane: 256 columns per wavefront
__global__ void __launch_bounds__(64) rows4_kernel(const vf4* __restrict__ src, vf4* __restrict__ dst, int rows, size_t ld4, int rw, float* sink) {
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  const size_t col = (size_t)blk * 64 + threadIdx.x;
  const int r0 = blockIdx.y * rw;
  float acc = 0.f;
#pragma unroll 8
  for (int r = r0; r < r0 + rw && r < rows; ++r) {
    if (MODE == 0) { vf4 v = {(float)r, 1.f, 2.f, 3.f}; __builtin_nontemporal_store(v, dst + (size_t)r * ld4 + col); }
    else if (MODE == 1) { vf4 v = __builtin_nontemporal_load(src + (size_t)r * ld4 + col); v.x += 1.f; __builtin_nontemporal_store(v, dst + (size_t)r * ld4 + col); }
    else { vf4 v = __builtin_nontemporal_load(src + (size_t)r * ld4 + col); acc += v.x + v.w; }
  }
  if (MODE == 2 && acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 270;
  const size_t B = (size_t)1 << (argc > 2 ? atoi(argv[2]) : 20);
  const size_t LD = B + (size_t)(argc > 3 ? atoi(argv[3]) : 0);
  float *src, *dst, *sink;
  hipMalloc(&src, rows * LD * 4); hipMalloc(&dst, rows * LD * 4); hipMalloc(&sink, 4);
  hipMemset(src, 0, rows * LD * 4);
  printf("rows %d  B %zu  ld %zu (row stride %zu bytes)\n", rows, B, LD, LD * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"write", "copy", "read"};
  for (int mode = 0; mode < 3; ++mode)
    for (int rw : {rows, 30, 10}) {
      dim3 grid(B / 64, (rows + rw - 1) / rw);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(rows_kernel<0>, grid, dim3(64), 0, 0, src, dst, rows, LD, rw, sink);
        else if (mode == 1) hipLaunchKernelGGL(rows_kernel<1>, grid, dim3(64), 0, 0, src, dst, rows, LD, rw, sink);
        else hipLaunchKernelGGL(rows_kernel<2>, grid, dim3(64), 0, 0, src, dst, rows, LD, rw, sink);
      };
      for (int i = 0; i < 2; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double bytes = (double)rows * B * 4 * (mode == 1 ? 2 : 1);
      printf("%-5s rows/wave %3d  grid.y %3d  %8.1f us  %6.3f TB/s\n", names[mode], rw, grid.y, ms * 1e3, bytes / ms / 1e9);
    }
  for (int mode = 0; mode < 3; ++mode)
    for (int rw : {rows, 30, 10}) {
      dim3 grid(B / 256, (rows + rw - 1) / rw);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(rows4_kernel<0>, grid, dim3(64), 0, 0, (const vf4*)src, (vf4*)dst, rows, LD / 4, rw, sink);
        else if (mode == 1) hipLaunchKernelGGL(rows4_kernel<1>, grid, dim3(64), 0, 0, (const vf4*)src, (vf4*)dst, rows, LD / 4, rw, sink);
        else hipLaunchKernelGGL(rows4_kernel<2>, grid, dim3(64), 0, 0, (const vf4*)src, (vf4*)dst, rows, LD / 4, rw, sink);
      };
      for (int i = 0; i < 2; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double bytes = (double)rows * B * 4 * (mode == 1 ? 2 : 1);
      printf("x4 %-5s rows/wave %3d  grid.y %3d  %8.1f us  %6.3f TB/s\n", names[mode], rw, grid.y, ms * 1e3, bytes / ms / 1e9);
    }
  // contiguous fill / copy with vf4 (the shape torch's fill_ / copy_ use)
  {
    const size_t n4 = (size_t)rows * B / 4;
    dim3 grid((n4 + 255) / 256 / 4);
    for (int mode = 0; mode < 2; ++mode) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(rows4_kernel<0>, dim3(B / 256, 1), dim3(64), 0, 0, (const vf4*)src, (vf4*)dst, rows, B / 4, rows, sink);
      };
      (void)launch;
    }
  }
  return 0;
}
