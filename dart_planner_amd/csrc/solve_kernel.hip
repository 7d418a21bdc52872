// solve_kernel.hip -- the batched solve's kernel (one solve per group and launch), launch logic and C entry points; the solver itself
// (one solve by the G lanes of a group) is solve_device.hpp.
#include "solve_device.hpp"

namespace se3mpc {

// 2nd launch-bounds argument = wavefronts per SIMD the register allocation must leave room for: two
// resident wavefronts per SIMD (<= 256 VGPR+AGPR each) overlap each other's DPP/LDS latencies.
template <typename IO, int G>
__global__ void __launch_bounds__(64, SE3MPC_SOLVE_WAVES)
solve_kernel(SolveDev q, int B, const IO* __restrict__ p0g, const IO* __restrict__ v0g, const IO* __restrict__ goalg,
             const IO* __restrict__ x0g, IO* __restrict__ Xg, se3mpc_solve_info* __restrict__ infog,
             IO* __restrict__ accg, IO* __restrict__ attg, IO* __restrict__ ratesg, IO* __restrict__ thrustg,
             unsigned long long* doneg, unsigned long long ticket) {
  HIP_DYNAMIC_SHARED(unsigned char, lds_raw)
  constexpr int P = kWave / G, J = kSlots;
  const int lane = lane_id();
  const int k = lane & (G - 1);            // horizon step owned by this lane
  const int grp = lane / G;                // problem slot inside the wavefront
  const int pb = blockIdx.x * P + grp;     // problem index
  if (pb >= B) return;                     // (a whole group leaves together)
  // (n = 9N, N, n3 = 3N are declared by the solver body below)
  // Two-tier memory: the first launch gives every problem LDS for `mlds` L-BFGS pairs (2-4: all a solve with the reference's options
  // ever stores -- it stops after 1-3 iterations, i.e. at most two updates) so that two wavefronts per SIMD fit a CU's LDS; a
  // problem that needs a pair more leaves with task = SE3MPC_TASK_OVERFLOW and is re-solved from scratch by the second launch
  // (mlds = maxcor, only_overflow = 1), in which every other group exits at once.
  if (q.only_overflow && infog[pb].task != SE3MPC_TASK_OVERFLOW) return;
  double goal[3] = {0.0, 0.0, 0.0};
  if (q.has_goal) {
#pragma unroll
    for (int a = 0; a < 3; ++a) goal[a] = (double)goalg[pb * 3 + a];
  }
  double ps[3] = {0.0, 0.0, 0.0}, vs[3] = {0.0, 0.0, 0.0};
  if (x0g == nullptr) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { ps[a] = (double)p0g[pb * 3 + a]; vs[a] = (double)v0g[pb * 3 + a]; }
  }
#ifdef SE3MPC_SOLVE_PROFILE
  __shared__ unsigned long long tsec[16];                     // in LDS (lane 0 adds): eight SGPR pairs of counters would change the register allocation measured
  if (lane < 16) tsec[lane] = 0;
  __syncthreads();
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
#endif
  const IO* x0row = x0g != nullptr ? x0g + (size_t)pb * q.n : nullptr;
  const bool cold = x0row == nullptr;
  double x[J];
#include "solve_body.inc"

  SE3MPC_TICK(5)
  // ===================================================================== results
  if (live) {
#pragma unroll
    for (int j = 0; j < J; ++j) Xg[(size_t)pb * n + (j / 3) * n3 + 3 * k + (j % 3)] = (IO)x[j];
  }
  if (k == 0 && infog != nullptr) {
    se3mpc_solve_info r;
    r.fun = f; r.nit = nit; r.nfev = nfev; r.status = status; r.task = task;
    infog[pb] = r;
  }
  // a restart solve asks for x and info only (every trajectory output null): nothing to extract; a closed loop that reads the plan in
  // place wants the accelerations but no attitudes / rates / thrust magnitudes: no frames to build
  const bool want_frames = attg != nullptr || ratesg != nullptr || thrustg != nullptr;
  // ---- _extract_solution_from_result (planner.py:582-654): lane k = step k holds T_k in its own registers
  const double t0 = x[6], t1 = x[7], t2 = x[8];
  if (live && accg != nullptr) {
    const size_t o = (size_t)pb * n3 + 3 * k;
    accg[o] = (IO)(t0 / q.mass); accg[o + 1] = (IO)(t1 / q.mass); accg[o + 2] = (IO)(t2 / q.mass - q.grav);
  }
  if (want_frames) {
  const double mag = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
  const bool valid = live && mag > 1e-6;
  double b1[3] = {0, 0, 0}, b2[3] = {0, 0, 0}, b3[3] = {0, 0, 0};
  double roll = 0.0, pitch = 0.0, yaw = 0.0;
  if (valid) {
    b3[0] = t0 / mag; b3[1] = t1 / mag; b3[2] = t2 / mag;
    b1[0] = 0.0; b1[1] = -b3[2]; b1[2] = b3[1];
    const double n1 = sqrt(b1[1] * b1[1] + b1[2] * b1[2]);
    if (n1 > 1e-6) { b1[1] /= n1; b1[2] /= n1; } else { b1[0] = 1.0; b1[1] = 0.0; b1[2] = 0.0; }
    b2[0] = b3[1] * b1[2] - b3[2] * b1[1];
    b2[1] = b3[2] * b1[0] - b3[0] * b1[2];
    b2[2] = b3[0] * b1[1] - b3[1] * b1[0];
    roll = atan2(b2[2], b3[2]);
    pitch = asin(fmin(fmax(-b1[2], -1.0), 1.0));
    yaw = n1 > 1e-6 ? (b1[1] == 0.0 ? b1[1] : copysign(1.5707963267948966, b1[1])) : 0.0;   // atan2(b1y, b1x) with b1x exactly 0, or b1 = (1,0,0)
  }
  // prev_R of step k = R of the nearest earlier step with |T| > 1e-6 (planner.py:641-650): fetched from that step's lane
  const uint64_t vmask = group_ballot<G>(valid);
  const uint64_t below = vmask & ((k == 0) ? 0ull : (~0ull >> (64 - k)));
  const bool has_prev = valid && below != 0ull;
  const int pk = has_prev ? 63 - __builtin_clzll(below) : k;          // lane of the group that holds the previous frame (itself: unused)
  double q1[3], q2[3], q3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { q1[c] = group_gather<G>(b1[c], pk); q2[c] = group_gather<G>(b2[c], pk); q3[c] = group_gather<G>(b3[c], pk); }
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  if (has_prev) {
    double d1[3], d2[3], d3[3];
    for (int c = 0; c < 3; ++c) {
      d1[c] = (b1[c] - q1[c]) / q.dt;
      d2[c] = (b2[c] - q2[c]) / q.dt;
      d3[c] = (b3[c] - q3[c]) / q.dt;
    }
    w0 = b3[0] * d2[0] + b3[1] * d2[1] + b3[2] * d2[2];
    w1 = b1[0] * d3[0] + b1[1] * d3[1] + b1[2] * d3[2];
    w2 = b2[0] * d1[0] + b2[1] * d1[1] + b2[2] * d1[2];
  }
  if (live) {
    const size_t o = (size_t)pb * n3 + 3 * k;
    if (attg != nullptr) { attg[o] = (IO)roll; attg[o + 1] = (IO)pitch; attg[o + 2] = (IO)yaw; }
    if (ratesg != nullptr) { ratesg[o] = (IO)w0; ratesg[o + 1] = (IO)w1; ratesg[o + 2] = (IO)w2; }
    if (thrustg != nullptr) thrustg[(size_t)pb * N + k] = (IO)mag;
  }
  }   // want_frames
  // se3mpc_plan_host_*: a ONE-wavefront launch tells the waiting host it is done -- every store above made visible to the system, then
  // the ticket (the first lane of the wavefront belongs to problem 0, which exists; groups of a wavefront reach this point together)
  if (doneg != nullptr) {
    __threadfence_system();
    if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(doneg) = ticket;
  }
#ifdef SE3MPC_SOLVE_PROFILE
  __syncthreads();
  SE3MPC_TICK(6)
  if (lane == 0) tsec[7] = tlast - tstart;
  __syncthreads();
  if (lane == 0 && attg != nullptr && (size_t)3 * N * sizeof(IO) >= 128) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(attg + (size_t)pb * 3 * N);
    for (int i = 0; i < 16; ++i) dst[i] = tsec[i];
  }
#endif
}

static int g_solver_variant = 0;   // bit 0: published sequential Cauchy search also while the memory is empty; bits 8-15: forced group size (0 = automatic)


// Group size for problems of horizon N: the smallest group that holds the horizon packs the most problems into a wavefront -- the
// group-uniform algebra (line search, middle matrices, reductions) is then shared by 64 / G problems.  Measured on MI355X
// (profiles/r03_solve_group_probe.txt): the smallest group wins at every batch size from 1 K problems up, also where a wider group
// would put more wavefronts on the chip (N = 6: 1 K / 4 K / 8 K / 64 K problems take 32 / 37 / 48 / 180 us at G = 8 against
// 33 / 48 / 51 / 335 us at G = 16); the cost of co-resident problems diverging (a problem that stops after one iteration waits for
// its neighbours' third) is about an eighth of a wavefront's time.
static int solve_group_size(int N) {
  const int forced = (g_solver_variant >> 8) & 0xFF;
  const int G = N <= 8 ? 8 : (N <= 16 ? 16 : (N <= 32 ? 32 : 64));
  if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced >= G ? forced : G;
  return G;
}

// done != nullptr: the host-latency form (se3mpc_plan_host_*): the whole batch must fit ONE wavefront, the kernel stores `ticket`
// into *done (host-visible memory) as its last act.
template <typename IO>
int solve_impl(const se3mpc_params* p, int B, const IO* p0, const IO* v0, const IO* goal, const IO* x0, IO* X,
               se3mpc_solve_info* info, IO* acc, IO* att, IO* rates, IO* thrust, void* stream,
               unsigned long long* done = nullptr, unsigned long long ticket = 0) {
  if (p == nullptr) return SE3MPC_ERR_NULL;
  int rc = check_params_impl(p);
  if (rc) return rc;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!p0 || !v0 || !X || (p->has_goal && !goal)) return SE3MPC_ERR_NULL;
  SolveDev q = make_solve_dev(*p);
  q.seq_cauchy = g_solver_variant & 1;
  if (q.N > kWave) return SE3MPC_ERR_SHAPE;               // horizon <= 64 (check_params_impl says the same)
  hipStream_t s = (hipStream_t)stream;
  // (one problem alone in its wavefront runs the whole-wavefront form: its group-uniform values are SGPRs, its branches scalar)
  const int G = (B == 1 && ((g_solver_variant >> 8) & 0xFF) == 0) ? kWave : solve_group_size(q.N);
  const int waves = (int)(((long)B * G + kWave - 1) / kWave);
  if (done != nullptr && waves != 1) return SE3MPC_ERR_SHAPE;
#define SE3MPC_SOLVE_CASE(GG)                                                                                               \
  {                                                                                                                         \
    /* the attribute is raised once per instantiation and size, not on every plan (it is a driver call) */                  \
    static size_t lds_allowed = 64 * 1024;                                                                                  \
    if (solve_lds_bytes(q.mlds, GG, sizeof(IO)) > lds_allowed) {                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<IO, GG>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)solve_lds_bytes(q.mlds, GG, sizeof(IO)));                                               \
      lds_allowed = solve_lds_bytes(q.mlds, GG, sizeof(IO));                                                                \
    }                                                                                                                       \
  }                                                                                                                         \
  hipLaunchKernelGGL((solve_kernel<IO, GG>), dim3(waves), dim3(kWave), solve_lds_bytes(q.mlds, GG, sizeof(IO)), s, q, B, p0, \
                     v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket)
#define SE3MPC_SOLVE_LAUNCH()               \
  if (G == 8) { SE3MPC_SOLVE_CASE(8); }         \
  else if (G == 16) { SE3MPC_SOLVE_CASE(16); }  \
  else if (G == 32) { SE3MPC_SOLVE_CASE(32); }  \
  else { SE3MPC_SOLVE_CASE(64); }
  // Two tiers buy occupancy (two resident wavefronts per SIMD): the first gets the most pairs (<= 4, >= 2) whose LDS still lets
  // eight wavefronts share a CU's 160 KiB.  While every wavefront of the launch is resident at once even with the full memory's LDS
  // footprint (256 CUs x the wavefronts whose LDS fits a CU, at most one per SIMD) there is nothing to buy, and a single launch
  // with the full memory is quicker.
  constexpr size_t kLdsPerCu = 160 * 1024;
  int fast = 4;
  while (fast > 2 && solve_lds_bytes(fast, G, sizeof(IO)) > kLdsPerCu / 8) --fast;
  size_t full_per_cu = kLdsPerCu / solve_lds_bytes(q.m, G, sizeof(IO));
  if (full_per_cu > 4) full_per_cu = 4;
  const bool two_tier = info != nullptr && q.m > fast && (size_t)waves > 256 * full_per_cu;    // the tiers talk through info[].task
  q.mlds = two_tier ? fast : q.m;
  SE3MPC_SOLVE_LAUNCH();
  rc = launch_status("se3mpc_solve");
  if (rc != SE3MPC_OK || !two_tier) return rc;
  q.mlds = q.m;
  q.only_overflow = 1;
  SE3MPC_SOLVE_LAUNCH();
#undef SE3MPC_SOLVE_LAUNCH
#undef SE3MPC_SOLVE_CASE
  return launch_status("se3mpc_solve(second tier)");
}

}  // namespace se3mpc

using namespace se3mpc;

extern "C" int se3mpc_set_solver_variant(int variant) {
  const int forced = (variant >> 8) & 0xFF;
  if (variant < 0 || (variant & ~0xFF01) != 0 || !(forced == 0 || forced == 8 || forced == 16 || forced == 32 || forced == 64)) return SE3MPC_ERR_SHAPE;
  se3mpc::g_solver_variant = variant;
  return SE3MPC_OK;
}

extern "C" int se3mpc_solve_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                                const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att,
                                float* rates, float* thrust, void* stream) {
  return solve_impl<float>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream);
}
extern "C" int se3mpc_solve_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                                const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att,
                                double* rates, double* thrust, void* stream) {
  return solve_impl<double>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream);
}

// Host-latency form of the solve for SE3MPCPlanner.plan_trajectory (planner.py:215-228: one problem, the caller blocks until the plan
// exists): ONE call launches the solve on the caller's host-pinned, device-mapped buffers and returns when the result is there.
// Instead of hipStreamSynchronize (an interrupt / signal wait of 10-20 us on top of the kernel) the kernel's last act is a
// system-scope store of `ticket` into *done, on which this function spins.
template <typename IO>
static int plan_host_impl(const se3mpc_params* p, int B, const IO* p0, const IO* v0, const IO* goal, const IO* x0, IO* X,
                          se3mpc_solve_info* info, IO* acc, IO* att, IO* rates, IO* thrust, unsigned long long* done,
                          unsigned long long ticket, double timeout_us, void* stream) {
  if (done == nullptr) return SE3MPC_ERR_NULL;
  volatile unsigned long long* flag = done;
  if (*flag == ticket) return SE3MPC_ERR_SHAPE;            // the ticket must differ from what the flag holds
  const int rc = solve_impl<IO>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream, done, ticket);
  if (rc != SE3MPC_OK) return rc;
  if (B == 0) return SE3MPC_OK;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 1; *flag != ticket; ++spins) {
    if ((spins & 255u) == 0) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3 > timeout_us) {
        // (a device that does not make the store visible while the host polls, or a very long solve: the ordinary wait)
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return launch_status("se3mpc_plan_host(synchronize)");
        return *flag == ticket ? SE3MPC_OK : SE3MPC_ERR_LAUNCH;
      }
    }
  }
  return SE3MPC_OK;
}

extern "C" int se3mpc_plan_host_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                                    const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att, float* rates,
                                    float* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us,
                                    void* stream) {
  return plan_host_impl<float>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket, timeout_us, stream);
}
extern "C" int se3mpc_plan_host_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                                    const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att, double* rates,
                                    double* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us,
                                    void* stream) {
  return plan_host_impl<double>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket, timeout_us, stream);
}
