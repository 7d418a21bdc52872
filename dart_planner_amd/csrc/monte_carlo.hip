// monte_carlo.hip -- the receding-horizon closed-loop Monte-Carlo of BASELINE config 5's named test shape
// (/root/reference/tests/test_monte_carlo_sim.py:24-72: `cycles` planning cycles of `substeps` control + simulator steps per drone) in
// ONE launch: every planning cycle of every drone runs inside the same kernel -- solve (solve_body.inc, the batched solver's own
// code: same bits as se3mpc_solve_*), plan handed to the controller through LDS, `substeps` x (plan sample -> geometric controller ->
// simulator step) (closed_loop_device.hpp, the closed-loop kernel's own code: same bits as se3mpc_closed_loop_*), next cycle.
//
// Why: driven as 2 x cycles launches (se3mpc_solve_* + se3mpc_closed_loop_* per cycle, control/closed_loop.py) every cycle waits at two
// kernel boundaries for its SLOWEST drone.  In the loop's own statistics 99.4 % of the solves stop after one L-BFGS-B iteration and a
// handful per cycle take three -- three times as long -- so each solve launch lasted ~30 us for ~11 us of typical work, 33 times in a
// row.  Drones are independent: here a wavefront's drones run their cycles back to back and only pay for their own slow solves.
//
// Mapping: the solver's (solve_device.hpp): G lanes per drone, lane k = horizon step k, 64 / G drones per wavefront.  The controller /
// simulator recurrence of a drone is strictly sequential scalar code: it runs on the group's first lane (the others wait at the
// group barrier), its 50 bytes of state parked in LDS while the solver has the registers.
#include "solve_device.hpp"
#pragma clang fp contract(off)
#include "closed_loop_device.hpp"
#pragma clang fp contract(fast)

namespace se3mpc {

// LDS of one wavefront behind the solver's image: per drone (group) the plan [3][G][3] in the IO type (positions, velocities,
// accelerations as the solver stores them), its G stamps, and the drone's state: pos, vel, att, omega, wind (15 IO), time (1 double),
// controller record (SE3MPC_CONTROLLER_STATE_WORDS doubles)
template <typename IO>
__host__ __device__ constexpr size_t mc_group_bytes(int G) {
  return ((size_t)(9 * G + 16) * sizeof(IO) + (size_t)(G + 1 + SE3MPC_CONTROLLER_STATE_WORDS) * sizeof(double) + 15) / 16 * 16 +
         (sizeof(CtrlDev<IO>) + sizeof(SimDev<IO>) + 15) / 16 * 16;
}

// One wavefront per SIMD (512 registers): the kernel is a chain of dependent scalar recurrences -- a second resident wavefront would
// only matter from 8192 x (G / 8) drones up -- and at the solver's 256-register budget the loop-invariant constants the compiler hoists
// out of the cycle loop cost this kernel ~90 spilled registers.
#ifndef SE3MPC_MC_WAVES
#define SE3MPC_MC_WAVES 1
#endif
template <typename IO, int G>
__global__ void __launch_bounds__(64, SE3MPC_MC_WAVES)
monte_carlo_kernel(SolveDev q, CtrlDev<IO> ctl, SimDev<IO> sim, int B, int cycles, int substeps, double sim_dt, size_t solver_lds,
                   const IO* __restrict__ goalg, const IO* __restrict__ windg, long long wind_stride, double* __restrict__ timeg,
                   IO* __restrict__ posg, IO* __restrict__ velg, IO* __restrict__ attg, IO* __restrict__ omegag,
                   double* __restrict__ stateg, IO* __restrict__ Xg, IO* __restrict__ accg, se3mpc_solve_info* __restrict__ infog,
                   int* __restrict__ overflowed) {
  HIP_DYNAMIC_SHARED(unsigned char, lds_raw)
  constexpr int P = kWave / G, J = kSlots;
  const int lane = lane_id();
  const int k = lane & (G - 1);
  const int grp = lane / G;
  const int pb = blockIdx.x * P + grp;     // drone index
  if (pb >= B) return;                     // (a whole group leaves together)
  // ---- this drone's LDS block (pointers are re-derived inside each phase: only `grp` stays live while the solver has the registers)
#define SE3MPC_MC_BLOCK()                                                                                                              \
  unsigned char* gb = lds_raw + solver_lds + (size_t)grp * mc_group_bytes<IO>(G);                                                       \
  double* stamps = reinterpret_cast<double*>(gb);                                  /* [G] */                                            \
  double* s_time = stamps + G;                                                     /* [1] */                                            \
  double* s_ctrl = s_time + 1;                                                     /* [SE3MPC_CONTROLLER_STATE_WORDS] */                \
  IO* planP = reinterpret_cast<IO*>(s_ctrl + SE3MPC_CONTROLLER_STATE_WORDS);       /* [G][3] */                                         \
  IO* planV = planP + 3 * G;                                                                                                            \
  IO* planA = planV + 3 * G;                                                                                                            \
  IO* s_vec = planA + 3 * G;                                                       /* pos, vel, att, omega, wind: [15] */               \
  /* the controller's and the simulator's constants: parked in LDS too, so that the ~50 scalar registers they would occupy as kernel */ \
  /* arguments are free while the solver runs (they are read back inside the act phase only) */                                        \
  CtrlDev<IO>* l_ctl = reinterpret_cast<CtrlDev<IO>*>(gb + ((size_t)(9 * G + 16) * sizeof(IO) + (size_t)(G + 1 + SE3MPC_CONTROLLER_STATE_WORDS) * sizeof(double) + 15) / 16 * 16); \
  SimDev<IO>* l_sim = reinterpret_cast<SimDev<IO>*>(l_ctl + 1);                                                                         \
  (void)stamps; (void)s_time; (void)s_ctrl; (void)planP; (void)planV; (void)planA; (void)s_vec; (void)l_ctl; (void)l_sim;
  double goal[3] = {0.0, 0.0, 0.0};
  if (q.has_goal) {
#pragma unroll
    for (int a = 0; a < 3; ++a) goal[a] = (double)goalg[pb * 3 + a];
  }
  if (k == 0) {
    SE3MPC_MC_BLOCK()
    for (int i = 0; i < 3; ++i) {
      s_vec[i] = posg[3 * pb + i]; s_vec[3 + i] = velg[3 * pb + i]; s_vec[6 + i] = attg[3 * pb + i]; s_vec[9 + i] = omegag[3 * pb + i];
      s_vec[12 + i] = windg != nullptr ? windg[(size_t)pb * wind_stride + i] : (IO)0;
    }
    *s_time = timeg[pb];
    *l_ctl = ctl; *l_sim = sim;
    for (int i = 0; i < SE3MPC_CONTROLLER_STATE_WORDS; ++i) s_ctrl[i] = stateg[(size_t)pb * SE3MPC_CONTROLLER_STATE_WORDS + i];
  }
  group_sync<G>();
  const IO* x0row = nullptr;               // every cycle re-plans from the reference's cold start
  const bool cold = true;
  for (int cycle = 0; cycle < cycles; ++cycle) {
    // ---- plan: the batched solver's body on (pos, vel) as the solve kernel would read them from its [B][3] arrays
    double ps[3], vs[3];
    {
      SE3MPC_MC_BLOCK()
#pragma unroll
      for (int a = 0; a < 3; ++a) { ps[a] = (double)s_vec[a]; vs[a] = (double)s_vec[3 + a]; }
    }
    double x[J];
    {
#include "solve_body.inc"
      if (task == SE3MPC_TASK_OVERFLOW && k == 0) atomicAdd(overflowed, 1);      // the LDS image was too small for this solve: the caller falls back
      SE3MPC_MC_BLOCK()
      // the plan as se3mpc_solve_* stores it (rounded to the IO type) and as se3mpc_closed_loop_* reads it
      if (live) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { planP[3 * k + a] = (IO)x[a]; planV[3 * k + a] = (IO)x[3 + a]; }
        planA[3 * k + 0] = (IO)(x[6] / q.mass); planA[3 * k + 1] = (IO)(x[7] / q.mass); planA[3 * k + 2] = (IO)(x[8] / q.mass - q.grav);
        stamps[k] = plan_stamp(cycle, substeps, sim_dt, k, q.dt);
      }
      if (cycle == cycles - 1 && live) {
        if (Xg != nullptr) {
#pragma unroll
          for (int j = 0; j < J; ++j) Xg[(size_t)pb * n + (j / 3) * n3 + 3 * k + (j % 3)] = (IO)x[j];
        }
        if (accg != nullptr) {
          const size_t o = (size_t)pb * n3 + 3 * k;
          accg[o] = planA[3 * k + 0]; accg[o + 1] = planA[3 * k + 1]; accg[o + 2] = planA[3 * k + 2];
        }
        if (infog != nullptr && k == 0) {
          se3mpc_solve_info r;
          r.fun = f; r.nit = nit; r.nfev = nfev; r.status = status; r.task = task;
          infog[pb] = r;
        }
      }
    }
    group_sync<G>();
    // ---- act: `substeps` x (sample the plan, geometric controller, simulator step) on the group's first lane
    if (k == 0) {
      SE3MPC_MC_BLOCK()
      const CtrlDev<IO> c = *l_ctl;
      const SimDev<IO> m = *l_sim;
      CtrlRegs<IO> s = load_ctrl<IO>(s_ctrl);
      IO p[3], v[3], a[3], w[3], wd[3];
      for (int i = 0; i < 3; ++i) { p[i] = s_vec[i]; v[i] = s_vec[3 + i]; a[i] = s_vec[6 + i]; w[i] = s_vec[9 + i]; wd[i] = s_vec[12 + i]; }
      double t = *s_time;
      const IO dt = (IO)sim_dt;
      PlanCursor<IO> cur;
      cursor_reset(cur);
      for (int step = 0; step < substeps; ++step) {
        IO tp[3], tv[3], ta[3];
        if (!(sim_dt > 0.0)) cur.idx = 0;
        sample_plan<IO>(t, q.N, stamps, planP, planV, planA, tp, tv, ta, cur);
        IO th, tq[3];
        int fl;
        control_step<IO>(c, s, t, p, v, a, w, tp, tv, ta, (IO)0, (IO)0, th, tq, fl);
        simulator_step<IO>(m, p, v, a, w, t, th, tq, dt, sim_dt, wd);
      }
      for (int i = 0; i < 3; ++i) { s_vec[i] = p[i]; s_vec[3 + i] = v[i]; s_vec[6 + i] = a[i]; s_vec[9 + i] = w[i]; }
      *s_time = t;
      store_ctrl<IO>(s_ctrl, s);
    }
    group_sync<G>();
  }
  if (k == 0) {
    SE3MPC_MC_BLOCK()
    for (int i = 0; i < 3; ++i) { posg[3 * pb + i] = s_vec[i]; velg[3 * pb + i] = s_vec[3 + i]; attg[3 * pb + i] = s_vec[6 + i]; omegag[3 * pb + i] = s_vec[9 + i]; }
    timeg[pb] = *s_time;
    for (int i = 0; i < SE3MPC_CONTROLLER_STATE_WORDS; ++i) stateg[(size_t)pb * SE3MPC_CONTROLLER_STATE_WORDS + i] = s_ctrl[i];
  }
#undef SE3MPC_MC_BLOCK
}

template <typename IO>
int monte_carlo_impl(const se3mpc_params* p, const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int cycles,
                     int substeps, double sim_dt, const IO* goal, const IO* wind, long long wind_stride, double* time, IO* pos, IO* vel,
                     IO* att, IO* omega, double* state, IO* X_last, IO* acc_last, se3mpc_solve_info* info_last, int32_t* overflowed,
                     void* stream) {
  if (p == nullptr || cp == nullptr || sp == nullptr) return SE3MPC_ERR_NULL;
  int rc = check_params_impl(p);
  if (rc) return rc;
  rc = check_controller_params(cp);
  if (rc) return rc;
  rc = check_simulator_params(sp);
  if (rc) return rc;
  if (B < 0 || cycles < 0 || substeps < 0 || p->horizon > kWave) return SE3MPC_ERR_SHAPE;
  if (!std::isfinite(sim_dt)) return SE3MPC_ERR_PARAM;
  if (wind != nullptr && !(wind_stride == 0 || wind_stride >= 3)) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!time || !pos || !vel || !att || !omega || !state || !overflowed || (p->has_goal && !goal)) return SE3MPC_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(overflowed, 0, sizeof(int32_t), s) != hipSuccess) return launch_status("se3mpc_monte_carlo(memset)");
  SolveDev q = make_solve_dev(*p);
  const int G = p->horizon <= 8 ? 8 : (p->horizon <= 16 ? 16 : (p->horizon <= 32 ? 32 : 64));
  const int waves = (int)(((long)B * G + kWave - 1) / kWave);
  // pairs of L-BFGS memory the LDS image holds: as many (<= maxcor) as still let every wavefront of the launch be resident at once,
  // at least 4 -- a solve with the reference's options stores at most two; a solve that needs more raises `overflowed`
  const size_t extra = (size_t)(kWave / G) * mc_group_bytes<IO>(G);
  const size_t per_cu = (size_t)(waves + 255) / 256;                  // wavefronts a CU must hold for the whole launch to be resident
  const size_t budget = (size_t)160 * 1024 / (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  int mlds = q.m;
  while (mlds > 4 && solve_lds_bytes(mlds, G, sizeof(IO)) + extra > budget) --mlds;
  q.mlds = mlds;
  const size_t solver_lds = (solve_lds_bytes(mlds, G, sizeof(IO)) + 15) / 16 * 16;
  const size_t lds = solver_lds + extra;
  const CtrlDev<IO> c = make_ctrl_dev<IO>(*cp);
  const SimDev<IO> m = make_sim_dev<IO>(*sp);
#define SE3MPC_MC_CASE(GG)                                                                                                       \
  {                                                                                                                              \
    if (lds > 64 * 1024)                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&monte_carlo_kernel<IO, GG>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                                       \
    hipLaunchKernelGGL((monte_carlo_kernel<IO, GG>), dim3(waves), dim3(kWave), lds, s, q, c, m, B, cycles, substeps, sim_dt, solver_lds, \
                       goal, wind, wind_stride, time, pos, vel, att, omega, state, X_last, acc_last, info_last, overflowed);    \
  }
  if (G == 8) SE3MPC_MC_CASE(8)
  else if (G == 16) SE3MPC_MC_CASE(16)
  else if (G == 32) SE3MPC_MC_CASE(32)
  else SE3MPC_MC_CASE(64)
#undef SE3MPC_MC_CASE
  return launch_status("se3mpc_monte_carlo");
}

}  // namespace se3mpc

using namespace se3mpc;

extern "C" int se3mpc_monte_carlo_f32(const se3mpc_params* p, const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B,
                                      int cycles, int substeps, double sim_dt, const float* goal, const float* wind, long long wind_stride,
                                      double* time, float* pos, float* vel, float* att, float* omega, double* state, float* X_last,
                                      float* acc_last, se3mpc_solve_info* info_last, int32_t* overflowed, void* stream) {
  return monte_carlo_impl<float>(p, cp, sp, B, cycles, substeps, sim_dt, goal, wind, wind_stride, time, pos, vel, att, omega, state, X_last,
                                 acc_last, info_last, overflowed, stream);
}
extern "C" int se3mpc_monte_carlo_f64(const se3mpc_params* p, const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B,
                                      int cycles, int substeps, double sim_dt, const double* goal, const double* wind, long long wind_stride,
                                      double* time, double* pos, double* vel, double* att, double* omega, double* state, double* X_last,
                                      double* acc_last, se3mpc_solve_info* info_last, int32_t* overflowed, void* stream) {
  return monte_carlo_impl<double>(p, cp, sp, B, cycles, substeps, sim_dt, goal, wind, wind_stride, time, pos, vel, att, omega, state, X_last,
                                  acc_last, info_last, overflowed, stream);
}
