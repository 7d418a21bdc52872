// closed_loop.hip -- the consumer side of the Planner->Controller contract on the device, one drone per lane:
// plan sample -> geometric controller -> simulator step, repeated `nsteps` times in ONE launch.
//
// Reference arithmetic (unit-stripped, reproduced with its quirks; "controller.py" =
// src/dart_planner/control/geometric_controller.py, "onboard.py" = src/dart_planner/control/onboard_controller.py,
// "simulator.py" = src/dart_planner/utils/drone_simulator.py):
//   * plan sample   OnboardController._interpolate_trajectory                onboard.py:43-93
//   * control law   GeometricController.compute_control + everything it calls  controller.py:413-512, :160-257, :548-658,
//                   :660-715, :813-828; compute_body_rate_command               controller.py:706-726
//   * simulator     DroneSimulator.step                                       simulator.py:52-72
// The glue compute_control_from_trajectory(state, trajectory, t) is a stub in the reference (controller.py:873-875);
// here it is the composition of the two reference functions above (sampler -> compute_control, yaw = yaw rate = 0).
//
// Every drone is an independent, strictly sequential recurrence with ~50 bytes of state, so the kernel is bound by
// the dependent-instruction latency of one lane (sin/cos/acos/sqrt chains), not by HBM; throughput comes from the number
// of drones in flight.  The plan is read in place from the solver's outputs (problem layout: P and V blocks of X with
// stride 9N, accelerations with stride 3N), the state arrays are the [B][3] arrays the next solve reads as p0 / v0, so a
// receding-horizon Monte-Carlo alternates se3mpc_solve_* and se3mpc_closed_loop_* with no glue kernels in between.
//
// Contraction is off in this file: the controller's saturation / singularity / failsafe branches compare against values
// NumPy computes without FMA; keeping products and sums separately rounded keeps the f64 path on the oracle's side of
// every branch (transcendental functions still differ by an ulp).
#pragma clang fp contract(off)
#include "closed_loop_device.hpp"

namespace se3mpc {

// ---- one control call per drone (compute_control / compute_body_rate_command for B drones)
template <typename R>
__global__ void __launch_bounds__(64)
control_kernel(CtrlDev<R> c, int B, const double* __restrict__ time, const R* __restrict__ pos, const R* __restrict__ vel,
               const R* __restrict__ att, const R* __restrict__ omega, const R* __restrict__ dpos, const R* __restrict__ dvel,
               const R* __restrict__ dacc, const R* __restrict__ yaw, const R* __restrict__ yaw_rate, double* __restrict__ state,
               R* __restrict__ thrust, R* __restrict__ torque, R* __restrict__ body_thrust, R* __restrict__ body_rates,
               int32_t* __restrict__ flags, const double* __restrict__ sample_time, int N, const double* __restrict__ timestamps,
               long long ts_stride, const R* __restrict__ P, long long strideP, const R* __restrict__ V, long long strideV,
               const R* __restrict__ A, long long strideA, R* __restrict__ target) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  CtrlRegs<R> s = load_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS);
  R p[3], v[3], a[3], w[3], dp[3], dv[3], da[3];
  for (int i = 0; i < 3; ++i) { p[i] = pos[3 * b + i]; v[i] = vel[3 * b + i]; a[i] = att[3 * b + i]; w[i] = omega[3 * b + i]; }
  if (sample_time != nullptr) {                       // compute_control_from_trajectory: the target is the plan sampled at sample_time
    PlanCursor<R> cur;
    cursor_reset(cur);
    sample_plan<R>(sample_time[b], N, timestamps + (size_t)b * ts_stride, P + (size_t)b * strideP,
                   V != nullptr ? V + (size_t)b * strideV : nullptr, A != nullptr ? A + (size_t)b * strideA : nullptr, dp, dv, da, cur);
    if (target != nullptr) for (int i = 0; i < 3; ++i) { target[9 * b + i] = dp[i]; target[9 * b + 3 + i] = dv[i]; target[9 * b + 6 + i] = da[i]; }
  } else {
    for (int i = 0; i < 3; ++i) { dp[i] = dpos[3 * b + i]; dv[i] = dvel[3 * b + i]; da[i] = dacc != nullptr ? dacc[3 * b + i] : (R)0; }
  }
  R th, tq[3];
  int fl;
  control_step<R>(c, s, time[b], p, v, a, w, dp, dv, da, yaw != nullptr ? yaw[b] : (R)0, yaw_rate != nullptr ? yaw_rate[b] : (R)0, th, tq, fl);
  store_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS, s);
  if (thrust != nullptr) thrust[b] = th;
  if (torque != nullptr) for (int i = 0; i < 3; ++i) torque[3 * b + i] = tq[i];
  if (body_thrust != nullptr) body_thrust[b] = fmin(fmax(th / c.max_thrust, (R)0), (R)1);   // controller.py:721
  if (body_rates != nullptr) {
    const R inr[3] = {(R)0.1, (R)0.1, (R)0.2};                                    // controller.py:717 (its own inertia, not the config's)
    for (int i = 0; i < 3; ++i) body_rates[3 * b + i] = w[i] + (tq[i] / inr[i]) * (R)0.001;   // :718-720
  }
  if (flags != nullptr) flags[b] = fl;
}

// ---- one compute_control_fast call per drone
template <typename R>
__global__ void __launch_bounds__(64)
control_fast_kernel(CtrlDev<R> c, FastDev<R> f, int B, double dt, const R* __restrict__ pos, const R* __restrict__ vel,
                    const R* __restrict__ att, const R* __restrict__ omega, const R* __restrict__ dpos, const R* __restrict__ dvel,
                    const R* __restrict__ dacc, const R* __restrict__ yaw, const R* __restrict__ yaw_rate, double* __restrict__ state,
                    R* __restrict__ thrust, R* __restrict__ torque, int32_t* __restrict__ flags) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  CtrlRegs<R> s = load_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS);
  R p[3], v[3], a[3], w[3], dp[3], dv[3], da[3];
  for (int i = 0; i < 3; ++i) {
    p[i] = pos[3 * b + i]; v[i] = vel[3 * b + i]; a[i] = att[3 * b + i]; w[i] = omega[3 * b + i];
    dp[i] = dpos[3 * b + i]; dv[i] = dvel[3 * b + i]; da[i] = dacc != nullptr ? dacc[3 * b + i] : (R)0;
  }
  R th, tq[3];
  int fl;
  control_step_fast<R>(c, f, s, dt, p, v, a, w, dp, dv, da, yaw != nullptr ? yaw[b] : (R)0, yaw_rate != nullptr ? yaw_rate[b] : (R)0, th, tq, fl);
  store_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS, s);
  if (thrust != nullptr) thrust[b] = th;
  if (torque != nullptr) for (int i = 0; i < 3; ++i) torque[3 * b + i] = tq[i];
  if (flags != nullptr) flags[b] = fl;
}

// ---- the controller's building blocks, one call per drone: what the reference's own controller tests call directly
// (tests/control/test_geometric_controller_anti_windup.py, test_geometric_controller_yaw_singularity.py, tests/test_controller_torque_calculation.py)
template <typename R>
struct Inertia9 {
  R m[9];
  int use;
};

// _update_integral_error(vel_error, dt, thrust_saturated, torque_saturated) (controller.py:536-564): the saturation flags are ARGUMENTS
// (sat bit 0 thrust, bits 1..3 torque x/y/z; null = none), the unsaturated thrust / torques are read from the record.
template <typename R>
__global__ void __launch_bounds__(64)
integral_update_kernel(CtrlDev<R> c, int B, const R* __restrict__ vel_error, double dt, const int32_t* __restrict__ sat, double* __restrict__ state) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  CtrlRegs<R> s = load_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS);
  const int keep = s.flags, sb = sat != nullptr ? sat[b] : 0;
  s.flags = (s.flags & ~(4 | 8 | 16)) | (((sb >> 1) & 7) << 2);
  const R ve[3] = {vel_error[3 * b], vel_error[3 * b + 1], vel_error[3 * b + 2]};
  update_integral(c, s, ve, (R)dt, (sb & 1) != 0);
  s.flags = keep;
  store_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS, s);
}

// _geometric_attitude_control / _fast_geometric_attitude_control(att, ang_vel, b3_des, yaw_des, yaw_rate_des) (:643-704, :348-411)
template <typename R>
__global__ void __launch_bounds__(64)
attitude_torque_kernel(CtrlDev<R> c, Inertia9<R> I, int B, const R* __restrict__ att, const R* __restrict__ omega, const R* __restrict__ b3_des,
                       const R* __restrict__ yaw, const R* __restrict__ yaw_rate, double* __restrict__ state, R* __restrict__ torque,
                       int32_t* __restrict__ flags) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  CtrlRegs<R> s = load_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS);
  R a[3], w[3], b3[3], tq[3];
  for (int i = 0; i < 3; ++i) { a[i] = att[3 * b + i]; w[i] = omega[3 * b + i]; b3[i] = b3_des[3 * b + i]; }
  int fl = 0;
  attitude_torque<R>(c, s, (R)ldexp(1.0, -s.halvings), b3, a, w, yaw != nullptr ? yaw[b] : (R)0, yaw_rate != nullptr ? yaw_rate[b] : (R)0,
                     I.use ? I.m : (const R*)nullptr, tq, fl);
  store_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS, s);
  if (torque != nullptr) for (int i = 0; i < 3; ++i) torque[3 * b + i] = tq[i];
  if (flags != nullptr) flags[b] = fl;
}

// _detect_yaw_singularity(yaw_vector, b3_des) (:160-189) and the frame: method < 0 -> what _geometric_attitude_control builds (the configured
// fallback only when singular); method >= 0 -> _handle_yaw_singularity(yaw_vector, b3_des, current_yaw, method) (:191-252) whatever cos_angle says.
template <typename R>
__global__ void __launch_bounds__(64)
desired_frame_kernel(CtrlDev<R> c, int B, int method, const R* __restrict__ yaw_vector, const R* __restrict__ b3_des, const R* __restrict__ current_yaw,
                     R* __restrict__ frame, R* __restrict__ cos_angle, int32_t* __restrict__ singular) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  R yv[3], b3[3], b1[3], b2[3], ca;
  for (int i = 0; i < 3; ++i) { yv[i] = yaw_vector[3 * b + i]; b3[i] = b3_des[3 * b + i]; }
  const R cyaw = current_yaw != nullptr ? current_yaw[b] : (R)0;
  bool sing;
  desired_frame<R>(c, method < 0 ? c.fallback : method, method >= 0, yv, b3, cos(cyaw), sin(cyaw), b1, b2, ca, sing);
  if (frame != nullptr) for (int i = 0; i < 3; ++i) { frame[9 * b + i] = b1[i]; frame[9 * b + 3 + i] = b2[i]; frame[9 * b + 6 + i] = b3[i]; }
  if (cos_angle != nullptr) cos_angle[b] = ca;
  if (singular != nullptr) singular[b] = sing ? 1 : 0;
}

// ---- the closed loop: nsteps x (sample, control, simulate) per drone in one launch
template <typename R>
__global__ void __launch_bounds__(64)
closed_loop_kernel(CtrlDev<R> c, SimDev<R> m, int B, int nsteps, double sim_dt, int N, const double* __restrict__ timestamps,
                   long long ts_stride, const R* __restrict__ P, long long strideP, const R* __restrict__ V, long long strideV,
                   const R* __restrict__ A, long long strideA, double* __restrict__ time, R* __restrict__ pos, R* __restrict__ vel,
                   R* __restrict__ att, R* __restrict__ omega, double* __restrict__ state, const R* __restrict__ wind,
                   long long wind_stride, int gust_step, R gx, R gy, R gz, int stop_at_plan_end, R* __restrict__ log_state,
                   R* __restrict__ log_cmd, double* __restrict__ log_time, int32_t* __restrict__ steps_taken) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  CtrlRegs<R> s = load_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS);
  R p[3], v[3], a[3], w[3], wd[3] = {(R)0, (R)0, (R)0};
  for (int i = 0; i < 3; ++i) {
    p[i] = pos[3 * b + i]; v[i] = vel[3 * b + i]; a[i] = att[3 * b + i]; w[i] = omega[3 * b + i];
    if (wind != nullptr) wd[i] = wind[(size_t)b * wind_stride + i];
  }
  double t = time[b];
  const double* ts = timestamps + (size_t)b * ts_stride;
  const R* Pb = P + (size_t)b * strideP;
  const R* Vb = V != nullptr ? V + (size_t)b * strideV : nullptr;
  const R* Ab = A != nullptr ? A + (size_t)b * strideA : nullptr;
  const double ts_last = ts[N - 1];
  const R dt = (R)sim_dt;
  int taken = 0;
  bool active = true;
  PlanCursor<R> cur;
  cursor_reset(cur);
  for (int step = 0; step < nsteps; ++step) {
    if (stop_at_plan_end && t > ts_last) active = false;                          // contract tests :130-131 / :263-264 (`break`)
    if (log_state != nullptr) {
      R* ls = log_state + ((size_t)step * B + b) * 12;
      for (int i = 0; i < 3; ++i) { ls[i] = p[i]; ls[3 + i] = v[i]; ls[6 + i] = a[i]; ls[9 + i] = w[i]; }
    }
    if (log_time != nullptr) log_time[(size_t)step * B + b] = t;
    R th = (R)NAN, tq[3] = {(R)NAN, (R)NAN, (R)NAN};
    if (active) {
      R tp[3], tv[3], ta[3];
      if (!(sim_dt > 0.0)) cur.idx = 0;                                        // a clock that does not advance: search from the start
      sample_plan<R>(t, N, ts, Pb, Vb, Ab, tp, tv, ta, cur);
      int fl;
      control_step<R>(c, s, t, p, v, a, w, tp, tv, ta, (R)0, (R)0, th, tq, fl);
      if (step == gust_step) { wd[0] = gx; wd[1] = gy; wd[2] = gz; }              // the gust of contract test :293-296
      simulator_step<R>(m, p, v, a, w, t, th, tq, dt, sim_dt, wd);
      ++taken;
    }
    if (log_cmd != nullptr) {
      R* lc = log_cmd + ((size_t)step * B + b) * 4;
      lc[0] = th; lc[1] = tq[0]; lc[2] = tq[1]; lc[3] = tq[2];
    }
  }
  for (int i = 0; i < 3; ++i) { pos[3 * b + i] = p[i]; vel[3 * b + i] = v[i]; att[3 * b + i] = a[i]; omega[3 * b + i] = w[i]; }
  time[b] = t;
  store_ctrl<R>(state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS, s);
  if (steps_taken != nullptr) steps_taken[b] = taken;
}

// ---- DroneSimulator.step alone (simulator.py:52-72) for B drones with given commands
template <typename R>
__global__ void __launch_bounds__(64)
simulator_step_kernel(SimDev<R> m, int B, double dt_d, const R* __restrict__ thrust, const R* __restrict__ torque, const R* __restrict__ wind,
                      long long wind_stride, double* __restrict__ time, R* __restrict__ pos, R* __restrict__ vel, R* __restrict__ att,
                      R* __restrict__ omega) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  R p[3], v[3], a[3], w[3], tq[3], wd[3];
  for (int i = 0; i < 3; ++i) {
    p[i] = pos[3 * b + i]; v[i] = vel[3 * b + i]; a[i] = att[3 * b + i]; w[i] = omega[3 * b + i]; tq[i] = torque[3 * b + i];
    wd[i] = wind != nullptr ? wind[(size_t)b * wind_stride + i] : (R)0;
  }
  double t = time[b];
  simulator_step<R>(m, p, v, a, w, t, thrust[b], tq, (R)dt_d, dt_d, wd);
  for (int i = 0; i < 3; ++i) { pos[3 * b + i] = p[i]; vel[3 * b + i] = v[i]; att[3 * b + i] = a[i]; omega[3 * b + i] = w[i]; }
  time[b] = t;
}

__global__ void __launch_bounds__(64)
controller_reset_kernel(int B, double hover, double* __restrict__ state) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double* s = state + (size_t)b * SE3MPC_CONTROLLER_STATE_WORDS;
  for (int i = 0; i < SE3MPC_CONTROLLER_STATE_WORDS; ++i) s[i] = 0.0;
  s[3] = __builtin_nan("");                                                      // last_time = None (controller.py:92)
  s[4] = hover;                                                                   // last_valid_thrust = mass * gravity (:105)
}

template <typename R>
int control_impl(const se3mpc_controller_params* cp, int B, const double* time, const R* pos, const R* vel, const R* att, const R* omega,
                 const R* dpos, const R* dvel, const R* dacc, const R* yaw, const R* yaw_rate, double* state, R* thrust, R* torque,
                 R* body_thrust, R* body_rates, int32_t* flags, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!time || !pos || !vel || !att || !omega || !dpos || !dvel || !state) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(control_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), B, time, pos, vel,
                     att, omega, dpos, dvel, dacc, yaw, yaw_rate, state, thrust, torque, body_thrust, body_rates, flags,
                     (const double*)nullptr, 0, (const double*)nullptr, 0LL, (const R*)nullptr, 0LL, (const R*)nullptr, 0LL, (const R*)nullptr, 0LL,
                     (R*)nullptr);
  return launch_status("se3mpc_control");
}

template <typename R>
int control_fast_impl(const se3mpc_controller_params* cp, double vehicle_mass, double vehicle_gravity, int B, double dt, const R* pos,
                      const R* vel, const R* att, const R* omega, const R* dpos, const R* dvel, const R* dacc, const R* yaw,
                      const R* yaw_rate, double* state, R* thrust, R* torque, int32_t* flags, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (!std::isfinite(vehicle_mass) || !std::isfinite(vehicle_gravity) || !(vehicle_mass > 0.0) || dt != dt) return SE3MPC_ERR_PARAM;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!pos || !vel || !att || !omega || !dpos || !dvel || !state) return SE3MPC_ERR_NULL;
  FastDev<R> f;
  f.gravity = (R)vehicle_gravity;
  f.min_thrust_abs = (R)(cp->min_thrust * vehicle_mass * vehicle_gravity);       // controller.py:127
  f.hover = (R)(vehicle_mass * vehicle_gravity);                                  // :280
  hipLaunchKernelGGL(control_fast_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), f, B, dt, pos, vel,
                     att, omega, dpos, dvel, dacc, yaw, yaw_rate, state, thrust, torque, flags);
  return launch_status("se3mpc_control_fast");
}

template <typename R>
int integral_update_impl(const se3mpc_controller_params* cp, int B, const R* vel_error, double dt, const int32_t* saturation, double* state,
                         void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (dt != dt) return SE3MPC_ERR_PARAM;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!vel_error || !state) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(integral_update_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), B, vel_error, dt,
                     saturation, state);
  return launch_status("se3mpc_controller_integral_update");
}

template <typename R>
int attitude_torque_impl(const se3mpc_controller_params* cp, int B, const R* att, const R* omega, const R* b3_des, const R* yaw,
                         const R* yaw_rate, const double* inertia, double* state, R* torque, int32_t* flags, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  Inertia9<R> I;
  I.use = inertia != nullptr;
  for (int i = 0; i < 9; ++i) {
    if (inertia != nullptr && !std::isfinite(inertia[i])) return SE3MPC_ERR_PARAM;
    I.m[i] = inertia != nullptr ? (R)inertia[i] : (R)0;
  }
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!att || !omega || !b3_des || !state) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(attitude_torque_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), I, B, att, omega,
                     b3_des, yaw, yaw_rate, state, torque, flags);
  return launch_status("se3mpc_controller_attitude_torque");
}

template <typename R>
int desired_frame_impl(const se3mpc_controller_params* cp, int B, int method, const R* yaw_vector, const R* b3_des, const R* current_yaw,
                       R* frame, R* cos_angle, int32_t* singular, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (B < 0 || method > 3) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!yaw_vector || !b3_des) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(desired_frame_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), B, method, yaw_vector,
                     b3_des, current_yaw, frame, cos_angle, singular);
  return launch_status("se3mpc_controller_desired_frame");
}

template <typename R>
int control_plan_impl(const se3mpc_controller_params* cp, int B, const double* time, const double* sample_time, const R* pos, const R* vel,
                      const R* att, const R* omega, int N, const double* timestamps, long long ts_stride, const R* P, long long strideP,
                      const R* V, long long strideV, const R* A, long long strideA, double* state, R* thrust, R* torque, R* body_thrust,
                      R* body_rates, int32_t* flags, R* target, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (B < 0 || N < 1 || N > 4096 || ts_stride < 0 || strideP < 0 || strideV < 0 || strideA < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!time || !sample_time || !pos || !vel || !att || !omega || !timestamps || !P || !state) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(control_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), B, time, pos, vel,
                     att, omega, (const R*)nullptr, (const R*)nullptr, (const R*)nullptr, (const R*)nullptr, (const R*)nullptr, state, thrust,
                     torque, body_thrust, body_rates, flags, sample_time, N, timestamps, ts_stride, P, strideP, V, strideV, A, strideA, target);
  return launch_status("se3mpc_control_plan");
}

template <typename R>
int simulator_step_impl(const se3mpc_simulator_params* sp, int B, double dt, const R* thrust, const R* torque, const R* wind,
                        long long wind_stride, double* time, R* pos, R* vel, R* att, R* omega, void* stream) {
  int rc = check_simulator_params(sp);
  if (rc) return rc;
  if (!std::isfinite(dt)) return SE3MPC_ERR_PARAM;
  if (B < 0 || wind_stride < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!thrust || !torque || !time || !pos || !vel || !att || !omega) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(simulator_step_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_sim_dev<R>(*sp), B, dt, thrust,
                     torque, wind, wind_stride, time, pos, vel, att, omega);
  return launch_status("se3mpc_simulator_step");
}

template <typename R>
int closed_loop_impl(const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int nsteps, double sim_dt, int N,
                     const double* timestamps, long long ts_stride, const R* P, long long strideP, const R* V, long long strideV,
                     const R* A, long long strideA, double* time, R* pos, R* vel, R* att, R* omega, double* state, const R* wind,
                     long long wind_stride, int gust_step, const double* gust_wind, int stop_at_plan_end, R* log_state, R* log_cmd,
                     double* log_time, int32_t* steps_taken, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  rc = check_simulator_params(sp);
  if (rc) return rc;
  if (!std::isfinite(sim_dt)) return SE3MPC_ERR_PARAM;
  if (B < 0 || nsteps < 0 || N < 1 || N > 4096 || ts_stride < 0 || strideP < 0 || strideV < 0 || strideA < 0 || wind_stride < 0)
    return SE3MPC_ERR_SHAPE;
  if (B == 0 || nsteps == 0) return SE3MPC_OK;
  if (!timestamps || !P || !time || !pos || !vel || !att || !omega || !state || (gust_step >= 0 && !gust_wind)) return SE3MPC_ERR_NULL;
  const R gx = gust_step >= 0 ? (R)gust_wind[0] : (R)0, gy = gust_step >= 0 ? (R)gust_wind[1] : (R)0, gz = gust_step >= 0 ? (R)gust_wind[2] : (R)0;
  hipLaunchKernelGGL(closed_loop_kernel<R>, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, make_ctrl_dev<R>(*cp), make_sim_dev<R>(*sp),
                     B, nsteps, sim_dt, N, timestamps, ts_stride, P, strideP, V, strideV, A, strideA, time, pos, vel, att, omega, state, wind,
                     wind_stride, gust_step, gx, gy, gz, stop_at_plan_end, log_state, log_cmd, log_time, steps_taken);
  return launch_status("se3mpc_closed_loop");
}

}  // namespace se3mpc

using namespace se3mpc;

extern "C" int se3mpc_controller_default_params(se3mpc_controller_params* out) {
  if (out == nullptr) return SE3MPC_ERR_NULL;
  // GeometricControllerConfig (controller.py:26-77) after _apply_tuning_profile("sitl_optimized") (:140-158, control_config.py:95-111);
  // mass / gravity / inertia: VehicleParams (common/vehicle_params.py:19-23); max_torque_xyz: compute_max_torque_xyz's safe default (:108-115)
  const se3mpc_controller_params d = {{20.0, 20.0, 25.0}, {1.5, 1.5, 2.0}, {10.0, 10.0, 12.0}, {18.0, 18.0, 8.0}, {7.0, 7.0, 3.5},
                                      {0.02, 0.02, 0.04}, {0.5, 0.5, 0.05}, {2.0, 2.0, 3.0},
                                      2.5, M_PI / 4.0, 1.0, 9.80665, 22.0, 0.8, 1.0, 0.6, 0.1, 0.99, 0.95, 0.1, 0.0, 0, 0};
  *out = d;
  return SE3MPC_OK;
}

extern "C" int se3mpc_simulator_default_params(se3mpc_simulator_params* out) {
  if (out == nullptr) return SE3MPC_ERR_NULL;
  const se3mpc_simulator_params d = {1.5, 9.81, {0.1, 0.1, 0.2}, 20.0, 10.0};     // simulator.py:41-50
  *out = d;
  return SE3MPC_OK;
}

extern "C" int se3mpc_controller_reset(const se3mpc_controller_params* cp, int B, double* state, void* stream) {
  int rc = check_controller_params(cp);
  if (rc) return rc;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!state) return SE3MPC_ERR_NULL;
  hipLaunchKernelGGL(controller_reset_kernel, dim3(grid_for(B, 64)), dim3(64), 0, (hipStream_t)stream, B, cp->mass * cp->gravity, state);
  return launch_status("se3mpc_controller_reset");
}

#define SE3MPC_DEFINE_LOOP_API(SUF, R)                                                                                      \
  extern "C" int se3mpc_control_##SUF(const se3mpc_controller_params* cp, int B, const double* time, const R* pos, const R* vel, \
                                      const R* att, const R* omega, const R* dpos, const R* dvel, const R* dacc, const R* yaw, \
                                      const R* yaw_rate, double* state, R* thrust, R* torque, R* body_thrust, R* body_rates,  \
                                      int32_t* flags, void* stream) {                                                       \
    return control_impl<R>(cp, B, time, pos, vel, att, omega, dpos, dvel, dacc, yaw, yaw_rate, state, thrust, torque,       \
                           body_thrust, body_rates, flags, stream);                                                        \
  }                                                                                                                         \
  extern "C" int se3mpc_control_fast_##SUF(const se3mpc_controller_params* cp, double vehicle_mass, double vehicle_gravity, int B,  \
                                           double dt, const R* pos, const R* vel, const R* att, const R* omega, const R* dpos,   \
                                           const R* dvel, const R* dacc, const R* yaw, const R* yaw_rate, double* state, R* thrust, \
                                           R* torque, int32_t* flags, void* stream) {                                          \
    return control_fast_impl<R>(cp, vehicle_mass, vehicle_gravity, B, dt, pos, vel, att, omega, dpos, dvel, dacc, yaw, yaw_rate, \
                                state, thrust, torque, flags, stream);                                                      \
  }                                                                                                                         \
  extern "C" int se3mpc_controller_integral_update_##SUF(const se3mpc_controller_params* cp, int B, const R* vel_error, double dt,   \
                                                         const int32_t* saturation, double* state, void* stream) {             \
    return integral_update_impl<R>(cp, B, vel_error, dt, saturation, state, stream);                                        \
  }                                                                                                                         \
  extern "C" int se3mpc_controller_attitude_torque_##SUF(const se3mpc_controller_params* cp, int B, const R* att, const R* omega,   \
                                                         const R* b3_des, const R* yaw, const R* yaw_rate, const double* inertia, \
                                                         double* state, R* torque, int32_t* flags, void* stream) {           \
    return attitude_torque_impl<R>(cp, B, att, omega, b3_des, yaw, yaw_rate, inertia, state, torque, flags, stream);          \
  }                                                                                                                         \
  extern "C" int se3mpc_controller_desired_frame_##SUF(const se3mpc_controller_params* cp, int B, int method, const R* yaw_vector,  \
                                                       const R* b3_des, const R* current_yaw, R* frame, R* cos_angle,        \
                                                       int32_t* singular, void* stream) {                                   \
    return desired_frame_impl<R>(cp, B, method, yaw_vector, b3_des, current_yaw, frame, cos_angle, singular, stream);         \
  }                                                                                                                         \
  extern "C" int se3mpc_control_plan_##SUF(const se3mpc_controller_params* cp, int B, const double* time, const double* sample_time, \
                                           const R* pos, const R* vel, const R* att, const R* omega, int N, const double* timestamps, \
                                           long long ts_stride, const R* P, long long strideP, const R* V, long long strideV,   \
                                           const R* A, long long strideA, double* state, R* thrust, R* torque, R* body_thrust,  \
                                           R* body_rates, int32_t* flags, R* target, void* stream) {                           \
    return control_plan_impl<R>(cp, B, time, sample_time, pos, vel, att, omega, N, timestamps, ts_stride, P, strideP, V, strideV, A, \
                                strideA, state, thrust, torque, body_thrust, body_rates, flags, target, stream);               \
  }                                                                                                                         \
  extern "C" int se3mpc_simulator_step_##SUF(const se3mpc_simulator_params* sp, int B, double dt, const R* thrust, const R* torque, \
                                             const R* wind, long long wind_stride, double* time, R* pos, R* vel, R* att, R* omega, \
                                             void* stream) {                                                                \
    return simulator_step_impl<R>(sp, B, dt, thrust, torque, wind, wind_stride, time, pos, vel, att, omega, stream);         \
  }                                                                                                                         \
  extern "C" int se3mpc_closed_loop_##SUF(const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B,      \
                                          int nsteps, double sim_dt, int N, const double* timestamps, long long ts_stride,   \
                                          const R* P, long long strideP, const R* V, long long strideV, const R* A,          \
                                          long long strideA, double* time, R* pos, R* vel, R* att, R* omega, double* state,  \
                                          const R* wind, long long wind_stride, int gust_step, const double* gust_wind,      \
                                          int stop_at_plan_end, R* log_state, R* log_cmd, double* log_time,                  \
                                          int32_t* steps_taken, void* stream) {                                             \
    return closed_loop_impl<R>(cp, sp, B, nsteps, sim_dt, N, timestamps, ts_stride, P, strideP, V, strideV, A, strideA, time, \
                               pos, vel, att, omega, state, wind, wind_stride, gust_step, gust_wind, stop_at_plan_end,       \
                               log_state, log_cmd, log_time, steps_taken, stream);                                          \
  }

SE3MPC_DEFINE_LOOP_API(f32, float)
SE3MPC_DEFINE_LOOP_API(f64, double)
