// closed_loop_device.hpp -- device code of the consumer side of the contract (plan sample -> geometric controller -> simulator step, one
// drone per lane), shared by closed_loop.hip (its kernels and C entry points) and monte_carlo.hip (the fused receding-horizon loop).
// INCLUDE UNDER `#pragma clang fp contract(off)`: the controller's saturation / singularity / failsafe branches compare against values
// NumPy computes without FMA (see closed_loop.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "se3mpc_common.hpp"

static_assert(sizeof(se3mpc_controller_params) == 304, "se3mpc_controller_params is part of the C ABI (capi.py mirrors it)");
static_assert(sizeof(se3mpc_simulator_params) == 56, "se3mpc_simulator_params is part of the C ABI");

namespace se3mpc {

template <typename R>
struct CtrlDev {
  R kp_pos[3], ki_pos[3], kd_pos[3], kp_att[3], kd_att[3], inertia[3], max_torque[3], max_int_axis[3];
  R max_integral_pos, max_tilt, cos_max_tilt, gravity, max_thrust, min_thrust_abs, hover, track_thr, vel_thr, kb, decay, sat_thr,
      yaw_sing_thr, dh_cos, dh_sin;
  int anti_windup, fallback;
};

template <typename R>
static CtrlDev<R> make_ctrl_dev(const se3mpc_controller_params& p) {
  CtrlDev<R> c;
  for (int i = 0; i < 3; ++i) {
    c.kp_pos[i] = (R)p.kp_pos[i]; c.ki_pos[i] = (R)p.ki_pos[i]; c.kd_pos[i] = (R)p.kd_pos[i];
    c.kp_att[i] = (R)p.kp_att[i]; c.kd_att[i] = (R)p.kd_att[i]; c.inertia[i] = (R)p.inertia[i];
    c.max_torque[i] = (R)p.max_torque_xyz[i]; c.max_int_axis[i] = (R)p.max_integral_per_axis[i];
  }
  c.max_integral_pos = (R)p.max_integral_pos; c.max_tilt = (R)p.max_tilt_angle; c.cos_max_tilt = (R)std::cos(p.max_tilt_angle);
  c.gravity = (R)p.gravity; c.max_thrust = (R)p.max_thrust;
  c.min_thrust_abs = (R)(p.min_thrust * p.mass * p.gravity);              // controller.py:469
  c.hover = (R)(p.mass * p.gravity);                                       // controller.py:105
  c.track_thr = (R)p.tracking_error_threshold; c.vel_thr = (R)p.velocity_error_threshold;
  c.kb = (R)p.back_calculation_gain; c.decay = (R)p.integral_decay_factor; c.sat_thr = (R)p.saturation_threshold;
  c.yaw_sing_thr = (R)p.yaw_singularity_threshold;
  c.dh_cos = (R)std::cos(p.default_heading_yaw); c.dh_sin = (R)std::sin(p.default_heading_yaw);
  c.anti_windup = p.anti_windup_method; c.fallback = p.yaw_fallback_method;
  return c;
}

template <typename R>
struct SimDev {
  R mass, gravity, inertia[3], max_thrust, max_torque;
};
template <typename R>
static SimDev<R> make_sim_dev(const se3mpc_simulator_params& p) {
  SimDev<R> s;
  s.mass = (R)p.mass; s.gravity = (R)p.gravity; s.max_thrust = (R)p.max_thrust; s.max_torque = (R)p.max_torque;
  for (int i = 0; i < 3; ++i) s.inertia[i] = (R)p.inertia[i];
  return s;
}

// The mutable members of GeometricController that compute_control reads or writes (controller.py:87-105), in registers.
// In memory: double[SE3MPC_CONTROLLER_STATE_WORDS] per drone (include/se3mpc.h).
template <typename R>
struct CtrlRegs {
  R integral[3];
  double last_time;          // NaN = None
  R last_valid_thrust, unsat_thrust, unsat_torque[3];
  int failsafe_count, halvings, flags;   // flags: 1 failsafe_active, 2 last_thrust_saturated, 4/8/16 last_torque_saturated x/y/z
};

template <typename R>
__device__ __forceinline__ CtrlRegs<R> load_ctrl(const double* __restrict__ s) {
  CtrlRegs<R> r;
  for (int i = 0; i < 3; ++i) { r.integral[i] = (R)s[i]; r.unsat_torque[i] = (R)s[6 + i]; }
  r.last_time = s[3]; r.last_valid_thrust = (R)s[4]; r.unsat_thrust = (R)s[5];
  r.failsafe_count = (int)s[9]; r.halvings = (int)s[10]; r.flags = (int)s[11];
  return r;
}
template <typename R>
__device__ __forceinline__ void store_ctrl(double* __restrict__ s, const CtrlRegs<R>& r) {
  for (int i = 0; i < 3; ++i) { s[i] = (double)r.integral[i]; s[6 + i] = (double)r.unsat_torque[i]; }
  s[3] = r.last_time; s[4] = (double)r.last_valid_thrust; s[5] = (double)r.unsat_thrust;
  s[9] = (double)r.failsafe_count; s[10] = (double)r.halvings; s[11] = (double)r.flags;
}

template <typename R>
__device__ __forceinline__ R norm3(const R v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
// sin and cos of one angle with ONE argument reduction (the same two values the separate calls return)
__device__ __forceinline__ void sin_cos(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void sin_cos(double x, double& s, double& c) { sincos(x, &s, &c); }
template <typename R>
__device__ __forceinline__ R sign_of(R v) { return v > (R)0 ? (R)1 : (v < (R)0 ? (R)-1 : v); }   // np.sign (NaN stays NaN)
template <typename R>
__device__ __forceinline__ void cross3(const R a[3], const R b[3], R o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// _get_failsafe_command (controller.py:813-828)
template <typename R>
__device__ __forceinline__ void enter_failsafe(CtrlRegs<R>& s) {
  if (!(s.flags & 1)) {
    s.halvings += 1;                                                     // :817-821 gains * 0.5
    s.integral[0] = s.integral[1] = s.integral[2] = (R)0;                // :823
    s.failsafe_count += 1;                                               // :825
  }
  s.flags |= 1;                                                          // :827
}

// Output flags of one control call.
enum { CF_FAILSAFE = 1, CF_BAD_DT = 2, CF_THRUST_SAT = 4, CF_SINGULAR = 8, CF_TILT = 16, CF_TORQUE_SAT_X = 32 };

// _update_integral_error (controller.py:548-578) with its anti-windup method and _clamp_integral_per_axis; the torque saturation flags
// and unsaturated torques in `s` are still the previous call's (the current torque is computed afterwards, :480 vs :494).
template <typename R>
__device__ __forceinline__ void update_integral(const CtrlDev<R>& c, CtrlRegs<R>& s, const R ve[3], R dt, bool thrust_sat) {
  R upd[3];
  for (int i = 0; i < 3; ++i) upd[i] = ve[i] * dt;
  if (c.anti_windup == 0) {                                                    // clamping (:580-598)
    if (thrust_sat) for (int i = 0; i < 3; ++i) upd[i] = upd[i] * (R)0.1;
    for (int i = 0; i < 3; ++i) if (s.flags & (4 << i)) upd[i] = upd[i] * (R)0.1;
  } else if (c.anti_windup == 1) {                                             // back-calculation (:600-623)
    if (thrust_sat) {
      const R fb = (s.unsat_thrust - c.max_thrust) * c.kb;
      upd[0] = upd[0] - fb * (R)0.33; upd[1] = upd[1] - fb * (R)0.33; upd[2] = upd[2] - fb * (R)0.34;
    }
    for (int i = 0; i < 3; ++i)
      if (s.flags & (4 << i)) upd[i] = upd[i] - ((s.unsat_torque[i] - c.max_torque[i]) * c.kb) * (R)0.5;
  }
  R I[3];
  for (int i = 0; i < 3; ++i) {
    I[i] = s.integral[i] + upd[i];                                             // :574
    if (fabs(I[i]) > c.max_int_axis[i]) I[i] = sign_of(I[i]) * c.max_int_axis[i];   // :630-632
  }
  const R mag = norm3(I);
  if (mag > c.max_integral_pos) { const R f = c.max_integral_pos / mag; for (int i = 0; i < 3; ++i) I[i] = I[i] * f; }   // :635-637
  for (int i = 0; i < 3; ++i) {
    if (fabs(I[i]) > c.max_int_axis[i] * c.sat_thr) I[i] = I[i] * c.decay;     // :640-643
    s.integral[i] = I[i];
  }
}

// _detect_yaw_singularity (controller.py:160-189) and the desired frame of _geometric_attitude_control: the normal construction
// (:680-689) or, when |yaw_vector . b3| >= the threshold (or `force`: a direct _handle_yaw_singularity call), the fallback `method`
// (:191-257; 0 skip_yaw, 1 default_heading, 2 maintain_current, 3 any other string).  b3n as given (the callers normalise it, :667);
// (cy, sy) = cos / sin of the CURRENT yaw.
template <typename R>
__device__ __forceinline__ void desired_frame(const CtrlDev<R>& c, int method, bool force, const R yv[3], const R b3n[3], R cy, R sy, R b1[3],
                                              R b2[3], R& cos_angle, bool& singular) {
  cos_angle = fabs((yv[0] * b3n[0] + yv[1] * b3n[1]) + yv[2] * b3n[2]);            // :174
  singular = cos_angle >= c.yaw_sing_thr;                                         // :177
  if (singular || force) {                                                        // :191-257
    R proj[3] = {(R)1 - b3n[0] * b3n[0], (R)0 - b3n[0] * b3n[1], (R)0 - b3n[0] * b3n[2]};   // [1,0,0] - ([1,0,0].b3) b3
    const R np_ = norm3(proj);
    proj[0] = proj[0] / np_; proj[1] = proj[1] / np_; proj[2] = proj[2] / np_;
    if (method == 1 || method == 2) {                                             // default_heading / maintain_current
      const R hv[3] = {method == 1 ? c.dh_cos : cy, method == 1 ? c.dh_sin : sy, (R)0};
      R cx[3];
      cross3(hv, b3n, cx);
      const R nc = norm3(cx);
      if (nc > (R)1e-6) { b1[0] = cx[0] / nc; b1[1] = cx[1] / nc; b1[2] = cx[2] / nc; }
      else { b1[0] = proj[0]; b1[1] = proj[1]; b1[2] = proj[2]; }
    } else if (method != 0 || fabs(b3n[2]) < (R)0.99) {                           // skip_yaw (:212-217); unknown methods (:248-252)
      b1[0] = proj[0]; b1[1] = proj[1]; b1[2] = proj[2];
    } else {                                                                      // :218-220
      b1[0] = (R)1; b1[1] = (R)0; b1[2] = (R)0;
    }
  } else {                                                                        // :680-688
    cross3(yv, b3n, b1);
    const R n1 = norm3(b1);
    if (n1 > (R)1e-6) { b1[0] = b1[0] / n1; b1[1] = b1[1] / n1; b1[2] = b1[2] / n1; }
    else { b1[0] = (R)1; b1[1] = (R)0; b1[2] = (R)0; }
  }
  cross3(b3n, b1, b2);                                                            // :255 / :689
}

// _geometric_attitude_control (controller.py:660-715) == _fast_geometric_attitude_control (:348-411): torque command, unsaturated
// torques and the torque saturation flags of `s`.  b3 = desired thrust direction (normalised here, :667), scale = 0.5 ** failsafe
// halvings, Ifull = a full 3x3 inertia (row-major; the reference's tests assign one to _fast_inertia) or nullptr = diag(c.inertia).
template <typename R>
__device__ void attitude_torque(const CtrlDev<R>& c, CtrlRegs<R>& s, R scale, const R b3[3], const R att[3], const R omega[3], R yaw_des,
                                R yaw_rate_des, const R* Ifull, R torque_out[3], int& out_flags) {
  R cr, sr, cp, sp, cy, sy;
  sin_cos(att[0], sr, cr); sin_cos(att[1], sp, cp); sin_cos(att[2], sy, cy);
  const R Rm[3][3] = {{cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr},   // :774-789
                      {sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr},
                      {-sp, cp * sr, cp * cr}};
  R yc, ys;
  sin_cos(yaw_des, ys, yc);
  const R yv[3] = {yc, ys, (R)0};                                                 // :665
  const R n3 = norm3(b3);
  const R b3n[3] = {b3[0] / n3, b3[1] / n3, b3[2] / n3};                          // :667
  R b1[3], b2[3], cos_angle;
  bool singular;
  desired_frame(c, c.fallback, false, yv, b3n, cy, sy, b1, b2, cos_angle, singular);
  if (singular) out_flags |= CF_SINGULAR;
  const R Rd[3][3] = {{b1[0], b2[0], b3n[0]}, {b1[1], b2[1], b3n[1]}, {b1[2], b2[2], b3n[2]}};   // column_stack
  // eR = 0.5 vee(Rd^T R - R^T Rd) (:692): vee(M) = (M21, M02, M10)
  auto dtr = [&](const R A_[3][3], const R B_[3][3], int i, int j) { return (A_[0][i] * B_[0][j] + A_[1][i] * B_[1][j]) + A_[2][i] * B_[2][j]; };
  const R eR[3] = {(R)0.5 * (dtr(Rd, Rm, 2, 1) - dtr(Rm, Rd, 2, 1)), (R)0.5 * (dtr(Rd, Rm, 0, 2) - dtr(Rm, Rd, 0, 2)),
                   (R)0.5 * (dtr(Rd, Rm, 1, 0) - dtr(Rm, Rd, 1, 0))};
  const R eO[3] = {omega[0], omega[1], omega[2] - yaw_rate_des};                  // :693-695
  R Iw[3];
  if (Ifull != nullptr) for (int i = 0; i < 3; ++i) Iw[i] = (Ifull[3 * i] * omega[0] + Ifull[3 * i + 1] * omega[1]) + Ifull[3 * i + 2] * omega[2];
  else for (int i = 0; i < 3; ++i) Iw[i] = c.inertia[i] * omega[i];
  R cor[3];
  cross3(omega, Iw, cor);                                                         // :700
  int tsat = 0;
  for (int i = 0; i < 3; ++i) {
    R tq = (-(c.kp_att[i] * scale) * eR[i] - (c.kd_att[i] * scale) * eO[i]) + cor[i];   // :701
    s.unsat_torque[i] = tq;                                                       // :704
    if (fabs(tq) > c.max_torque[i]) { tq = sign_of(tq) * c.max_torque[i]; tsat |= (4 << i); }   // :708-711
    torque_out[i] = tq;
  }
  s.flags = (s.flags & ~(4 | 8 | 16)) | tsat;                                     // :713
  out_flags |= (tsat >> 2) * CF_TORQUE_SAT_X;
}

// The desired thrust direction with the tilt limit (controller.py:487-496 == :328-339 of the fast path), then the attitude law.
// tvw = thrust vector (world), tm = its SATURATED magnitude.
template <typename R>
__device__ void attitude_command(const CtrlDev<R>& c, CtrlRegs<R>& s, R scale, const R tvw[3], R tm, const R att[3], const R omega[3],
                                 R yaw_des, R yaw_rate_des, R torque_out[3], int& out_flags) {
  R b3[3];
  if (tm > (R)1e-6) { b3[0] = tvw[0] / tm; b3[1] = tvw[1] / tm; b3[2] = tvw[2] / tm; }
  else { b3[0] = (R)0; b3[1] = (R)0; b3[2] = (R)1; }
  // tilt = arccos(b3z) > max_tilt (controller.py:489-490) decided on the cosines: arccos is strictly decreasing on [-1, 1], and at the boundary the
  // rescaling below is the identity (sf = 1), so the one rounding-width band where the two forms could disagree changes nothing continuous
  if (fmin(fmax(b3[2], (R)-1), (R)1) < c.cos_max_tilt) {
    const R sf = c.cos_max_tilt / b3[2];
    b3[0] = b3[0] * sf; b3[1] = b3[1] * sf; b3[2] = c.cos_max_tilt;
    const R n = norm3(b3);
    b3[0] = b3[0] / n; b3[1] = b3[1] / n; b3[2] = b3[2] / n;
    out_flags |= CF_TILT;
  }
  attitude_torque(c, s, scale, b3, att, omega, yaw_des, yaw_rate_des, (const R*)nullptr, torque_out, out_flags);
}

// GeometricController.compute_control (controller.py:413-512) for one drone.
template <typename R>
__device__ void control_step(const CtrlDev<R>& c, CtrlRegs<R>& s, double t, const R pos[3], const R vel[3], const R att[3],
                             const R omega[3], const R dpos[3], const R dvel[3], const R dacc[3], R yaw_des, R yaw_rate_des,
                             R& thrust_out, R torque_out[3], int& out_flags) {
  out_flags = 0;
  const double dt_d = (s.last_time != s.last_time) ? 0.001 : t - s.last_time;   // :440
  s.last_time = t;                                                               // :441
  if (dt_d <= 0.0 || dt_d > 0.1) {                                               // :442-443
    enter_failsafe(s);
    thrust_out = s.last_valid_thrust; torque_out[0] = torque_out[1] = torque_out[2] = (R)0;
    out_flags = CF_FAILSAFE | CF_BAD_DT;
    return;
  }
  const R dt = (R)dt_d;
  const R scale = (R)ldexp(1.0, -s.halvings);                                    // 0.5 ** halvings, exact
  R pe[3], ve[3], tvw[3];
  for (int i = 0; i < 3; ++i) { pe[i] = dpos[i] - pos[i]; ve[i] = dvel[i] - vel[i]; }   // :445-446
  const R pen = norm3(pe), ven = norm3(ve);
  for (int i = 0; i < 3; ++i) {
    const R acc_pid = ((c.kp_pos[i] * scale) * pe[i] + (c.kd_pos[i] * scale) * ve[i]) + c.ki_pos[i] * s.integral[i];   // :453-457
    tvw[i] = dacc[i] + acc_pid;                                                  // :458
  }
  tvw[2] = tvw[2] + c.gravity;                                                   // :461 acc_des - (0, 0, -g)
  R tm = norm3(tvw);                                                             // :462
  s.unsat_thrust = tm;                                                           // :465
  bool thrust_sat = false;
  if (tm > c.max_thrust) { tm = c.max_thrust; thrust_sat = true; }               // :470-475
  else if (tm < c.min_thrust_abs) { tm = c.min_thrust_abs; thrust_sat = true; }
  s.flags = (s.flags & ~2) | (thrust_sat ? 2 : 0);                               // :477
  if (thrust_sat) out_flags |= CF_THRUST_SAT;
  update_integral(c, s, ve, dt, thrust_sat);                                     // :480
  // ---- _check_tracking_performance (:650-658)
  if (pen > c.track_thr && ven > c.vel_thr) s.failsafe_count += 1;
  else s.failsafe_count = s.failsafe_count > 1 ? s.failsafe_count - 1 : 0;
  if (s.failsafe_count > 100) {                                                  // :485-486
    enter_failsafe(s);
    thrust_out = s.last_valid_thrust; torque_out[0] = torque_out[1] = torque_out[2] = (R)0;
    out_flags |= CF_FAILSAFE;
    return;
  }
  attitude_command(c, s, scale, tvw, tm, att, omega, yaw_des, yaw_rate_des, torque_out, out_flags);   // :487-504
  s.last_valid_thrust = tm;                                                       // :505
  s.flags &= ~1;                                                                  // :506
  s.failsafe_count = 0;                                                           // :507
  thrust_out = tm;
}

// GeometricController.compute_control_fast (controller.py:253-346; compute_control_from_fast_state, :728-768, forwards to it) for one
// drone: the unit-free path of the reference's 400 Hz hardware loop (hardware/pixhawk_interface.py:401).  dt is an argument; an invalid
// one returns the VEHICLE's hover thrust and touches nothing (:279-280); gravity and the lower thrust limit come from the vehicle
// constants (common/vehicle_params.py:19-23 via :118-127), not the controller config; no tracking check, and last_time /
// last_valid_thrust / failsafe_active / failsafe_count stay as they are (gains halved by a compute_control failsafe stay halved).
template <typename R>
struct FastDev {
  R gravity, min_thrust_abs, hover;
};
template <typename R>
__device__ void control_step_fast(const CtrlDev<R>& c, const FastDev<R>& f, CtrlRegs<R>& s, double dt_d, const R pos[3], const R vel[3],
                                  const R att[3], const R omega[3], const R dpos[3], const R dvel[3], const R dacc[3], R yaw_des,
                                  R yaw_rate_des, R& thrust_out, R torque_out[3], int& out_flags) {
  out_flags = 0;
  if (dt_d <= 0.0 || dt_d > 0.1) {                                               // :279-280
    thrust_out = f.hover; torque_out[0] = torque_out[1] = torque_out[2] = (R)0;
    out_flags = CF_BAD_DT;
    return;
  }
  const R dt = (R)dt_d;
  const R scale = (R)ldexp(1.0, -s.halvings);
  R pe[3], ve[3], tvw[3];
  for (int i = 0; i < 3; ++i) { pe[i] = dpos[i] - pos[i]; ve[i] = dvel[i] - vel[i]; }   // :283-284
  for (int i = 0; i < 3; ++i) {
    const R acc_pid = ((c.kp_pos[i] * scale) * pe[i] + (c.kd_pos[i] * scale) * ve[i]) + c.ki_pos[i] * s.integral[i];   // :293-297
    tvw[i] = dacc[i] + acc_pid;                                                  // :298
  }
  tvw[2] = tvw[2] + f.gravity;                                                   // :301
  R tm = norm3(tvw);                                                             // :302
  s.unsat_thrust = tm;                                                           // :305
  bool thrust_sat = false;
  if (tm > c.max_thrust) { tm = c.max_thrust; thrust_sat = true; }               // :312-320 (the caller counts the saturations from the flags)
  else if (tm < f.min_thrust_abs) { tm = f.min_thrust_abs; thrust_sat = true; }
  s.flags = (s.flags & ~2) | (thrust_sat ? 2 : 0);                               // :322
  if (thrust_sat) out_flags |= CF_THRUST_SAT;
  update_integral(c, s, ve, dt, thrust_sat);                                     // :325
  attitude_command(c, s, scale, tvw, tm, att, omega, yaw_des, yaw_rate_des, torque_out, out_flags);   // :328-344
  thrust_out = tm;
}

// OnboardController._interpolate_trajectory (onboard.py:43-93).  ts: N timestamps (double); P, V, A: [N][3] rows (V, A may be null).
// PlanCursor carries what one drone's clock lets the next sample reuse: the search index (np.searchsorted(ts, t) for an earlier t of the
// same sorted plan -- the scan resumes there instead of paying up to N dependent loads per step) and the two plan rows that bracket it
// (a loop whose simulator step is longer than the plan's, or that has run past the plan's end, samples the same rows again and again).
template <typename R>
struct PlanCursor {
  int idx;          // first i with ts[i] >= t of the last sample (0 before the first)
  int rows_of;      // idx the cached rows belong to (-1: none)
  double t1, t2;
  R r1[9], r2[9];   // (P, V, A) of rows idx-1 and idx (clamped)
};
template <typename R>
__device__ __forceinline__ void cursor_reset(PlanCursor<R>& c) { c.idx = 0; c.rows_of = -1; }

template <typename R>
__device__ __forceinline__ void sample_plan(double t, int N, const double* __restrict__ ts, const R* __restrict__ P, const R* __restrict__ V,
                                            const R* __restrict__ A, R tp[3], R tv[3], R ta[3], PlanCursor<R>& c) {
  int idx = c.idx;                                                                // np.searchsorted(ts, t): first i with ts[i] >= t
  while (idx < N && ts[idx] < t) ++idx;
  c.idx = idx;
  if (idx != c.rows_of) {
    const int i1 = idx == 0 ? 0 : (idx >= N ? N - 1 : idx - 1), i2 = idx >= N ? N - 1 : idx;
    c.t1 = ts[i1]; c.t2 = ts[i2];
    for (int a = 0; a < 3; ++a) {
      c.r1[a] = P[3 * i1 + a]; c.r2[a] = P[3 * i2 + a];
      c.r1[3 + a] = V != nullptr ? V[3 * i1 + a] : (R)0; c.r2[3 + a] = V != nullptr ? V[3 * i2 + a] : (R)0;
      c.r1[6 + a] = A != nullptr ? A[3 * i1 + a] : (R)0; c.r2[6 + a] = A != nullptr ? A[3 * i2 + a] : (R)0;
    }
    c.rows_of = idx;
  }
  if (idx == 0 || idx >= N) {                                                     // :51-75: the first / last row as it stands
    for (int a = 0; a < 3; ++a) { tp[a] = c.r1[a]; tv[a] = c.r1[3 + a]; ta[a] = c.r1[6 + a]; }
    return;
  }
  const R f = (R)((t - c.t1) / (c.t2 - c.t1));                                    // :80
  for (int a = 0; a < 3; ++a) {
    tp[a] = c.r1[a] + f * (c.r2[a] - c.r1[a]);                                    // :81
    tv[a] = c.r1[3 + a] + f * (c.r2[3 + a] - c.r1[3 + a]);                        // :83-86
    ta[a] = c.r1[6 + a] + f * (c.r2[6 + a] - c.r1[6 + a]);                        // :88-91
  }
}

// DroneSimulator.step (simulator.py:52-72)
template <typename R>
__device__ __forceinline__ void simulator_step(const SimDev<R>& m, R pos[3], R vel[3], R att[3], R omega[3], double& t, R thrust,
                                               const R torque[3], R dt, double dt_d, const R wind[3]) {
  thrust = fmin(fmax(thrust, (R)0), m.max_thrust);                                // :54
  for (int i = 0; i < 3; ++i) {
    const R tq = fmin(fmax(torque[i], -m.max_torque), m.max_torque);              // :55
    const R wind_accel = wind[i] / m.mass;                                        // :57
    const R acc = ((i == 2 ? -m.gravity : (R)0) + (i == 2 ? thrust / m.mass : (R)0)) + wind_accel;   // :59
    vel[i] = vel[i] + acc * dt;                                                   // :60
    pos[i] = pos[i] + vel[i] * dt;                                                // :61
    const R ang_acc = tq / m.inertia[i];                                          // :63
    omega[i] = omega[i] + ang_acc * dt;                                           // :64
    att[i] = att[i] + omega[i] * dt;                                              // :65
  }
  t = t + dt_d;                                                                   // :67
}

static inline int check_controller_params(const se3mpc_controller_params* p) {
  if (p == nullptr) return SE3MPC_ERR_NULL;
  const double* d = reinterpret_cast<const double*>(p);
  for (int i = 0; i < 37; ++i)
    if (!std::isfinite(d[i])) return SE3MPC_ERR_PARAM;
  if (!(p->mass > 0.0) || p->anti_windup_method < 0 || p->anti_windup_method > 2 || p->yaw_fallback_method < 0 || p->yaw_fallback_method > 3)
    return SE3MPC_ERR_PARAM;
  return SE3MPC_OK;
}

static inline int check_simulator_params(const se3mpc_simulator_params* sp) {
  if (sp == nullptr) return SE3MPC_ERR_NULL;
  if (!(sp->mass > 0.0) || !std::isfinite(sp->gravity) || !(sp->inertia[0] > 0.0) || !(sp->inertia[1] > 0.0) || !(sp->inertia[2] > 0.0) ||
      !std::isfinite(sp->max_thrust) || !std::isfinite(sp->max_torque))
    return SE3MPC_ERR_PARAM;
  return SE3MPC_OK;
}

// the stamp of plan row k of planning cycle `cycle` as the host computes it (control/closed_loop.py: (cycle * substeps * sim_dt) + k * dt,
// two rounded products and a rounded sum -- no fused multiply-add)
__device__ __forceinline__ double plan_stamp(int cycle, int substeps, double sim_dt, int k, double dt) {
  const double base = (double)(cycle * substeps) * sim_dt;
  const double off = (double)k * dt;
  return base + off;
}

}  // namespace se3mpc
