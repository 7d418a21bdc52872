// se3mpc_common.hpp -- shared host/device plumbing of libse3mpc (gfx950 / CDNA4 only).
//
// Data model: a decision vector has 9N rows in the reference packing
// [P | V | T] (reference planner.py:361-376).  "Lane layout" stores a batch as [row][b] with
// b fastest, so one 64-lane wavefront reading one row touches one contiguous 256-B (f32) or
// 512-B (f64) segment of HBM; "problem layout" stores [b][row] and gives one wavefront a whole
// problem (the solver).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "se3mpc.h"

namespace se3mpc {

constexpr int kWave = 64;  // CDNA wavefront width

// Device-side constants derived once on the host from se3mpc_params (all in the kernel's
// arithmetic type so f32 kernels never touch f64 registers).
template <typename R>
struct DevParams {
  int N;
  int has_goal;
  R dt, half_dt2, inv_dt, mass, inv_mass, grav, hover;
  R wp, wv, wa, wT, term;
  R pos_b, v_max, txy, tz_lo, tz_hi;   // box of planner.py:378-402
  R v_max2, a_max2, t_max2, t_min2;    // planner.py:472-497
  R margin;
};

template <typename R>
inline DevParams<R> make_dev_params(const se3mpc_params& p) {
  DevParams<R> d;
  d.N = p.horizon;
  d.has_goal = p.has_goal;
  d.dt = (R)p.dt;
  d.half_dt2 = (R)(0.5 * p.dt * p.dt);
  d.inv_dt = (R)(1.0 / p.dt);
  d.mass = (R)p.mass;
  d.inv_mass = (R)(1.0 / p.mass);
  d.grav = (R)p.gravity;
  d.hover = (R)(p.mass * p.gravity);
  d.wp = (R)p.position_weight;
  d.wv = (R)p.velocity_weight;
  d.wa = (R)p.acceleration_weight;
  d.wT = (R)p.thrust_weight;
  d.term = (R)p.terminal_factor;
  d.pos_b = (R)p.position_bound;
  d.v_max = (R)p.max_velocity;
  d.txy = (R)(p.max_thrust * sin(p.max_tilt_angle));
  d.tz_lo = (R)p.min_thrust;
  d.tz_hi = (R)p.max_thrust;
  d.v_max2 = (R)(p.max_velocity * p.max_velocity);
  d.a_max2 = (R)(p.max_acceleration * p.max_acceleration);
  d.t_max2 = (R)(p.max_thrust * p.max_thrust);
  d.t_min2 = (R)(p.min_thrust * p.min_thrust);
  d.margin = (R)p.safety_margin;
  return d;
}

// Box bounds of decision-vector row `row` (planner.py:378-402).
template <typename R>
__device__ __forceinline__ void row_bounds(const DevParams<R>& q, int row, R& lo, R& hi) {
  const int N3 = 3 * q.N;
  if (row < N3) {
    lo = -q.pos_b; hi = q.pos_b;
  } else if (row < 2 * N3) {
    lo = -q.v_max; hi = q.v_max;
  } else {
    const int a = (row - 2 * N3) % 3;
    if (a == 2) { lo = q.tz_lo; hi = q.tz_hi; } else { lo = -q.txy; hi = q.txy; }
  }
}

// ---- host-side error plumbing -------------------------------------------------------------
void set_last_error(const char* what, hipError_t e);
int check_params_impl(const se3mpc_params* p);

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(what, e); return SE3MPC_ERR_LAUNCH; }
  return SE3MPC_OK;
}

inline int grid_for(int B, int block) { return (B + block - 1) / block; }

}  // namespace se3mpc
