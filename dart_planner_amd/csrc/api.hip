// api.hip -- library-level entry points of libse3mpc: version, defaults, validation, error text.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "se3mpc_common.hpp"

static_assert(sizeof(se3mpc_params) == 160, "se3mpc_params is part of the C ABI (dart_planner_amd/capi.py mirrors it)");
static_assert(sizeof(se3mpc_solve_info) == 24, "se3mpc_solve_info is part of the C ABI");

namespace se3mpc {

static thread_local char g_last_error[256] = "";

void set_last_error(const char* what, hipError_t e) {
  std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

int check_params_impl(const se3mpc_params* p) {
  if (p == nullptr) return SE3MPC_ERR_NULL;
  if (p->horizon < 1 || p->horizon > SE3MPC_MAX_HORIZON) return SE3MPC_ERR_HORIZON;
  const double pos[] = {p->dt, p->mass};
  for (double v : pos)
    if (!(v > 0.0) || !std::isfinite(v)) return SE3MPC_ERR_PARAM;
  const double fin[] = {p->gravity, p->position_weight, p->velocity_weight, p->acceleration_weight, p->thrust_weight,
                        p->terminal_factor, p->position_bound, p->max_velocity, p->max_acceleration, p->max_thrust,
                        p->min_thrust, p->max_tilt_angle, p->safety_margin, p->pgtol, p->ftol};
  for (double v : fin)
    if (!std::isfinite(v)) return SE3MPC_ERR_PARAM;
  if (p->max_thrust < p->min_thrust || p->position_bound < 0 || p->max_velocity < 0) return SE3MPC_ERR_PARAM;
  if (p->max_corrections < 1 || p->max_corrections > SE3MPC_MAX_CORRECTIONS) return SE3MPC_ERR_PARAM;
  if (p->max_iterations < 0 || p->max_linesearch < 1 || p->max_fun < 1) return SE3MPC_ERR_PARAM;
  return SE3MPC_OK;
}

}  // namespace se3mpc

extern "C" int se3mpc_abi_version(void) { return SE3MPC_ABI_VERSION; }

extern "C" const char* se3mpc_last_error(void) { return se3mpc::g_last_error; }

extern "C" int se3mpc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int good = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

extern "C" int se3mpc_default_params(se3mpc_params* out) {
  if (out == nullptr) return SE3MPC_ERR_NULL;
  std::memset(out, 0, sizeof(*out));
  out->horizon = 6;
  out->has_goal = 1;
  out->dt = 1.0 / 400.0;
  out->mass = 1.5;
  out->gravity = 9.81;
  out->position_weight = 100.0;
  out->velocity_weight = 10.0;
  out->acceleration_weight = 1.0;
  out->thrust_weight = 0.1;
  out->terminal_factor = 10.0;
  out->position_bound = 100.0;
  out->max_velocity = 10.0;
  out->max_acceleration = 15.0;
  out->max_thrust = 25.0;
  out->min_thrust = 2.0;
  out->max_tilt_angle = M_PI / 4.0;
  out->safety_margin = 1.5;
  out->max_iterations = 15;
  out->max_corrections = 10;
  out->max_linesearch = 20;
  out->max_fun = 15000;
  out->pgtol = 0.05;
  out->ftol = 0.5;
  return SE3MPC_OK;
}

extern "C" int se3mpc_check_params(const se3mpc_params* p) { return se3mpc::check_params_impl(p); }

extern "C" uint32_t se3mpc_key_index(uint64_t key) { return (uint32_t)(key & 0xFFFFFFFFull); }

extern "C" float se3mpc_key_cost(uint64_t key) {
  uint32_t u = (uint32_t)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  float f;
  std::memcpy(&f, &u, sizeof(f));
  return f;
}
