"""CPU oracle for the consumer side of the Planner->Controller contract -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement, batched over a leading drone axis, of

* ``GeometricController.compute_control`` / ``compute_body_rate_command`` with everything they call
  (``_update_integral_error``, the two anti-windup methods, ``_clamp_integral_per_axis``,
  ``_check_tracking_performance``, ``_geometric_attitude_control``, the yaw-singularity detection and its three
  fallbacks, ``_get_failsafe_command``) -- /root/reference/src/dart_planner/control/geometric_controller.py
  ("controller.py" below), gains from control_config.py:95-111 (profile "sitl_optimized");
* ``compute_control_fast`` / ``compute_control_from_fast_state`` (controller.py:253-411, :728-768), the unit-free variant the
  reference's 400 Hz hardware loop calls (hardware/pixhawk_interface.py:401);
* ``DroneSimulator.step`` -- /root/reference/src/dart_planner/utils/drone_simulator.py:52-72 ("simulator.py");
* ``OnboardController._interpolate_trajectory`` -- /root/reference/src/dart_planner/control/onboard_controller.py:43-93
  ("onboard.py"), the reference's own plan sampler, which is what the glue
  ``compute_control_from_trajectory(state, trajectory, t)`` (a stub in the reference, controller.py:873-875) composes
  with ``compute_control``.

Unit-stripped like the planner oracle (every quantity an SI magnitude).  The reference's quirks are kept as they are:
the "thrust" is the norm of an ACCELERATION (no mass factor, controller.py:461-462) yet is compared with newton limits
and handed to the simulator as newtons; b3 is divided by the SATURATED magnitude (:487); a successful command resets
``failsafe_count`` (:507), so the tracking failsafe (:485) can never fire; every new failsafe activation halves the
gains again (:817-821); the simulator's translation ignores attitude (simulator.py:59).

Pinning: ``tests/test_controller_oracle_golden.py`` checks every function here against vectors produced by running the
reference's own classes in the build container (``tests/golden/make_golden_controller.py`` ->
``tests/golden/controller_cases.npz``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline
leg import this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np


def _a(*v):
    return np.array(v, dtype=float)


@dataclass
class ControllerConfig:
    """GeometricControllerConfig (controller.py:26-77) after ``_apply_tuning_profile("sitl_optimized")`` (:140-158,
    control_config.py:95-111), with the values the reference instantiates in this container: mass / gravity / inertia from
    ``VehicleParams`` (common/vehicle_params.py:19-23), max_torque_xyz = the safe default of
    ``compute_max_torque_xyz`` (:108-115)."""
    kp_pos: np.ndarray = field(default_factory=lambda: _a(20.0, 20.0, 25.0))
    ki_pos: np.ndarray = field(default_factory=lambda: _a(1.5, 1.5, 2.0))
    kd_pos: np.ndarray = field(default_factory=lambda: _a(10.0, 10.0, 12.0))
    kp_att: np.ndarray = field(default_factory=lambda: _a(18.0, 18.0, 8.0))
    kd_att: np.ndarray = field(default_factory=lambda: _a(7.0, 7.0, 3.5))
    inertia: np.ndarray = field(default_factory=lambda: _a(0.02, 0.02, 0.04))
    max_torque_xyz: np.ndarray = field(default_factory=lambda: _a(0.5, 0.5, 0.05))
    max_integral_pos: float = 2.5
    max_tilt_angle: float = np.pi / 4
    mass: float = 1.0
    gravity: float = 9.80665
    max_thrust: float = 22.0
    min_thrust: float = 0.8
    tracking_error_threshold: float = 1.0
    velocity_error_threshold: float = 0.6
    anti_windup_method: str = "clamping"                 # or "back_calculation"
    max_integral_per_axis: np.ndarray = field(default_factory=lambda: _a(2.0, 2.0, 3.0))
    back_calculation_gain: float = 0.1
    integral_decay_factor: float = 0.99
    saturation_threshold: float = 0.95
    yaw_singularity_threshold: float = 0.1
    yaw_singularity_fallback_method: str = "skip_yaw"    # "default_heading", "maintain_current"
    default_heading_yaw: float = 0.0


@dataclass
class SimulatorConfig:
    """DroneSimulator.__init__ (simulator.py:41-50)."""
    mass: float = 1.5
    gravity: float = 9.81
    inertia: np.ndarray = field(default_factory=lambda: _a(0.1, 0.1, 0.2))
    max_thrust: float = 20.0
    max_torque: float = 10.0


class ControllerState:
    """The mutable members of GeometricController that compute_control reads or writes (controller.py:87-105), one row
    per drone.  ``halvings`` counts failsafe activations' gain halvings (:817-821: the gains are multiplied by 0.5
    -- exact in binary -- on every NEW activation)."""

    def __init__(self, B: int, cfg: ControllerConfig):
        self.integral = np.zeros((B, 3))                             # integral_vel_error
        self.last_time = np.full(B, np.nan)                          # None
        self.failsafe_active = np.zeros(B, dtype=bool)
        self.failsafe_count = np.zeros(B, dtype=np.int64)
        self.halvings = np.zeros(B, dtype=np.int64)
        self.last_valid_thrust = np.full(B, cfg.mass * cfg.gravity)  # :105
        self.thrust_saturated = np.zeros(B, dtype=bool)
        self.torque_saturated = np.zeros((B, 3), dtype=bool)
        self.unsaturated_thrust = np.zeros(B)
        self.unsaturated_torque = np.zeros((B, 3))

    def copy(self) -> "ControllerState":
        c = object.__new__(ControllerState)
        for k, v in vars(self).items():
            setattr(c, k, v.copy())
        return c


def euler_to_rotation_matrix(att):
    """controller.py:774-789, batched: att (B,3) -> R (B,3,3)."""
    roll, pitch, yaw = att[:, 0], att[:, 1], att[:, 2]
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.empty((len(att), 3, 3))
    R[:, 0, 0] = cy * cp; R[:, 0, 1] = cy * sp * sr - sy * cr; R[:, 0, 2] = cy * sp * cr + sy * sr
    R[:, 1, 0] = sy * cp; R[:, 1, 1] = sy * sp * sr + cy * cr; R[:, 1, 2] = sy * sp * cr - cy * sr
    R[:, 2, 0] = -sp; R[:, 2, 1] = cp * sr; R[:, 2, 2] = cp * cr
    return R


def _norm(v):
    return np.sqrt(np.sum(v * v, axis=-1))


def _failsafe(st: ControllerState, m: np.ndarray) -> None:
    """_get_failsafe_command (controller.py:813-828) for the drones selected by mask m."""
    new = m & ~st.failsafe_active
    st.halvings[new] += 1                                            # :817-821
    st.integral[new] = 0.0                                           # :823
    st.failsafe_count[new] += 1                                      # :825
    st.failsafe_active[m] = True                                     # :827


def update_integral_error(st: ControllerState, cfg: ControllerConfig, vel_error, dt, thrust_sat, m) -> None:
    """_update_integral_error (controller.py:548-578) + the anti-windup method (:580-623) + _clamp_integral_per_axis
    (:625-643), for the drones in mask m.  ``st.torque_saturated`` / ``st.unsaturated_torque`` still hold the PREVIOUS
    call's values here (the current torque is computed afterwards, :480 vs :494)."""
    upd = vel_error * dt[:, None]
    if cfg.anti_windup_method == "clamping":
        upd = np.where(thrust_sat[:, None], upd * 0.1, upd)          # :590-591
        upd = np.where(st.torque_saturated, upd * 0.1, upd)          # :594-596
    elif cfg.anti_windup_method == "back_calculation":
        Kb = cfg.back_calculation_gain
        fb = (st.unsaturated_thrust - cfg.max_thrust) * Kb           # :611
        upd = np.where(thrust_sat[:, None], upd - fb[:, None] * _a(0.33, 0.33, 0.34), upd)   # :613
        tfb = (st.unsaturated_torque - cfg.max_torque_xyz) * Kb      # :618
        upd = np.where(st.torque_saturated, upd - tfb * 0.5, upd)    # :620
    I = st.integral + upd                                            # :574
    lim = cfg.max_integral_per_axis
    I = np.where(np.abs(I) > lim, np.sign(I) * lim, I)               # :630-632
    mag = _norm(I)
    with np.errstate(divide="ignore", invalid="ignore"):
        I = np.where((mag > cfg.max_integral_pos)[:, None], I * (cfg.max_integral_pos / mag)[:, None], I)   # :635-637
    I = np.where(np.abs(I) > lim * cfg.saturation_threshold, I * cfg.integral_decay_factor, I)            # :640-643
    st.integral[m] = I[m]


def desired_frame(cfg: ControllerConfig, b3, att, yaw_des):
    """The desired rotation of _geometric_attitude_control (controller.py:665-690) with the yaw-singularity detection
    (:160-189) and its fallbacks (:191-257).  b3 (B,3) is normalised here (:667).  -> b1, b2, b3n, is_singular."""
    B = len(b3)
    yaw_vector = np.stack([np.cos(yaw_des), np.sin(yaw_des), np.zeros(B)], axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        b3n = b3 / _norm(b3)[:, None]
        cos_angle = np.abs(np.sum(yaw_vector * b3n, axis=1))         # :174
        singular = cos_angle >= cfg.yaw_singularity_threshold        # :177
        # normal case (:680-689)
        b1 = np.cross(yaw_vector, b3n)
        n1 = _norm(b1)
        b1 = np.where((n1 > 1e-6)[:, None], b1 / n1[:, None], _a(1.0, 0.0, 0.0))
        # skip_yaw construction (:212-220), also the last resort of the two other methods
        ex = np.tile(_a(1.0, 0.0, 0.0), (B, 1))
        proj = ex - b3n[:, :1] * b3n                                  # [1,0,0] - ([1,0,0] . b3) b3
        proj = proj / _norm(proj)[:, None]
        method = cfg.yaw_singularity_fallback_method
        if method == "default_heading" or method == "maintain_current":
            yv = (np.tile(_a(np.cos(cfg.default_heading_yaw), np.sin(cfg.default_heading_yaw), 0.0), (B, 1)) if method == "default_heading"
                  else np.stack([np.cos(att[:, 2]), np.sin(att[:, 2]), np.zeros(B)], axis=1))
            c = np.cross(yv, b3n)
            nc = _norm(c)
            b1s = np.where((nc > 1e-6)[:, None], c / nc[:, None], proj)      # :226-234 / :239-247
        elif method == "skip_yaw":
            b1s = np.where((np.abs(b3n[:, 2]) < 0.99)[:, None], proj, ex)   # :214-220
        else:                                                        # unknown method (:248-252): the projection, no 0.99 test
            b1s = proj
        b1 = np.where(singular[:, None], b1s, b1)
    b2 = np.cross(b3n, b1)                                           # :255 / :689
    return b1, b2, b3n, singular


def compute_control(st: ControllerState, cfg: ControllerConfig, t, pos, vel, att, omega, dpos, dvel, dacc, yaw_des=None,
                    yaw_rate_des=None) -> Tuple[np.ndarray, np.ndarray, Dict[str, np.ndarray]]:
    """GeometricController.compute_control (controller.py:413-512) for B drones at once; mutates ``st``.
    t (B,) = current_state.timestamp.  -> thrust (B,), torque (B,3), flags."""
    B = len(t)
    yaw_des = np.zeros(B) if yaw_des is None else np.asarray(yaw_des, float)
    yaw_rate_des = np.zeros(B) if yaw_rate_des is None else np.asarray(yaw_rate_des, float)
    first = np.isnan(st.last_time)
    dt = np.where(first, 0.001, t - np.where(first, 0.0, st.last_time))      # :440
    st.last_time = np.array(t, dtype=float)                                   # :441
    bad_dt = (dt <= 0) | (dt > 0.1)                                           # :442
    ok = ~bad_dt
    scale = 0.5 ** st.halvings.astype(float)
    kp_pos, kd_pos = cfg.kp_pos * scale[:, None], cfg.kd_pos * scale[:, None]
    kp_att, kd_att = cfg.kp_att * scale[:, None], cfg.kd_att * scale[:, None]
    pos_error, vel_error = dpos - pos, dvel - vel                             # :445-446
    pe, ve = _norm(pos_error), _norm(vel_error)
    acc_pid = kp_pos * pos_error + kd_pos * vel_error + cfg.ki_pos * st.integral   # :453-457
    acc_des = dacc + acc_pid
    tvw = acc_des - _a(0.0, 0.0, -cfg.gravity)                                # :461 (ENU gravity vector)
    tm = _norm(tvw)                                                           # :462
    st.unsaturated_thrust[ok] = tm[ok]                                        # :465
    min_thrust = cfg.min_thrust * cfg.mass * cfg.gravity                      # :469
    hi, lo = tm > cfg.max_thrust, tm < min_thrust
    tm = np.where(hi, cfg.max_thrust, np.where(lo, min_thrust, tm))           # :470-475
    thrust_sat = hi | lo
    st.thrust_saturated[ok] = thrust_sat[ok]                                  # :477
    update_integral_error(st, cfg, vel_error, dt, thrust_sat, ok)             # :480
    # _check_tracking_performance (:650-658)
    poor = (pe > cfg.tracking_error_threshold) & (ve > cfg.velocity_error_threshold)
    cnt = np.where(poor, st.failsafe_count + 1, np.maximum(0, st.failsafe_count - 1))
    st.failsafe_count[ok] = cnt[ok]
    tracking_fs = ok & (st.failsafe_count > 100)                              # :485-486
    run = ok & ~tracking_fs
    with np.errstate(divide="ignore", invalid="ignore"):
        b3 = np.where((tm > 1e-6)[:, None], tvw / tm[:, None], _a(0.0, 0.0, 1.0))   # :487-490 (divides by the saturated magnitude)
        tilt = np.arccos(np.clip(b3[:, 2], -1, 1))                            # :491
        over = tilt > cfg.max_tilt_angle
        ct = np.cos(cfg.max_tilt_angle)
        sf = ct / b3[:, 2]                                                    # :493
        b3t = np.stack([b3[:, 0] * sf, b3[:, 1] * sf, np.full(B, ct)], axis=1)
        b3t = b3t / _norm(b3t)[:, None]                                       # :494-496
        b3 = np.where(over[:, None], b3t, b3)
        b1, b2, b3n, singular = desired_frame(cfg, b3, att, yaw_des)
    R = euler_to_rotation_matrix(att)
    Rd = np.stack([b1, b2, b3n], axis=2)                                      # column_stack
    M = np.einsum("bki,bkj->bij", Rd, R) - np.einsum("bki,bkj->bij", R, Rd)   # R_des.T @ R - R.T @ R_des (:692)
    eR = 0.5 * np.stack([M[:, 2, 1], M[:, 0, 2], M[:, 1, 0]], axis=1)
    eOmega = omega - np.stack([np.zeros(B), np.zeros(B), yaw_rate_des], axis=1)   # :693-695
    coriolis = np.cross(omega, cfg.inertia * omega)                           # :700
    torque = -kp_att * eR - kd_att * eOmega + coriolis                        # :701
    st.unsaturated_torque[run] = torque[run]                                  # :704
    sat = np.abs(torque) > cfg.max_torque_xyz                                 # :708-711
    torque = np.where(sat, np.sign(torque) * cfg.max_torque_xyz, torque)
    st.torque_saturated[run] = sat[run]                                       # :713
    # success bookkeeping (:505-507)
    st.last_valid_thrust[run] = tm[run]
    st.failsafe_active[run] = False
    st.failsafe_count[run] = 0
    # failsafe returns (:443, :486)
    fs = bad_dt | tracking_fs
    _failsafe(st, fs)
    thrust = np.where(fs, st.last_valid_thrust, tm)
    torque = np.where(fs[:, None], 0.0, torque)
    return thrust, torque, dict(failsafe=fs, bad_dt=bad_dt, thrust_saturated=thrust_sat & ok, singular=singular & run,
                                tilt_limited=over & run, dt=dt)


@dataclass
class VehicleConstants:
    """``get_control_constants()`` (common/vehicle_params.py:68-77 over the ``VehicleParams`` defaults, :19-23): the mass and gravity
    the FAST path uses (controller.py:118-127) -- not the controller config's."""
    mass: float = 1.0
    gravity: float = 9.80665


def compute_control_fast(st: ControllerState, cfg: ControllerConfig, veh: VehicleConstants, dt, pos, vel, att, omega, dpos, dvel, dacc,
                         yaw_des=None, yaw_rate_des=None) -> Tuple[np.ndarray, np.ndarray, Dict[str, np.ndarray]]:
    """GeometricController.compute_control_fast (controller.py:253-346) with _fast_geometric_attitude_control (:348-411) -- what
    compute_control_from_fast_state (:728-768) forwards to, the unit-free path of the reference's 400 Hz loop
    (hardware/pixhawk_interface.py:401) -- for B drones at once; mutates ``st``.  dt: scalar or (B,).  Against compute_control: dt is an
    argument and an invalid one returns the vehicle's hover thrust with NO failsafe and no state change (:279-280); gravity and the lower
    thrust limit come from the vehicle constants (:303, :317, :127); no tracking check; last_time / last_valid_thrust / failsafe_active /
    failsafe_count are not touched (the gains stay halved if compute_control halved them).  -> thrust (B,), torque (B,3), flags (the
    reference counts the saturations it reports here, :315-320, :404)."""
    B = len(pos)
    dt = np.broadcast_to(np.asarray(dt, float), (B,))
    yaw_des = np.zeros(B) if yaw_des is None else np.asarray(yaw_des, float)
    yaw_rate_des = np.zeros(B) if yaw_rate_des is None else np.asarray(yaw_rate_des, float)
    bad_dt = (dt <= 0) | (dt > 0.1)                                           # :279
    run = ~bad_dt
    scale = 0.5 ** st.halvings.astype(float)
    kp_pos, kd_pos = cfg.kp_pos * scale[:, None], cfg.kd_pos * scale[:, None]
    kp_att, kd_att = cfg.kp_att * scale[:, None], cfg.kd_att * scale[:, None]
    pos_error, vel_error = dpos - pos, dvel - vel                             # :283-284
    acc_pid = kp_pos * pos_error + kd_pos * vel_error + cfg.ki_pos * st.integral   # :293-297
    tvw = (dacc + acc_pid) - _a(0.0, 0.0, -veh.gravity)                       # :298, :301
    tm = _norm(tvw)                                                           # :302
    st.unsaturated_thrust[run] = tm[run]                                      # :305
    min_thrust = cfg.min_thrust * veh.mass * veh.gravity                      # :127
    hi, lo = tm > cfg.max_thrust, tm < min_thrust
    tm = np.where(hi, cfg.max_thrust, np.where(lo, min_thrust, tm))           # :312-320
    thrust_sat = hi | lo
    st.thrust_saturated[run] = thrust_sat[run]                                # :322
    update_integral_error(st, cfg, vel_error, dt, thrust_sat, run)            # :325
    with np.errstate(divide="ignore", invalid="ignore"):
        b3 = np.where((tm > 1e-6)[:, None], tvw / tm[:, None], _a(0.0, 0.0, 1.0))   # :328-331
        tilt = np.arccos(np.clip(b3[:, 2], -1, 1))                            # :334
        over = tilt > cfg.max_tilt_angle
        ct = np.cos(cfg.max_tilt_angle)
        sf = ct / b3[:, 2]
        b3t = np.stack([b3[:, 0] * sf, b3[:, 1] * sf, np.full(B, ct)], axis=1)
        b3t = b3t / _norm(b3t)[:, None]                                       # :335-339
        b3 = np.where(over[:, None], b3t, b3)
        b1, b2, b3n, singular = desired_frame(cfg, b3, att, yaw_des)          # :359-385
    R = euler_to_rotation_matrix(att)
    Rd = np.stack([b1, b2, b3n], axis=2)
    M = np.einsum("bki,bkj->bij", Rd, R) - np.einsum("bki,bkj->bij", R, Rd)   # :388
    eR = 0.5 * np.stack([M[:, 2, 1], M[:, 0, 2], M[:, 1, 0]], axis=1)
    eOmega = omega - np.stack([np.zeros(B), np.zeros(B), yaw_rate_des], axis=1)   # :391-392
    coriolis = np.cross(omega, cfg.inertia * omega)                           # :396 (np.diag(inertia) @ omega)
    torque = -kp_att * eR - kd_att * eOmega + coriolis                        # :397
    st.unsaturated_torque[run] = torque[run]                                  # :400
    sat = np.abs(torque) > cfg.max_torque_xyz                                 # :403-408
    torque = np.where(sat, np.sign(torque) * cfg.max_torque_xyz, torque)
    st.torque_saturated[run] = sat[run]                                       # :410
    thrust = np.where(bad_dt, veh.mass * veh.gravity, tm)                     # :280
    torque = np.where(bad_dt[:, None], 0.0, torque)
    return thrust, torque, dict(bad_dt=bad_dt, thrust_saturated=thrust_sat & run, torque_saturated=sat & run[:, None], singular=singular & run,
                                tilt_limited=over & run)


def compute_body_rate_command(st, cfg, t, pos, vel, att, omega, dpos, dvel, dacc, yaw_des=None, yaw_rate_des=None):
    """compute_body_rate_command (controller.py:706-726): -> normalised thrust (B,), body rates (B,3), thrust, torque."""
    thrust, torque, _ = compute_control(st, cfg, t, pos, vel, att, omega, dpos, dvel, dacc, yaw_des, yaw_rate_des)
    angular_accel = torque / _a(0.1, 0.1, 0.2)                                # :717-718
    body_rates = omega + angular_accel * 0.001                                # :719-720
    return np.clip(thrust / cfg.max_thrust, 0.0, 1.0), body_rates, thrust, torque


def interpolate_trajectory(t, timestamps, P, V=None, A=None):
    """OnboardController._interpolate_trajectory (onboard.py:43-93), batched over B sample times.
    t (B,); timestamps (N,) shared or (B,N); P, V, A (N,3) shared or (B,N,3) -> target pos, vel, acc (B,3)."""
    t = np.asarray(t, float)
    B = len(t)
    ts = np.broadcast_to(np.asarray(timestamps, float), (B, np.shape(timestamps)[-1]))
    n = ts.shape[1]
    bc = lambda X: None if X is None else np.broadcast_to(np.asarray(X, float), (B, n, 3))
    P, V, A = bc(P), bc(V), bc(A)
    idx = np.sum(ts < t[:, None], axis=1)                                     # np.searchsorted(ts, t) (side="left")
    rows = np.arange(B)
    i2 = np.clip(idx, 1, n - 1) if n > 1 else np.zeros(B, dtype=int)
    i1 = i2 - 1 if n > 1 else i2
    with np.errstate(divide="ignore", invalid="ignore"):
        f = (t - ts[rows, i1]) / (ts[rows, i2] - ts[rows, i1])                # :80
    first, last = idx == 0, idx >= n

    def pick(X):
        if X is None:
            return np.zeros((B, 3))
        mid = X[rows, i1] + f[:, None] * (X[rows, i2] - X[rows, i1])          # :81, :86, :91
        return np.where(first[:, None], X[:, 0], np.where(last[:, None], X[:, -1], mid))
    return pick(P), pick(V), pick(A)


def simulator_step(sim: SimulatorConfig, pos, vel, att, omega, t, thrust, torque, dt, wind=None):
    """DroneSimulator.step (simulator.py:52-72), batched.  wind (3,) or (B,3) in newtons.  -> new (pos, vel, att, omega, t)."""
    B = len(thrust)
    wind = np.zeros(3) if wind is None else np.asarray(wind, float)
    thrust = np.clip(thrust, 0, sim.max_thrust)                               # :54
    torque = np.clip(torque, -sim.max_torque, sim.max_torque)                 # :55
    wind_accel = wind / sim.mass                                              # :57
    acc = _a(0, 0, -sim.gravity) + np.stack([np.zeros(B), np.zeros(B), thrust / sim.mass], axis=1) + wind_accel   # :59
    new_vel = vel + acc * dt                                                  # :60
    new_pos = pos + new_vel * dt                                              # :61
    angular_accel = torque / sim.inertia                                      # :63 (np.linalg.solve with a diagonal matrix)
    new_omega = omega + angular_accel * dt                                    # :64
    new_att = att + new_omega * dt                                            # :65
    return new_pos, new_vel, new_att, new_omega, t + dt


def closed_loop(cfg: ControllerConfig, sim: SimulatorConfig, st: ControllerState, pos, vel, att, omega, t, timestamps, P, V, A,
                nsteps: int, sim_dt: float, wind=None, gust_step: Optional[int] = None, gust_wind=None,
                stop_at_plan_end: bool = True, log: bool = True):
    """The loop of the reference's closed-loop contract tests (tests/test_planner_controller_contract.py:115-162,
    :255-316): per step  t = state.timestamp; [stop this drone once t > timestamps[-1]];  target = plan sampled at t;
    cmd = compute_control(state, target);  [the gust replaces the wind at `gust_step`, :293-296];  state = simulator.step.
    A stopped drone keeps its state (`break`).  -> final state, per-step logs (states BEFORE each step, commands)."""
    pos, vel, att, omega, t = (np.array(a, dtype=float) for a in (pos, vel, att, omega, t))
    B = len(t)
    wind = np.zeros((B, 3)) if wind is None else np.broadcast_to(np.asarray(wind, float), (B, 3)).copy()
    active = np.ones(B, dtype=bool)
    ts_last = np.broadcast_to(np.asarray(timestamps, float), (B, np.shape(timestamps)[-1]))[:, -1]
    logs = dict(pos=[], vel=[], att=[], omega=[], t=[], thrust=[], torque=[], active=[], failsafe=[])
    for step in range(nsteps):
        if stop_at_plan_end:
            active &= ~(t > ts_last)                                          # :130-131 / :263-264
        tp, tv, ta = interpolate_trajectory(t, timestamps, P, V, A)
        before = st.copy()
        thrust, torque, fl = compute_control(st, cfg, t, pos, vel, att, omega, tp, tv, ta)
        for k, v in vars(st).items():                                         # a stopped drone's controller is not called again
            v[~active] = getattr(before, k)[~active]
        if gust_step is not None and step == gust_step:
            wind = np.broadcast_to(np.asarray(gust_wind, float), (B, 3)).copy()
        n = simulator_step(sim, pos, vel, att, omega, t, thrust, torque, sim_dt, wind)
        if log:
            for k, v in (("pos", pos), ("vel", vel), ("att", att), ("omega", omega), ("t", t), ("thrust", thrust), ("torque", torque),
                         ("active", active), ("failsafe", fl["failsafe"])):
                logs[k].append(np.array(v))
        a3 = active[:, None]
        pos, vel, att, omega = (np.where(a3, new, old) for new, old in zip(n[:4], (pos, vel, att, omega)))
        t = np.where(active, n[4], t)
    return dict(pos=pos, vel=vel, att=att, omega=omega, t=t, active=active), {k: np.array(v) for k, v in logs.items()}
