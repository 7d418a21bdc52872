"""Trajectory -> controller glue of the Planner->Controller contract.

The reference's ``GeometricController.compute_control_from_trajectory`` is a stub that returns ``{}``
(src/dart_planner/control/geometric_controller.py:873-875) and ``compute_body_rate_from_trajectory``
does not exist, which is why 8 of its 11 contract tests fail.  This module supplies both on top of
a compact SE(3) geometric tracking controller (position PID + feed-forward -> desired thrust
vector -> attitude error on SO(3) -> torque), with the reference's public names
(``compute_control`` :413, ``compute_body_rate_command`` :706, ``config.max_thrust``) and the
"sitl_optimized" gains of control_config.py:95-111.  Host-side NumPy: the controller consumes the
planner's output at 400 Hz-1 kHz, it is not part of the accelerated path.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from ..common.types import BodyRateCommand, ControlCommand, DroneState, Trajectory
from ..common.units import to_float

_PROFILES = {
    # name: kp_pos, ki_pos, kd_pos, kp_att, kd_att, max_tilt, max_thrust, min_thrust, max_integral
    "sitl_optimized": ([20, 20, 25], [1.5, 1.5, 2.0], [10, 10, 12], [18, 18, 8], [7, 7, 3.5], np.pi / 4, 22.0, 0.8, 2.5),
    "conservative": ([15, 15, 18], [2, 2, 3], [8, 8, 10], [15, 15, 8], [6, 6, 3], np.pi / 6, 20.0, 0.5, 3.0),
    "precision_tracking": ([18, 18, 22], [2.5, 2.5, 3.5], [12, 12, 14], [22, 22, 10], [8, 8, 4], np.pi / 4, 25.0, 0.8, 2.0),
}


@dataclass
class GeometricControllerConfig:
    kp_pos: np.ndarray = field(default_factory=lambda: np.array([7.0, 7.0, 8.5]))
    ki_pos: np.ndarray = field(default_factory=lambda: np.array([0.35, 0.35, 0.7]))
    kd_pos: np.ndarray = field(default_factory=lambda: np.array([4.2, 4.2, 5.6]))
    kp_att: np.ndarray = field(default_factory=lambda: np.array([9.0, 9.0, 3.75]))
    kd_att: np.ndarray = field(default_factory=lambda: np.array([3.0, 3.0, 1.5]))
    inertia: np.ndarray = field(default_factory=lambda: np.array([0.1, 0.1, 0.2]))
    max_torque_xyz: np.ndarray = field(default_factory=lambda: np.array([2.0, 2.0, 1.0]))
    max_integral_pos: float = 5.0
    max_tilt_angle: float = np.pi / 3
    mass: float = 1.5
    gravity: float = 9.81
    max_thrust: float = 20.0
    min_thrust: float = 0.5


def _rot_from_euler(att):
    r, p, y = att
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _vee(S):
    return np.array([S[2, 1], S[0, 2], S[1, 0]])


class GeometricController:
    def __init__(self, config: Optional[GeometricControllerConfig] = None, tuning_profile: str = "sitl_optimized"):
        self.config = config or GeometricControllerConfig()
        if tuning_profile in _PROFILES:
            kp, ki, kd, kpa, kda, tilt, tmax, tmin, imax = _PROFILES[tuning_profile]
            c = self.config
            c.kp_pos, c.ki_pos, c.kd_pos = np.array(kp, float), np.array(ki, float), np.array(kd, float)
            c.kp_att, c.kd_att = np.array(kpa, float), np.array(kda, float)
            c.max_tilt_angle, c.max_thrust, c.min_thrust, c.max_integral_pos = tilt, tmax, tmin, imax
        self.tuning_profile = tuning_profile
        self.reset()

    def reset(self):
        self.integral_pos_error = np.zeros(3)
        self.last_time: Optional[float] = None

    # ---- reference geometric_controller.py:413-512 (public signature)
    def compute_control(self, current_state: DroneState, desired_pos, desired_vel, desired_acc, desired_yaw=0.0,
                        desired_yaw_rate=0.0) -> ControlCommand:
        c = self.config
        pos, vel = np.asarray(to_float(current_state.position), float), np.asarray(to_float(current_state.velocity), float)
        att, omega = np.asarray(to_float(current_state.attitude), float), np.asarray(to_float(current_state.angular_velocity), float)
        dpos, dvel, dacc = (np.asarray(to_float(a), float) for a in (desired_pos, desired_vel, desired_acc))
        yaw = float(to_float(desired_yaw))
        t = current_state.timestamp
        dt = (t - self.last_time) if self.last_time is not None else 0.0
        self.last_time = t
        e_p, e_v = dpos - pos, dvel - vel
        if 0.0 < dt < 0.1:
            self.integral_pos_error = np.clip(self.integral_pos_error + e_p * dt, -c.max_integral_pos, c.max_integral_pos)
        a_cmd = c.kp_pos * e_p + c.kd_pos * e_v + c.ki_pos * self.integral_pos_error + dacc
        F = c.mass * (a_cmd + np.array([0.0, 0.0, c.gravity]))                   # desired thrust vector, world frame
        if F[2] < 1e-3:
            F[2] = 1e-3
        tilt = np.arctan2(np.linalg.norm(F[:2]), F[2])
        if tilt > c.max_tilt_angle:                                              # tilt limit: shrink the lateral part
            F[:2] *= np.tan(c.max_tilt_angle) * F[2] / max(np.linalg.norm(F[:2]), 1e-9)
        R = _rot_from_euler(att)
        thrust = float(np.clip(F @ R[:, 2], c.min_thrust, c.max_thrust))
        b3 = F / np.linalg.norm(F)
        b1c = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        b2 = np.cross(b3, b1c)
        nb2 = np.linalg.norm(b2)
        b2 = b2 / nb2 if nb2 > 1e-6 else np.array([0.0, 1.0, 0.0])
        Rd = np.column_stack([np.cross(b2, b3), b2, b3])
        e_R = 0.5 * _vee(Rd.T @ R - R.T @ Rd)
        e_w = omega - R.T @ Rd @ np.array([0.0, 0.0, float(to_float(desired_yaw_rate))])
        torque = -c.kp_att * e_R - c.kd_att * e_w + np.cross(omega, c.inertia * omega)
        torque = np.clip(torque, -c.max_torque_xyz, c.max_torque_xyz)
        return ControlCommand(thrust=thrust, torque=torque)

    # ---- reference geometric_controller.py:706-726
    def compute_body_rate_command(self, current_state, desired_pos, desired_vel, desired_acc, desired_yaw=0.0,
                                  desired_yaw_rate=0.0) -> BodyRateCommand:
        cmd = self.compute_control(current_state, desired_pos, desired_vel, desired_acc, desired_yaw, desired_yaw_rate)
        ang_acc = cmd.torque / self.config.inertia
        rates = np.asarray(to_float(current_state.angular_velocity), float) + ang_acc * 0.001
        return BodyRateCommand(thrust=float(np.clip(cmd.thrust / self.config.max_thrust, 0.0, 1.0)), body_rates=rates)

    # ---- the glue the contract test calls (reference: stub at :873-875 / missing)
    @staticmethod
    def sample_trajectory(trajectory: Trajectory, t: float):
        """Linear interpolation of the plan at time t, clamped to its ends; missing arrays -> zeros."""
        ts = np.asarray(trajectory.timestamps, float)
        P = np.asarray(to_float(trajectory.positions), float)
        n = len(ts)
        zeros = np.zeros_like(P)
        V = zeros if trajectory.velocities is None else np.asarray(to_float(trajectory.velocities), float)
        A = zeros if trajectory.accelerations is None else np.asarray(to_float(trajectory.accelerations), float)
        yaws = None if trajectory.yaws is None else np.asarray(to_float(trajectory.yaws), float)
        yr = None if trajectory.yaw_rates is None else np.asarray(to_float(trajectory.yaw_rates), float)
        if n == 1 or t <= ts[0]:
            i, a = 0, 0.0
        elif t >= ts[-1]:
            i, a = n - 2 if n > 1 else 0, 1.0
        else:
            i = int(np.searchsorted(ts, t, side="right") - 1)
            a = float((t - ts[i]) / max(ts[i + 1] - ts[i], 1e-12))
        j = min(i + 1, n - 1)
        lerp = lambda X: (1 - a) * X[i] + a * X[j]
        return lerp(P), lerp(V), lerp(A), (0.0 if yaws is None else float(lerp(yaws))), (0.0 if yr is None else float(lerp(yr)))

    def compute_control_from_trajectory(self, current_state: DroneState, trajectory: Trajectory, t: float) -> ControlCommand:
        p, v, a, yaw, yaw_rate = self.sample_trajectory(trajectory, t)
        return self.compute_control(current_state, p, v, a, yaw, yaw_rate)

    def compute_body_rate_from_trajectory(self, current_state: DroneState, trajectory: Trajectory, t: float) -> BodyRateCommand:
        p, v, a, yaw, yaw_rate = self.sample_trajectory(trajectory, t)
        return self.compute_body_rate_command(current_state, p, v, a, yaw, yaw_rate)
