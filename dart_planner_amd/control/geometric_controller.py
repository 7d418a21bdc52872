"""GeometricController -- host-side mirror of the reference's controller, backed by the HIP kernels of
``dart_planner_amd/csrc/closed_loop.hip``.

Same class / method names, arguments and defaults as ``src/dart_planner/control/geometric_controller.py`` of
DART-Planner ("controller.py" below): ``GeometricControllerConfig`` (:26-77), the tuning profiles of
``control/control_config.py`` (:52-198), ``compute_control`` (:413-512), ``compute_body_rate_command`` (:706-726),
``reset`` (:840-860), ``get_performance_metrics``.  Every number is produced on the device (``se3mpc_control_*``,
``se3mpc_control_plan_*``): the controller's members -- integral, last time stamp, failsafe state, saturation flags --
live in a 12-word device record; there is no CPU fallback.

The trajectory glue the Planner->Controller contract test calls is a stub in the reference
(``compute_control_from_trajectory`` returns ``{}``, :873-875) and ``compute_body_rate_from_trajectory`` does not exist.
Here both are the composition of two reference functions: the plan sampled at ``t`` with
``OnboardController._interpolate_trajectory`` (control/onboard_controller.py:43-93), then ``compute_control`` with the
default yaw arguments.  For B drones at once, and for whole closed loops in one launch, use
``dart_planner_amd.ops.Ops.control`` / ``closed_loop`` directly (see ``dart_planner_amd/control/closed_loop.py``).
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from ..capi import CONTROLLER_STATE_WORDS, ControllerParams
from ..common.types import BodyRateCommand, ControlCommand, DroneState, Trajectory
from ..common.units import to_float

# control/control_config.py:52-198: kp_pos, ki_pos, kd_pos, kp_att, kd_att, ff_pos, ff_vel, max_tilt_angle, max_thrust, min_thrust,
# tracking_error_threshold, velocity_error_threshold, max_integral_pos
_PI = np.pi
TUNING_PROFILES = {
    "conservative": ([15, 15, 18], [2, 2, 3], [8, 8, 10], [15, 15, 8], [6, 6, 3], 1.5, 1.0, _PI / 6, 20.0, 0.5, 1.5, 0.8, 3.0),
    "aggressive": ([25, 25, 30], [1, 1, 1.5], [12, 12, 15], [20, 20, 10], [8, 8, 4], 2.0, 1.2, _PI / 3, 25.0, 1.0, 0.8, 0.5, 2.0),
    "sitl_optimized": ([20, 20, 25], [1.5, 1.5, 2.0], [10, 10, 12], [18, 18, 8], [7, 7, 3.5], 1.8, 1.1, _PI / 4, 22.0, 0.8, 1.0, 0.6, 2.5),
    "precision_tracking": ([18, 18, 22], [2.5, 2.5, 3.5], [12, 12, 14], [22, 22, 10], [8, 8, 4], 2.2, 1.3, _PI / 4, 25.0, 0.8, 0.8, 0.4, 2.0),
    "enhanced_tracking": ([22, 22, 28], [2.0, 2.0, 2.5], [11, 11, 13], [20, 20, 9], [8, 8, 4], 2.0, 1.2, _PI / 4, 24.0, 0.8, 0.9, 0.5, 2.2),
    "tracking_optimized": ([21, 21, 26], [1.8, 1.8, 2.2], [10.5, 10.5, 12.5], [18.5, 18.5, 8.2], [7.2, 7.2, 3.6], 1.9, 1.15, _PI / 4, 22.0, 0.8,
                           0.95, 0.55, 2.3),
    "original": ([10, 10, 12], [0.5, 0.5, 1.0], [6, 6, 8], [12, 12, 5], [4, 4, 2], 1.2, 0.8, _PI / 3, 20.0, 0.5, 2.0, 1.0, 5.0),
}


def _a(*v):
    return np.array(v, dtype=float)


@dataclass
class GeometricControllerConfig:
    """controller.py:26-77, same fields and defaults; mass / gravity / inertia are ``VehicleParams`` (common/vehicle_params.py:19-23),
    max_torque_xyz the safe default of ``compute_max_torque_xyz`` (:108-115) -- the values the reference instantiates."""
    kp_pos: np.ndarray = field(default_factory=lambda: _a(7.0, 7.0, 8.5))
    ki_pos: np.ndarray = field(default_factory=lambda: _a(0.35, 0.35, 0.7))
    kd_pos: np.ndarray = field(default_factory=lambda: _a(4.2, 4.2, 5.6))
    kp_att: np.ndarray = field(default_factory=lambda: _a(9.0, 9.0, 3.75))
    kd_att: np.ndarray = field(default_factory=lambda: _a(3.0, 3.0, 1.5))
    inertia: np.ndarray = field(default_factory=lambda: _a(0.02, 0.02, 0.04))
    max_torque_xyz: np.ndarray = field(default_factory=lambda: _a(0.5, 0.5, 0.05))
    ff_pos: float = 1.2                      # never read by the control law, as in the reference
    ff_vel: float = 0.8                      # never read
    max_integral_pos: float = 5.0
    max_tilt_angle: float = np.pi / 3
    mass: float = 1.0
    gravity: float = 9.80665
    max_thrust: float = 20.0
    min_thrust: float = 0.5
    tracking_error_threshold: float = 2.0
    velocity_error_threshold: float = 1.0
    anti_windup_method: str = "clamping"
    max_integral_per_axis: np.ndarray = field(default_factory=lambda: _a(2.0, 2.0, 3.0))
    back_calculation_gain: float = 0.1
    integral_decay_factor: float = 0.99
    saturation_threshold: float = 0.95
    yaw_singularity_threshold: float = 0.1
    yaw_singularity_fallback_method: str = "skip_yaw"
    default_heading_yaw: float = 0.0
    yaw_singularity_warning_threshold: float = 0.3   # logging only in the reference


class GeometricController:
    """controller.py:79-881 on the device, one drone."""

    def __init__(self, config: Optional[GeometricControllerConfig] = None, tuning_profile: str = "sitl_optimized", *,
                 precision: str = "f64", device=None, vehicle_mass: float = 1.0, vehicle_gravity: float = 9.80665):
        if config is None:
            config = GeometricControllerConfig()
        if tuning_profile:
            self._apply_tuning_profile(config, tuning_profile)
        self.config = config
        self.tuning_profile_name = tuning_profile
        self.precision = precision
        self._device = device
        self._ops = None
        self._state = None
        self.position_errors, self.velocity_errors, self.control_outputs = [], [], []
        self._thrust_saturation_count = 0
        self._torque_saturation_count = 0
        # get_control_constants() (common/vehicle_params.py:68-77): what the fast path takes mass and gravity from (controller.py:118-127)
        self._fast_mass, self._fast_gravity_magnitude = float(vehicle_mass), float(vehicle_gravity)
        self._inertia_override = None

    def _apply_tuning_profile(self, config: GeometricControllerConfig, profile_name: str) -> None:   # controller.py:140-158
        prof = TUNING_PROFILES.get(profile_name)
        if prof is None:
            return                                          # the reference logs a warning and keeps the defaults
        kp, ki, kd, kpa, kda, ffp, ffv, tilt, tmax, tmin, terr, verr, imax = prof
        config.kp_pos, config.ki_pos, config.kd_pos = _a(*kp), _a(*ki), _a(*kd)
        config.kp_att, config.kd_att = _a(*kpa), _a(*kda)
        config.ff_pos, config.ff_vel, config.max_tilt_angle = ffp, ffv, tilt
        config.max_thrust, config.min_thrust, config.max_integral_pos = tmax, tmin, imax
        config.tracking_error_threshold, config.velocity_error_threshold = terr, verr

    # ------------------------------------------------------------------ device plumbing
    def _get_ops(self):
        if self._ops is None:
            from ..ops import Ops, TorchBackend
            self._ops = Ops(TorchBackend(self._device))      # raises without a HIP device / built library
        return self._ops

    def _dev(self, a, kind=None):
        import torch
        dt = {"f32": torch.float32, "f64": torch.float64}[kind or self.precision]
        return torch.as_tensor(np.ascontiguousarray(np.asarray(to_float(a), dtype=float))).to(device=self._get_ops().be.device, dtype=dt)

    def _params(self) -> ControllerParams:
        return ControllerParams.from_config(self.config)

    def _members(self):
        if self._state is None:
            self._state = self._get_ops().controller_state(self._params(), 1)
        return self._state

    def _state_args(self, current_state: DroneState):
        z = np.zeros(3)
        att = np.asarray(to_float(current_state.attitude), float)
        if att.shape[0] == 4:                                # quaternion attitude (controller.py:785-803): normalised (identity below 1e-6), to Euler angles once
            nq = float(np.sqrt(att @ att))
            w, x, y, zq = (att / nq) if nq > 1e-6 else np.array([1.0, 0.0, 0.0, 0.0])
            att = np.array([np.arctan2(2 * (w * x + y * zq), 1 - 2 * (x * x + y * y)), np.arcsin(np.clip(2 * (w * y - zq * x), -1, 1)),
                            np.arctan2(2 * (w * zq + x * y), 1 - 2 * (y * y + zq * zq))])
        row = lambda a: self._dev(np.asarray(to_float(a), float).reshape(1, 3))
        return (self._dev([float(current_state.timestamp)], "f64"), row(current_state.position), row(current_state.velocity), row(att),
                row(current_state.angular_velocity if current_state.angular_velocity is not None else z))

    def _bookkeep(self, out):
        fl = int(out["flags"].cpu()[0])
        thrust = float(out["thrust"].cpu()[0])
        torque = out["torque"].cpu().numpy()[0].astype(float)
        if not fl & 1:
            self.control_outputs.append([thrust, *torque])
        return thrust, torque

    # ------------------------------------------------------------------ controller.py:413-512, :706-726
    def compute_control(self, current_state: DroneState, desired_pos, desired_vel, desired_acc, desired_yaw=0.0,
                        desired_yaw_rate=0.0) -> ControlCommand:
        t, p, v, a, w = self._state_args(current_state)
        row = lambda x: self._dev(np.asarray(to_float(x), float).reshape(1, 3))
        out = self._get_ops().control(self._params(), self._members(), t, p, v, a, w, row(desired_pos), row(desired_vel), row(desired_acc),
                                      self._dev([float(to_float(desired_yaw))]), self._dev([float(to_float(desired_yaw_rate))]))
        thrust, torque = self._bookkeep(out)
        return ControlCommand(thrust=thrust, torque=torque)

    def compute_body_rate_command(self, current_state: DroneState, desired_pos, desired_vel, desired_acc, desired_yaw=0.0,
                                  desired_yaw_rate=0.0) -> BodyRateCommand:
        t, p, v, a, w = self._state_args(current_state)
        row = lambda x: self._dev(np.asarray(to_float(x), float).reshape(1, 3))
        out = self._get_ops().control(self._params(), self._members(), t, p, v, a, w, row(desired_pos), row(desired_vel), row(desired_acc),
                                      self._dev([float(to_float(desired_yaw))]), self._dev([float(to_float(desired_yaw_rate))]),
                                      want_body_rate=True)
        self._bookkeep(out)
        return BodyRateCommand(thrust=float(out["body_thrust"].cpu()[0]), body_rates=out["body_rates"].cpu().numpy()[0].astype(float))

    # ------------------------------------------------------------------ controller.py:253-411, :728-768 (the 400 Hz hardware loop's path)
    def compute_control_fast(self, pos, vel, att, ang_vel, desired_pos, desired_vel, desired_acc, desired_yaw: float = 0.0,
                             desired_yaw_rate: float = 0.0, dt: float = 0.001):
        """-> (thrust: float newtons, torque: ndarray (3,)).  Shares integral, halved gains and saturation flags with compute_control."""
        row = lambda x: self._dev(np.asarray(x, float).reshape(1, 3))
        out = self._get_ops().control_fast(self._params(), self._members(), float(dt), row(pos), row(vel), row(att), row(ang_vel), row(desired_pos),
                                           row(desired_vel), row(desired_acc), self._dev([float(desired_yaw)]), self._dev([float(desired_yaw_rate)]),
                                           vehicle_mass=self._fast_mass, vehicle_gravity=self._fast_gravity_magnitude)
        fl = int(out["flags"].cpu()[0])
        self._thrust_saturation_count += int(bool(fl & 4))               # :315, :319 (only this path counts)
        self._torque_saturation_count += bin((fl >> 5) & 7).count("1")   # :404
        return float(out["thrust"].cpu()[0]), out["torque"].cpu().numpy()[0].astype(float)

    def compute_control_from_fast_state(self, fast_state, desired_pos, desired_vel, desired_acc, desired_yaw: float = 0.0,
                                        desired_yaw_rate: float = 0.0, dt: float = 0.001):
        return self.compute_control_fast(fast_state.position, fast_state.velocity, fast_state.attitude, fast_state.angular_velocity, desired_pos,
                                         desired_vel, desired_acc, desired_yaw, desired_yaw_rate, dt)

    # ------------------------------------------------------------------ the glue of the contract test (reference: stub :873-875 / missing)
    def _plan_args(self, trajectory: Trajectory):
        P = np.asarray(to_float(trajectory.positions), float)
        V = None if trajectory.velocities is None else self._dev(np.asarray(to_float(trajectory.velocities), float))
        A = None if trajectory.accelerations is None else self._dev(np.asarray(to_float(trajectory.accelerations), float))
        return self._dev(np.asarray(trajectory.timestamps, float), "f64"), self._dev(P), V, A

    def _from_trajectory(self, current_state: DroneState, trajectory: Trajectory, t: float, body_rate: bool, target: bool = False):
        tt, p, v, a, w = self._state_args(current_state)
        ts, P, V, A = self._plan_args(trajectory)
        return self._get_ops().control_plan(self._params(), self._members(), tt, self._dev([float(t)], "f64"), p, v, a, w, ts, P, V, A,
                                            want_body_rate=body_rate, want_target=target)

    def compute_control_from_trajectory(self, current_state: DroneState, trajectory: Trajectory, t: float) -> ControlCommand:
        thrust, torque = self._bookkeep(self._from_trajectory(current_state, trajectory, t, False))
        return ControlCommand(thrust=thrust, torque=torque)

    def compute_body_rate_from_trajectory(self, current_state: DroneState, trajectory: Trajectory, t: float) -> BodyRateCommand:
        out = self._from_trajectory(current_state, trajectory, t, True)
        self._bookkeep(out)
        return BodyRateCommand(thrust=float(out["body_thrust"].cpu()[0]), body_rates=out["body_rates"].cpu().numpy()[0].astype(float))

    def sample_trajectory(self, trajectory: Trajectory, t: float):
        """The plan sampled at t as the glue sees it (onboard_controller.py:43-93) -> (position, velocity, acceleration, 0.0, 0.0).
        Evaluated on the device against a scratch copy of the controller members (the controller itself is not advanced)."""
        keep = self._members().clone()
        st = DroneState(timestamp=float(t), position=np.zeros(3), velocity=np.zeros(3))
        tg = self._from_trajectory(st, trajectory, t, False, target=True)["target"].cpu().numpy()[0].astype(float)
        self._state.copy_(keep)
        return tg[0:3], tg[3:6], tg[6:9], 0.0, 0.0

    # ------------------------------------------------------------------ the building blocks the reference's controller tests call directly
    FALLBACK_METHODS = {"skip_yaw": 0, "default_heading": 1, "maintain_current": 2}

    @property
    def _fast_inertia(self) -> np.ndarray:                   # controller.py:125; its tests assign a full matrix
        return np.diag(np.asarray(self.config.inertia, float)) if self._inertia_override is None else self._inertia_override

    @_fast_inertia.setter
    def _fast_inertia(self, m) -> None:
        self._inertia_override = np.asarray(m, float).reshape(3, 3)

    def _update_integral_error(self, vel_error, dt: float, thrust_saturated: bool = False, torque_saturated=None) -> None:   # :536-564
        sat = int(bool(thrust_saturated)) | sum((2 << i) for i in range(3) if torque_saturated is not None and bool(torque_saturated[i]))
        import torch
        ops = self._get_ops()
        ops.controller_integral_update(self._params(), self._members(), self._dev(np.asarray(vel_error, float).reshape(1, 3)), float(dt),
                                       torch.tensor([sat], dtype=torch.int32, device=ops.be.device))

    def _attitude_torque(self, att, ang_vel, b3_des, yaw_des, yaw_rate_des, inertia):
        row = lambda x: self._dev(np.asarray(to_float(x), float).reshape(1, 3))
        out = self._get_ops().controller_attitude_torque(self._params(), self._members(), row(att), row(ang_vel), row(b3_des), self._dev([float(yaw_des)]),
                                                         self._dev([float(yaw_rate_des)]), inertia=inertia)
        return out["torque"].cpu().numpy()[0].astype(float), int(out["flags"].cpu()[0])

    def _fast_geometric_attitude_control(self, att, ang_vel, b3_des, yaw_des: float, yaw_rate_des: float) -> np.ndarray:   # :348-411
        torque, fl = self._attitude_torque(att, ang_vel, b3_des, yaw_des, yaw_rate_des, self._inertia_override)
        self._torque_saturation_count += bin((fl >> 5) & 7).count("1")     # :404
        return torque

    def _geometric_attitude_control(self, state: DroneState, b3_des, yaw_des: float, yaw_rate_des: float, thrust_mag: float, dt: float):   # :643-704
        torque, _ = self._attitude_torque(state.attitude, state.angular_velocity, b3_des, yaw_des, yaw_rate_des, None)
        return thrust_mag, torque

    def _frame(self, yaw_vector, b3_des, current_yaw, method: int):
        row = lambda x: self._dev(np.asarray(x, float).reshape(1, 3))
        out = self._get_ops().controller_desired_frame(self._params(), row(yaw_vector), row(b3_des), self._dev([float(current_yaw)]), method)
        return out["frame"].cpu().numpy()[0].astype(float), float(out["cos_angle"].cpu()[0]), bool(int(out["singular"].cpu()[0]))

    def _detect_yaw_singularity(self, yaw_vector, b3_des):   # :160-189
        _, cos_angle, singular = self._frame(yaw_vector, b3_des, 0.0, -1)
        return singular, cos_angle, self.config.yaw_singularity_fallback_method

    def _handle_yaw_singularity(self, yaw_vector, b3_des, current_yaw: float, fallback_method: str):   # :191-252
        f, _, _ = self._frame(yaw_vector, b3_des, current_yaw, self.FALLBACK_METHODS.get(fallback_method, 3))
        return f[0:3], f[3:6], f[6:9]

    # ------------------------------------------------------------------ members of the reference class, kept in the device record
    def _word(self, i):
        return float(self._members().cpu()[0, i])

    def _set_words(self, lo: int, values) -> None:
        import torch
        st = self._members()
        st[0, lo:lo + len(values)] = torch.as_tensor(np.asarray(values, float), dtype=st.dtype, device=st.device)

    def _set_flag_bits(self, mask: int, bits: int) -> None:
        self._set_words(11, [float((int(self._word(11)) & ~mask) | bits)])

    @property
    def integral_vel_error(self) -> np.ndarray:
        return self._members().cpu().numpy()[0, 0:3].astype(float)

    @integral_vel_error.setter
    def integral_vel_error(self, v) -> None:
        self._set_words(0, np.asarray(v, float).reshape(3))

    @property
    def last_time(self):
        v = self._word(3)
        return None if v != v else v

    @property
    def last_valid_thrust(self) -> float:
        return self._word(4)

    @property
    def failsafe_active(self) -> bool:
        return bool(int(self._word(11)) & 1)

    @property
    def failsafe_count(self) -> int:
        return int(self._word(9))

    @property
    def gain_scale(self) -> float:
        """0.5 ** (number of failsafe activations): the reference halves kp_pos, kd_pos, kp_att, kd_att in place on every new
        activation (controller.py:817-821); here ``config`` keeps the profile's values and this factor is applied on the device."""
        return 0.5 ** int(self._word(10))

    @property
    def last_thrust_saturated(self) -> bool:
        return bool(int(self._word(11)) & 2)

    @last_thrust_saturated.setter
    def last_thrust_saturated(self, v) -> None:
        self._set_flag_bits(2, 2 if v else 0)

    @property
    def last_torque_saturated(self) -> np.ndarray:
        f = int(self._word(11))
        return np.array([bool(f & 4), bool(f & 8), bool(f & 16)])

    @last_torque_saturated.setter
    def last_torque_saturated(self, v) -> None:
        self._set_flag_bits(4 | 8 | 16, sum((4 << i) for i in range(3) if bool(v[i])))

    @property
    def unsaturated_thrust(self) -> float:
        return self._word(5)

    @unsaturated_thrust.setter
    def unsaturated_thrust(self, v) -> None:
        self._set_words(5, [float(v)])

    @property
    def unsaturated_torque(self) -> np.ndarray:
        return self._members().cpu().numpy()[0, 6:9].astype(float)

    @unsaturated_torque.setter
    def unsaturated_torque(self, v) -> None:
        self._set_words(6, np.asarray(v, float).reshape(3))

    def reset(self) -> None:                                 # controller.py:853-869
        if self._state is not None:
            # the reference's reset() does not undo the gain halving its failsafe did IN PLACE on self.config (controller.py:817-821): the
            # count of halvings (word 10 of the record, from which the kernels derive the gains) survives a reset
            halvings = self._word(10)
            self._get_ops().lib.controller_reset(self._params(), 1, self._state.data_ptr(), self._get_ops().be.stream())
            self._set_words(10, [halvings])
        self.position_errors, self.velocity_errors, self.control_outputs = [], [], []
        self._thrust_saturation_count = self._torque_saturation_count = 0

    def get_performance_metrics(self) -> dict:               # controller.py:821-851 (without the error logs the device does not keep)
        return {"anti_windup_method": self.config.anti_windup_method, "integral_magnitude": float(np.linalg.norm(self.integral_vel_error)),
                "integral_per_axis": self.integral_vel_error.tolist(), "thrust_saturation_count": self._thrust_saturation_count,
                "torque_saturation_count": self._torque_saturation_count, "failsafe_activations": self.failsafe_count,
                "yaw_singularity_threshold": self.config.yaw_singularity_threshold,
                "yaw_singularity_fallback_method": self.config.yaw_singularity_fallback_method,
                "yaw_singularity_warning_threshold": self.config.yaw_singularity_warning_threshold}


assert CONTROLLER_STATE_WORDS == 12
