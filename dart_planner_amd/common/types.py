"""Boundary dataclasses of the Planner->Controller contract.

Same field names, order and defaults as the reference (src/dart_planner/common/types.py:30-140);
fields hold SI magnitudes as float64 ndarrays (see common/units.py) instead of pint Quantities.
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .units import ensure_units


@dataclass
class FastDroneState:
    """reference types.py:30-56"""
    timestamp: float
    position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    attitude: np.ndarray = field(default_factory=lambda: np.zeros(3))
    angular_velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))

    @classmethod
    def from_drone_state(cls, state: "DroneState") -> "FastDroneState":
        return cls(state.timestamp, np.array(state.position), np.array(state.velocity), np.array(state.attitude),
                   np.array(state.angular_velocity))


@dataclass
class DroneState:
    """reference types.py:63-101"""
    timestamp: float
    position: np.ndarray = field(default_factory=lambda: np.zeros(3))
    velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    attitude: np.ndarray = field(default_factory=lambda: np.zeros(3))
    angular_velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    motor_rpms: Optional[np.ndarray] = field(default_factory=lambda: np.zeros(4))

    def __post_init__(self):
        self.position = ensure_units(self.position, "m", "DroneState.position")
        self.velocity = ensure_units(self.velocity, "m/s", "DroneState.velocity")
        self.attitude = ensure_units(self.attitude, "rad", "DroneState.attitude")
        self.angular_velocity = ensure_units(self.angular_velocity, "rad/s", "DroneState.angular_velocity")

    def to_fast_state(self) -> FastDroneState:
        return FastDroneState.from_drone_state(self)


@dataclass
class ControlCommand:
    """reference types.py:103-113: thrust [N], torque [N m]."""
    thrust: float = 0.0
    torque: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def __post_init__(self):
        self.thrust = float(ensure_units(self.thrust, "N", "ControlCommand.thrust"))
        self.torque = ensure_units(self.torque, "N*m", "ControlCommand.torque")


@dataclass
class BodyRateCommand:
    """reference types.py:115-125: normalised thrust in [0, 1], body rates [rad/s]."""
    thrust: float
    body_rates: np.ndarray = field(default_factory=lambda: np.zeros(3))

    def __post_init__(self):
        self.body_rates = ensure_units(self.body_rates, "rad/s", "BodyRateCommand.body_rates")


@dataclass
class Trajectory:
    """reference types.py:127-140"""
    timestamps: np.ndarray
    positions: np.ndarray
    velocities: Optional[np.ndarray] = None
    accelerations: Optional[np.ndarray] = None
    attitudes: Optional[np.ndarray] = None
    body_rates: Optional[np.ndarray] = None
    thrusts: Optional[np.ndarray] = None
    yaws: Optional[np.ndarray] = None
    yaw_rates: Optional[np.ndarray] = None
