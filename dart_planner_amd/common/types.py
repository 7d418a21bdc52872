"""Boundary dataclasses of the Planner->Controller contract.

Field names, order and defaults follow the reference (src/dart_planner/common/types.py:30-140) so callers can
construct and read them the same way; values are SI magnitudes held as float64 ndarrays (common/units.py)
instead of pint Quantities."""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .units import ensure_units


def _vec3():
    return field(default_factory=lambda: np.zeros(3))


def _to_si(obj, table) -> None:
    """Normalise the listed attributes of a dataclass instance to SI magnitudes (accepts numbers, arrays and
    pint-like quantities)."""
    owner = type(obj).__name__
    for name, unit in table:
        setattr(obj, name, ensure_units(getattr(obj, name), unit, f"{owner}.{name}"))


_STATE_UNITS = (("position", "m"), ("velocity", "m/s"), ("attitude", "rad"), ("angular_velocity", "rad/s"))


@dataclass
class FastDroneState:
    """Unit-free state for high-rate loops (reference :30-56)."""
    timestamp: float
    position: np.ndarray = _vec3()
    velocity: np.ndarray = _vec3()
    attitude: np.ndarray = _vec3()
    angular_velocity: np.ndarray = _vec3()

    @classmethod
    def from_drone_state(cls, state: "DroneState") -> "FastDroneState":
        return cls(state.timestamp, *(np.array(getattr(state, n), dtype=float) for n, _ in _STATE_UNITS))


@dataclass
class DroneState:
    """Full vehicle state at one instant (reference :63-101)."""
    timestamp: float
    position: np.ndarray = _vec3()
    velocity: np.ndarray = _vec3()
    attitude: np.ndarray = _vec3()            # roll, pitch, yaw
    angular_velocity: np.ndarray = _vec3()
    motor_rpms: Optional[np.ndarray] = field(default_factory=lambda: np.zeros(4))

    def __post_init__(self):
        _to_si(self, _STATE_UNITS)

    def to_fast_state(self) -> FastDroneState:
        return FastDroneState.from_drone_state(self)


@dataclass
class ControlCommand:
    """Collective thrust [N] and body torque [N m] (reference :103-113)."""
    thrust: float = 0.0
    torque: np.ndarray = _vec3()

    def __post_init__(self):
        _to_si(self, (("thrust", "N"), ("torque", "N*m")))
        self.thrust = float(self.thrust)


@dataclass
class BodyRateCommand:
    """Normalised thrust in [0, 1] and body rates [rad/s] (reference :115-125)."""
    thrust: float
    body_rates: np.ndarray = _vec3()

    def __post_init__(self):
        _to_si(self, (("body_rates", "rad/s"),))


@dataclass
class Trajectory:
    """Time-indexed plan handed from planner to controller (reference :127-140)."""
    timestamps: np.ndarray
    positions: np.ndarray
    velocities: Optional[np.ndarray] = None
    accelerations: Optional[np.ndarray] = None
    attitudes: Optional[np.ndarray] = None     # roll, pitch, yaw
    body_rates: Optional[np.ndarray] = None
    thrusts: Optional[np.ndarray] = None       # |thrust vector| per step
    yaws: Optional[np.ndarray] = None
    yaw_rates: Optional[np.ndarray] = None
