"""Closed-loop test harness counterpart of the reference's DroneSimulator
(src/dart_planner/utils/drone_simulator.py:41-72): point-mass vertical thrust + Euler-angle
integration, actuator saturation and constant wind -- same constructor and ``step`` signature."""
from typing import Optional

import numpy as np

from ..common.types import ControlCommand, DroneState
from ..common.units import to_float


class DroneSimulator:
    def __init__(self, wind: Optional[np.ndarray] = None, max_thrust: float = 20.0, max_torque: float = 10.0) -> None:
        self.wind = np.zeros(3) if wind is None else np.array(wind, dtype=float)
        self.max_thrust = max_thrust
        self.max_torque = max_torque
        self.mass = 1.5
        self.gravity = 9.81
        self.inertia = np.diag([0.1, 0.1, 0.2])

    def step(self, state: DroneState, command: ControlCommand, dt: float) -> DroneState:
        thrust = float(np.clip(to_float(command.thrust), 0.0, self.max_thrust))
        torque = np.clip(np.asarray(to_float(command.torque), float), -self.max_torque, self.max_torque)
        acc = np.array([0.0, 0.0, thrust / self.mass - self.gravity]) + self.wind / self.mass
        vel = np.asarray(state.velocity, float) + acc * dt
        pos = np.asarray(state.position, float) + vel * dt
        omega = np.asarray(state.angular_velocity, float) + np.linalg.solve(self.inertia, torque) * dt
        att = np.asarray(state.attitude, float) + omega * dt
        return DroneState(timestamp=state.timestamp + dt, position=pos, velocity=vel, attitude=att, angular_velocity=omega)
