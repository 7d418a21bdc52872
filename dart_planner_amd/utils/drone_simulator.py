"""DroneSimulator -- mirror of the reference's closed-loop test harness (src/dart_planner/utils/drone_simulator.py:41-72):
point-mass vertical thrust + Euler-angle integration, actuator saturation and constant wind, same constructor and ``step``
signature.  ``step`` runs ``se3mpc_simulator_step_*`` (dart_planner_amd/csrc/closed_loop.hip) on the device -- the same code
the batched closed loop (``Ops.closed_loop``) applies after every control call; there is no CPU fallback."""
from typing import Optional

import numpy as np

from ..capi import SimulatorParams
from ..common.types import ControlCommand, DroneState
from ..common.units import to_float


class DroneSimulator:
    def __init__(self, wind: Optional[np.ndarray] = None, max_thrust: float = 20.0, max_torque: float = 10.0, *, device=None) -> None:
        self.wind = np.zeros(3) if wind is None else np.array(wind, dtype=float)
        self.max_thrust = max_thrust
        self.max_torque = max_torque
        self.mass = 1.5                                        # simulator.py:47-49
        self.gravity = 9.81
        self.inertia = np.diag([0.1, 0.1, 0.2])
        self._device = device
        self._ops = None

    def _get_ops(self):
        if self._ops is None:
            from ..ops import Ops, TorchBackend
            self._ops = Ops(TorchBackend(self._device))      # raises without a HIP device / built library
        return self._ops

    def _params(self) -> SimulatorParams:
        d = np.diag(self.inertia)
        return SimulatorParams.reference_defaults(mass=self.mass, gravity=self.gravity, max_thrust=self.max_thrust, max_torque=self.max_torque,
                                                  inertia=(SimulatorParams._fields_[2][1])(float(d[0]), float(d[1]), float(d[2])))

    def step(self, state: DroneState, command: ControlCommand, dt: float) -> DroneState:
        import torch
        ops = self._get_ops()
        dev = ops.be.device
        f = lambda a, shape: torch.as_tensor(np.ascontiguousarray(np.asarray(to_float(a), dtype=float).reshape(shape))).to(dev)
        z = np.zeros(3)
        t = f([state.timestamp], (1,))
        pos, vel = f(state.position, (1, 3)), f(state.velocity, (1, 3))
        att = f(state.attitude if state.attitude is not None else z, (1, 3))
        om = f(state.angular_velocity if state.angular_velocity is not None else z, (1, 3))
        ops.simulator_step(self._params(), t, pos, vel, att, om, f([command.thrust], (1,)), f(command.torque, (1, 3)), float(dt),
                           wind=f(self.wind, (3,)))
        h = lambda a: a.cpu().numpy()[0].astype(float)
        return DroneState(timestamp=float(t.cpu()[0]), position=h(pos), velocity=h(vel), attitude=h(att), angular_velocity=h(om))
