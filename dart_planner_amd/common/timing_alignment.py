"""Planner time step source.  The reference's SE3MPCPlanner constructor overwrites the caller's dt
with get_timing_manager().get_planner_dt() == 1 / control_frequency == 1/400 s
(planner.py:99-105; common/timing_alignment.py:76-78, :284-305; config/frozen_config.py:86)."""
import contextvars
import threading
from dataclasses import dataclass
from typing import Optional


@dataclass
class TimingConfig:
    control_frequency: float = 400.0      # frozen_config.py:86 control_loop_frequency_hz
    planning_frequency: float = 25.0


class TimingManager:
    def __init__(self, config: TimingConfig):
        self.config = config
        self.control_dt = 1.0 / config.control_frequency
        self.planning_dt = 1.0 / config.planning_frequency

    def get_planner_dt(self) -> float:
        return self.control_dt


_ctx: contextvars.ContextVar = contextvars.ContextVar("_se3mpc_timing_manager", default=None)
_lock = threading.Lock()


def get_timing_manager(config: Optional[TimingConfig] = None) -> TimingManager:
    mgr = _ctx.get()
    if mgr is None:
        with _lock:
            mgr = _ctx.get()
            if mgr is None:
                mgr = TimingManager(config or TimingConfig())
                _ctx.set(mgr)
    return mgr


def reset_timing_manager() -> None:
    _ctx.set(None)
