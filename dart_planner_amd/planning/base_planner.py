"""Planner base class and the name -> class registry.

Public surface as in the reference (src/dart_planner/planning/base_planner.py:16-136): ``BasePlanner`` with
``config``, ``obstacles``, ``planning_stats`` (keys total_plans / successful_plans / planning_times /
last_plan_time, the last 100 times kept), ``validate_goal``, ``validate_state``, ``reset_stats``; and
``PlannerFactory.register / create / list_available`` where an unknown name raises PlanningError."""
from typing import Any, Dict, List, Tuple

import numpy as np

from ..common.errors import PlanningError
from ..common.interfaces import IPlanner
from ..common.types import DroneState

_KEEP_TIMES = 100
_MIN_GOAL_ALTITUDE = 0.5       # m  (reference :71-72)
_MAX_STATE_SPEED = 20.0        # m/s per axis (reference :87-88)


def _fresh_stats() -> Dict[str, Any]:
    return dict(total_plans=0, successful_plans=0, planning_times=[], last_plan_time=0.0)


class BasePlanner(IPlanner):
    def __init__(self, config: Dict[str, Any]):
        self.config = config
        self.obstacles: List[Tuple[np.ndarray, float]] = []
        self.planning_stats = _fresh_stats()

    # -- input sanity (reference :64-90)
    def validate_goal(self, goal) -> bool:
        g = None if goal is None else np.asarray(goal)
        return g is not None and g.shape == (3,) and bool(g[2] >= _MIN_GOAL_ALTITUDE)

    def validate_state(self, state: DroneState) -> bool:
        if state is None:
            return False
        pos, vel = np.asarray(state.position, float), np.asarray(state.velocity, float)
        return bool(np.all(np.isfinite(pos)) and np.all(np.abs(vel) <= _MAX_STATE_SPEED))

    # -- statistics (reference :92-111)
    def _update_planning_stats(self, planning_time: float, success: bool) -> None:
        st = self.planning_stats
        st["total_plans"] += 1
        st["successful_plans"] += int(bool(success))
        st["last_plan_time"] = planning_time
        times = st["planning_times"]
        times.append(planning_time)                          # reference :98-102: the last 100 are kept
        if len(times) > _KEEP_TIMES:
            del times[:-_KEEP_TIMES]

    def reset_stats(self) -> None:
        self.planning_stats = _fresh_stats()


class PlannerFactory:
    """reference :114-136"""
    _planners: Dict[str, type] = {}

    @classmethod
    def register(cls, name: str, planner_class: type) -> None:
        cls._planners[name] = planner_class

    @classmethod
    def create(cls, name: str, config):
        try:
            planner_class = cls._planners[name]
        except KeyError:
            raise PlanningError(f"Unknown planner: {name}. Available: {sorted(cls._planners)}") from None
        return planner_class(config)

    @classmethod
    def list_available(cls) -> List[str]:
        return list(cls._planners)
