"""BasePlanner / PlannerFactory (reference: src/dart_planner/planning/base_planner.py:16-136)."""
from typing import Any, Dict, List, Tuple

import numpy as np

from ..common.interfaces import IPlanner
from ..common.types import DroneState


class BasePlanner(IPlanner):
    def __init__(self, config: Dict[str, Any]):
        self.config = config
        self.obstacles: List[Tuple[np.ndarray, float]] = []
        self.planning_stats = {"total_plans": 0, "successful_plans": 0, "planning_times": [], "last_plan_time": 0.0}

    def validate_goal(self, goal) -> bool:
        """reference base_planner.py:64-74"""
        if goal is None or np.shape(goal) != (3,):
            return False
        return not goal[2] < 0.5

    def validate_state(self, state: DroneState) -> bool:
        """reference base_planner.py:76-90"""
        if state is None:
            return False
        if np.any(np.isnan(state.position)) or np.any(np.isinf(state.position)):
            return False
        return not np.any(np.abs(state.velocity) > 20.0)

    def _update_planning_stats(self, planning_time: float, success: bool) -> None:
        st = self.planning_stats
        st["total_plans"] += 1
        st["last_plan_time"] = planning_time
        if success:
            st["successful_plans"] += 1
        st["planning_times"].append(planning_time)
        if len(st["planning_times"]) > 100:
            st["planning_times"] = st["planning_times"][-100:]

    def reset_stats(self) -> None:
        self.planning_stats = {"total_plans": 0, "successful_plans": 0, "planning_times": [], "last_plan_time": 0.0}


class PlannerFactory:
    _planners: Dict[str, type] = {}

    @classmethod
    def register(cls, name: str, planner_class: type):
        cls._planners[name] = planner_class

    @classmethod
    def create(cls, name: str, config):
        if name not in cls._planners:
            from ..common.errors import PlanningError
            raise PlanningError(f"Unknown planner: {name}. Available: {list(cls._planners.keys())}")
        return cls._planners[name](config)

    @classmethod
    def list_available(cls):
        return list(cls._planners.keys())
