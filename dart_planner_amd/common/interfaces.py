"""IPlanner: the duck-typed planner interface (reference: src/dart_planner/common/interfaces.py:81-105)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

from .types import DroneState, Trajectory


class IPlanner(ABC):
    @abstractmethod
    def plan_trajectory(self, current_state: DroneState, goal) -> Optional[Trajectory]: ...

    @abstractmethod
    def update_plan(self, current_state: DroneState, obstacles: List[Dict[str, Any]]) -> Trajectory: ...

    @abstractmethod
    def is_plan_valid(self, trajectory: Trajectory) -> bool: ...

    @abstractmethod
    def get_planning_stats(self) -> Dict[str, Any]: ...
