"""The slice of the reference's DI container the planner boundary needs
(src/dart_planner/common/di_container_v2.py:488-540, :690-697):
``get_container().create_planner_container().get_se3_planner(config=None)`` -> a process-wide
singleton when no config is given, a fresh instance otherwise; ``create_control_container()
.get_geometric_controller(tuning_profile)``.  No ZMQ / security / hardware registrations."""
import threading
from typing import Any, Callable, Dict, Optional


class DIContainerV2:
    def __init__(self):
        self._lock = threading.RLock()
        self._singletons: Dict[type, Any] = {}
        self._providers: Dict[type, Callable[[], Any]] = {}

    def register_singleton(self, cls: type, provider: Optional[Callable[[], Any]] = None) -> None:
        with self._lock:
            self._providers[cls] = provider or cls

    def resolve(self, cls: type):
        with self._lock:
            if cls not in self._singletons:
                self._singletons[cls] = self._providers.get(cls, cls)()
            return self._singletons[cls]

    def create_planner_container(self):
        return PlannerContainer(self)

    def create_control_container(self):
        return ControlContainer(self)


class PlannerContainer:
    def __init__(self, container: DIContainerV2):
        self.container = container

    def get_se3_planner(self, config=None):
        from ..planning.se3_mpc_planner import SE3MPCPlanner
        if config:
            return SE3MPCPlanner(config)
        return self.container.resolve(SE3MPCPlanner)


class ControlContainer:
    def __init__(self, container: DIContainerV2):
        self.container = container

    def get_geometric_controller(self, tuning_profile: str = "sitl_optimized"):
        from ..control.geometric_controller import GeometricController
        return GeometricController(tuning_profile=tuning_profile)


_container: Optional[DIContainerV2] = None
_container_lock = threading.Lock()


def get_container() -> DIContainerV2:
    global _container
    with _container_lock:
        if _container is None:
            _container = DIContainerV2()
        return _container


def reset_container() -> None:
    global _container
    with _container_lock:
        _container = None
