"""Import-name shim: `dart_planner.utils.drone_simulator` -> `dart_planner_amd.utils.drone_simulator` (put dart_planner_amd/compat on
PYTHONPATH to run code written against the reference's package name)."""
from dart_planner_amd.utils.drone_simulator import *  # noqa: F401,F403
from dart_planner_amd.utils.drone_simulator import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
