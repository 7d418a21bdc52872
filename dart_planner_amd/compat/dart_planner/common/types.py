"""Import-name shim: `dart_planner.common.types` -> `dart_planner_amd.common.types` (put dart_planner_amd/compat on
PYTHONPATH to run code written against the reference's package name)."""
from dart_planner_amd.common.types import *  # noqa: F401,F403
from dart_planner_amd.common.types import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
