"""Import-name shim: `dart_planner.perception.explicit_geometric_mapper` -> `dart_planner_amd.perception.explicit_geometric_mapper`
(put dart_planner_amd/compat on PYTHONPATH to run code written against the reference's package name)."""
from dart_planner_amd.perception.explicit_geometric_mapper import *  # noqa: F401,F403
from dart_planner_amd.perception.explicit_geometric_mapper import __dict__ as _d
globals().update({k: v for k, v in _d.items() if not k.startswith("__")})
