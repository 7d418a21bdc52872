"""Unit handling at the planner boundary.

The reference tags quantities with ``pint`` (src/dart_planner/common/units.py:17-110) and then feeds
them to bare-float NumPy/SciPy arithmetic; the only executable reading of that is "every quantity
is an SI magnitude" (SURVEY.md section 0-6).  This module implements exactly that reading without
depending on pint: plain numbers/arrays pass through, objects that look like a pint Quantity
(``.to(unit).magnitude``) are converted to the expected SI unit and stripped.
"""
from typing import Any, Optional

import numpy as np


def to_float(q: Any):
    """reference units.py:67-75: strip units if present."""
    if hasattr(q, "magnitude"):
        return q.magnitude
    return q


def ensure_units(value: Any, expected_unit: str, context: str = "") -> np.ndarray:
    """reference units.py:78-110, returning the SI magnitude instead of a Quantity."""
    if hasattr(value, "to") and hasattr(value, "magnitude"):
        try:
            return np.asarray(value.to(expected_unit).magnitude, dtype=float)
        except Exception as e:   # pint.DimensionalityError without importing pint
            raise ValueError(f"Unit mismatch in {context}: cannot convert {value!r} to {expected_unit}") from e
    if isinstance(value, (int, float, np.integer, np.floating, np.ndarray, list, tuple)):
        return np.asarray(value, dtype=float)
    raise ValueError(f"Expected Quantity or number, got {type(value)} in {context}")


def Q_(value, unit: Optional[str] = None) -> np.ndarray:
    """reference units.py:44-64 under identity units: the magnitude itself."""
    if unit is None and not isinstance(value, str):
        raise ValueError("Must provide unit when value is not a string")
    return np.asarray(value, dtype=float)
