"""Exception names the planner boundary uses (reference: src/dart_planner/common/errors.py)."""


class DARTPlannerError(Exception):
    pass


class PlanningError(DARTPlannerError):
    pass


class ConfigurationError(DARTPlannerError):
    pass
