"""ctypes binding of libse3mpc.so (include/se3mpc.h) -- raw pointers in, status codes out.

This is the only place the package touches the C ABI.  There is NO fallback: if the HIP library
has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C dart_planner_amd/csrc``) loading raises :class:`Se3mpcLibraryError`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "libse3mpc.so")

SE3MPC_MAX_HORIZON = 64
SE3MPC_MAX_CORRECTIONS = 10
SE3MPC_MAX_SPHERES = 256

STATUS_NAMES = {0: "SE3MPC_OK", -1: "SE3MPC_ERR_NULL", -2: "SE3MPC_ERR_HORIZON", -3: "SE3MPC_ERR_SHAPE",
                -4: "SE3MPC_ERR_PARAM", -5: "SE3MPC_ERR_WORKSPACE", -6: "SE3MPC_ERR_LAUNCH",
                -7: "SE3MPC_ERR_NO_DEVICE"}
TASK_MESSAGES = {1: "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL",
                 2: "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH",
                 3: "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT",
                 4: "STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT",
                 5: "ABNORMAL: "}


class Se3mpcLibraryError(RuntimeError):
    """libse3mpc.so is missing or does not export the ABI of include/se3mpc.h."""


class Se3mpcError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, fn: str, status: int, detail: str = ""):
        self.status = status
        super().__init__(f"{fn} -> {STATUS_NAMES.get(status, status)}" + (f": {detail}" if detail else ""))


class Params(C.Structure):
    """struct se3mpc_params (include/se3mpc.h) == SE3MPCConfig + ctor constants of the reference
    (src/dart_planner/planning/se3_mpc_planner.py:36-79, :149-151), unit-stripped."""
    _fields_ = [("horizon", C.c_int32), ("has_goal", C.c_int32), ("dt", C.c_double), ("mass", C.c_double),
                ("gravity", C.c_double), ("position_weight", C.c_double), ("velocity_weight", C.c_double),
                ("acceleration_weight", C.c_double), ("thrust_weight", C.c_double), ("terminal_factor", C.c_double),
                ("position_bound", C.c_double), ("max_velocity", C.c_double), ("max_acceleration", C.c_double),
                ("max_thrust", C.c_double), ("min_thrust", C.c_double), ("max_tilt_angle", C.c_double),
                ("safety_margin", C.c_double), ("max_iterations", C.c_int32), ("max_corrections", C.c_int32),
                ("max_linesearch", C.c_int32), ("max_fun", C.c_int32), ("pgtol", C.c_double), ("ftol", C.c_double)]

    @classmethod
    def reference_defaults(cls, **overrides) -> "Params":
        """The reference defaults, computed here the same way se3mpc_default_params() does (the
        test-suite checks the two agree)."""
        p = cls(horizon=6, has_goal=1, dt=1.0 / 400.0, mass=1.5, gravity=9.81, position_weight=100.0,
                velocity_weight=10.0, acceleration_weight=1.0, thrust_weight=0.1, terminal_factor=10.0,
                position_bound=100.0, max_velocity=10.0, max_acceleration=15.0, max_thrust=25.0, min_thrust=2.0,
                max_tilt_angle=math.pi / 4, safety_margin=1.5, max_iterations=15, max_corrections=10,
                max_linesearch=20, max_fun=15000, pgtol=0.05, ftol=0.5)
        names = cls._field_names()
        for k, v in overrides.items():
            if k not in names:
                raise AttributeError(f"se3mpc_params has no field {k!r}")
            setattr(p, k, v)
        return p

    @classmethod
    def _field_names(cls) -> frozenset:
        names = cls.__dict__.get("_names")
        if names is None:
            names = cls._names = frozenset(n for n, _ in cls._fields_)
        return names

    def copy(self, **overrides) -> "Params":
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        for k, v in overrides.items():
            setattr(q, k, v)
        return q

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class SolveInfo(C.Structure):
    """struct se3mpc_solve_info."""
    _fields_ = [("fun", C.c_double), ("nit", C.c_int32), ("nfev", C.c_int32), ("status", C.c_int32),
                ("task", C.c_int32)]


class VoxelMapDesc(C.Structure):
    """struct se3mpc_voxel_map: the caller-owned hash table of the device voxel map (device addresses)."""
    _fields_ = [("keys", C.c_void_p), ("prob", C.c_void_p), ("count", C.c_void_p), ("capacity", C.c_int32),
                ("reserved", C.c_int32), ("resolution", C.c_double), ("prior", C.c_double)]


class ControllerParams(C.Structure):
    """struct se3mpc_controller_params == GeometricControllerConfig after its tuning profile
    (src/dart_planner/control/geometric_controller.py:26-77, :140-158), unit-stripped."""
    _fields_ = [("kp_pos", C.c_double * 3), ("ki_pos", C.c_double * 3), ("kd_pos", C.c_double * 3), ("kp_att", C.c_double * 3),
                ("kd_att", C.c_double * 3), ("inertia", C.c_double * 3), ("max_torque_xyz", C.c_double * 3),
                ("max_integral_per_axis", C.c_double * 3), ("max_integral_pos", C.c_double), ("max_tilt_angle", C.c_double),
                ("mass", C.c_double), ("gravity", C.c_double), ("max_thrust", C.c_double), ("min_thrust", C.c_double),
                ("tracking_error_threshold", C.c_double), ("velocity_error_threshold", C.c_double),
                ("back_calculation_gain", C.c_double), ("integral_decay_factor", C.c_double), ("saturation_threshold", C.c_double),
                ("yaw_singularity_threshold", C.c_double), ("default_heading_yaw", C.c_double),
                ("anti_windup_method", C.c_int32), ("yaw_fallback_method", C.c_int32)]
    ANTI_WINDUP = {"clamping": 0, "back_calculation": 1}
    YAW_FALLBACK = {"skip_yaw": 0, "default_heading": 1, "maintain_current": 2}

    @classmethod
    def from_config(cls, cfg) -> "ControllerParams":
        """From any object with GeometricControllerConfig's attribute names (the mirror's dataclass, the oracle's)."""
        p = cls()
        for name, ctype in cls._fields_:
            if name == "anti_windup_method":
                p.anti_windup_method = cls.ANTI_WINDUP.get(cfg.anti_windup_method, 2)
            elif name == "yaw_fallback_method":
                p.yaw_fallback_method = cls.YAW_FALLBACK.get(cfg.yaw_singularity_fallback_method, 3)
            elif ctype is C.c_double:
                setattr(p, name, float(getattr(cfg, name)))
            else:
                setattr(p, name, (C.c_double * 3)(*[float(v) for v in getattr(cfg, name)]))
        return p


class SimulatorParams(C.Structure):
    """struct se3mpc_simulator_params == DroneSimulator.__init__ (src/dart_planner/utils/drone_simulator.py:41-50)."""
    _fields_ = [("mass", C.c_double), ("gravity", C.c_double), ("inertia", C.c_double * 3), ("max_thrust", C.c_double),
                ("max_torque", C.c_double)]

    @classmethod
    def reference_defaults(cls, **overrides) -> "SimulatorParams":
        p = cls(mass=1.5, gravity=9.81, inertia=(C.c_double * 3)(0.1, 0.1, 0.2), max_thrust=20.0, max_torque=10.0)
        for k, v in overrides.items():
            setattr(p, k, v)
        return p


CONTROLLER_STATE_WORDS = 12

_P = C.c_void_p
_I = C.c_int
_D = C.c_double
_PP = C.POINTER(Params)
_VP = C.POINTER(VoxelMapDesc)

# name -> argtypes after the leading (const se3mpc_params*) when `params` is True
_TYPED_API = {
    "init": (True, [_I, _I, _P, _P, _P, _I, _P, _P]),
    "cost_grad": (True, [_I, _I, _P, _P, _P, _P, _P]),
    "dynamics_residual": (True, [_I, _I, _P, _P, _P, _P, _P]),
    "obstacle_residual": (True, [_I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "physical_constraints": (True, [_I, _I, _P, _P, _P]),
    "extract": (True, [_I, _I, _P, _P, _P, _P, _P, _P]),
    "rollout_cost_grad": (True, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint32, _P]),
    "rollout_cost_grad_batched": (True, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.c_uint32, _P]),
    "rollout_obstacles": (True, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, C.c_uint32, _P]),
    "rollout_obstacles_batched": (True, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, C.c_uint32, _P]),
    "rollout_iterate": (True, [_I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint32, _P]),
    "rollout_iterate_obstacles": (True, [_I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _D, _P, _P, C.c_uint32, _P]),
    "shooting_finish": (True, [_I, _I, _P, _P, _I, C.c_uint32, _P, _P, _I, _D, _P, _P, _P]),
    "projected_step": (True, [_I, _I, _D, _P, _P, _P, _P]),
    "is_plan_valid": (True, [_I, _I, _P, _P, _P, _P]),
    "argmin": (False, [_I, _P, C.c_uint32, _P, _P]),
    "spheres_from_grid": (False, [_P, _P, _I, C.c_double, _I, C.c_double, _P, _I, _P, _P]),
    "transpose": (False, [_I, _I, _P, _I, _P, _I, _P]),
    "population_sums": (False, [_I, _I, _I, _P, _P, _D, _P, _D, _P, _P, _P]),
    "solve": (True, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "plan_host": (True, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_uint64, C.c_double, _P]),
}
# voxel map: se3mpc_voxel_<base>_<suffix>(const se3mpc_voxel_map*, ...)
_VOXEL_TYPED_API = {
    "query": [_P, _I, _P, _P],
    "trajectory_safe": [_P, _I, _I, C.c_longlong, _D, _D, _P, _P, _P],
    "local_spheres": [C.POINTER(C.c_double * 3), _D, _D, _I, _D, _P, _I, _P, _P, _P],
}
_VOXEL_PLAIN_API = {
    "se3mpc_voxel_clear": (C.c_int, [_VP, _P]),
    "se3mpc_voxel_insert": (C.c_int, [_VP, _P, _P, _D, _P, _I, _P, _P]),
    "se3mpc_voxel_update_rays": (C.c_int, [_VP, _P, _P, _P, _P, _I, _D, _D, _P, _P, _I, _P, _P, _P, _P]),
    "se3mpc_voxel_update_row_words": (C.c_longlong, [_I, _I]),
    "se3mpc_voxel_trace_rays": (C.c_int, [_VP, _P, _P, _P, _I, _P, _P, _I, _P, _P]),
    "se3mpc_voxel_export": (C.c_int, [_VP, _P, _P, _P, _P, _P]),
    "se3mpc_voxel_local_workspace": (C.c_int, [_I]),
}
_CP = C.POINTER(ControllerParams)
_SP = C.POINTER(SimulatorParams)
_LL = C.c_longlong
# consumer side of the contract: se3mpc_<base>_<suffix>(...)
_LOOP_TYPED_API = {
    "control": [_CP, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "control_fast": [_CP, _D, _D, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "controller_integral_update": [_CP, _I, _P, _D, _P, _P, _P],
    "controller_attitude_torque": [_CP, _I, _P, _P, _P, _P, _P, C.POINTER(C.c_double * 9), _P, _P, _P, _P],
    "controller_desired_frame": [_CP, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "control_plan": [_CP, _I, _P, _P, _P, _P, _P, _P, _I, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _P],
    "simulator_step": [_SP, _I, _D, _P, _P, _P, _LL, _P, _P, _P, _P, _P, _P],
    "closed_loop": [_CP, _SP, _I, _I, _D, _I, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _LL, _I,
                    C.POINTER(C.c_double * 3), _I, _P, _P, _P, _P, _P],
    "monte_carlo": [_PP, _CP, _SP, _I, _I, _I, _D, _P, _P, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
}
_PLAIN_API = {
    "se3mpc_controller_default_params": (C.c_int, [_CP]),
    "se3mpc_simulator_default_params": (C.c_int, [_SP]),
    "se3mpc_controller_reset": (C.c_int, [_CP, _I, _P, _P]),
    "se3mpc_abi_version": (C.c_int, []),
    "se3mpc_last_error": (C.c_char_p, []),
    "se3mpc_device_count": (C.c_int, []),
    "se3mpc_default_params": (C.c_int, [_PP]),
    "se3mpc_check_params": (C.c_int, [_PP]),
    "se3mpc_set_rollout_variant": (C.c_int, [_I]),
    "se3mpc_set_solver_variant": (C.c_int, [_I]),
    "se3mpc_reduce_keys": (C.c_int, [_P, _I, _I, _P, _P]),
    "se3mpc_key_index": (C.c_uint32, [C.c_uint64]),
    "se3mpc_key_cost": (C.c_float, [C.c_uint64]),
    "se3mpc_population_workspace": (C.c_int, [_I, _I]),
}


def exported_symbols() -> list:
    """Every symbol include/se3mpc.h declares (the CPU test-suite checks the .so exports them)."""
    names = list(_PLAIN_API)
    for base in _TYPED_API:
        names += [f"se3mpc_{base}_f32", f"se3mpc_{base}_f64"]
    for base in _LOOP_TYPED_API:
        names += [f"se3mpc_{base}_f32", f"se3mpc_{base}_f64"]
    names += list(_VOXEL_PLAIN_API)
    for base in _VOXEL_TYPED_API:
        names += [f"se3mpc_voxel_{base}_f32", f"se3mpc_voxel_{base}_f64"]
    return names


class Library:
    """A loaded libse3mpc.so.  Methods take raw device addresses (ints) and return nothing;
    a negative status raises :class:`Se3mpcError`."""

    def __init__(self, path: Optional[str] = None):
        self.path = path or os.environ.get("SE3MPC_LIBRARY", DEFAULT_LIBRARY)
        if not os.path.exists(self.path):
            raise Se3mpcLibraryError(
                f"{self.path} not found: the HIP extension has not been built.  Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C dart_planner_amd/csrc`). "
                "There is no CPU fallback.")
        try:
            self._dll = C.CDLL(self.path)
        except OSError as e:  # pragma: no cover
            raise Se3mpcLibraryError(f"cannot load {self.path}: {e}") from e
        missing = [s for s in exported_symbols() if not hasattr(self._dll, s)]
        if missing:
            raise Se3mpcLibraryError(f"{self.path} does not export {missing}")
        for name, (res, args) in _PLAIN_API.items():
            fn = getattr(self._dll, name)
            fn.restype, fn.argtypes = res, args
        for base, (has_params, args) in _TYPED_API.items():
            for suf in ("f32", "f64"):
                fn = getattr(self._dll, f"se3mpc_{base}_{suf}")
                fn.restype = C.c_int
                fn.argtypes = ([_PP] if has_params else []) + args
        for base, args in _LOOP_TYPED_API.items():
            for suf in ("f32", "f64"):
                fn = getattr(self._dll, f"se3mpc_{base}_{suf}")
                fn.restype, fn.argtypes = C.c_int, args
        for name, (res, args) in _VOXEL_PLAIN_API.items():
            fn = getattr(self._dll, name)
            fn.restype, fn.argtypes = res, args
        for base, args in _VOXEL_TYPED_API.items():
            for suf in ("f32", "f64"):
                fn = getattr(self._dll, f"se3mpc_voxel_{base}_{suf}")
                fn.restype = C.c_int
                fn.argtypes = [_VP] + args
        if self._dll.se3mpc_abi_version() != 1:
            raise Se3mpcLibraryError(f"{self.path}: ABI version {self._dll.se3mpc_abi_version()} != 1")

    # -- plain entry points -----------------------------------------------------------------
    def abi_version(self) -> int:
        return self._dll.se3mpc_abi_version()

    def last_error(self) -> str:
        return (self._dll.se3mpc_last_error() or b"").decode()

    def device_count(self) -> int:
        return self._dll.se3mpc_device_count()

    def default_params(self) -> Params:
        p = Params()
        self._check("se3mpc_default_params", self._dll.se3mpc_default_params(C.byref(p)))
        return p

    def check_params(self, p: Params) -> int:
        return self._dll.se3mpc_check_params(C.byref(p))

    def set_rollout_variant(self, variant: int) -> None:
        self._check("se3mpc_set_rollout_variant", self._dll.se3mpc_set_rollout_variant(variant))

    def set_solver_variant(self, variant: int) -> None:
        self._check("se3mpc_set_solver_variant", self._dll.se3mpc_set_solver_variant(variant))

    def reduce_keys(self, wave_keys: int, per_batch: int, nbatch: int, keys_out: int, stream: int) -> None:
        self._check("se3mpc_reduce_keys", self._dll.se3mpc_reduce_keys(wave_keys, per_batch, nbatch, keys_out, stream))

    def population_workspace(self, rows: int, B: int) -> int:
        return self._dll.se3mpc_population_workspace(rows, B)

    def key_index(self, key: int) -> int:
        return self._dll.se3mpc_key_index(C.c_uint64(key))

    def key_cost(self, key: int) -> float:
        return self._dll.se3mpc_key_cost(C.c_uint64(key))

    # -- typed entry points -----------------------------------------------------------------
    def call(self, base: str, suffix: str, *args, params: Optional[Params] = None) -> None:
        """Call se3mpc_<base>_<suffix>(params?, *args); raise on a negative status."""
        fn = getattr(self._dll, f"se3mpc_{base}_{suffix}")
        has_params = _TYPED_API[base][0]
        if has_params:
            if params is None:
                raise TypeError(f"se3mpc_{base} needs params")
            rc = fn(C.byref(params), *args)
        else:
            rc = fn(*args)
        self._check(f"se3mpc_{base}_{suffix}", rc)

    def call_status(self, base: str, suffix: str, *args, params: Optional[Params] = None) -> int:
        """Same as :meth:`call` but returns the raw status (used by the error-behaviour tests)."""
        fn = getattr(self._dll, f"se3mpc_{base}_{suffix}")
        if _TYPED_API[base][0]:
            return fn(C.byref(params) if params is not None else None, *args)
        return fn(*args)

    # -- consumer side of the contract --------------------------------------------------------
    def controller_default_params(self) -> ControllerParams:
        p = ControllerParams()
        self._check("se3mpc_controller_default_params", self._dll.se3mpc_controller_default_params(C.byref(p)))
        return p

    def simulator_default_params(self) -> SimulatorParams:
        p = SimulatorParams()
        self._check("se3mpc_simulator_default_params", self._dll.se3mpc_simulator_default_params(C.byref(p)))
        return p

    def controller_reset(self, cp: ControllerParams, B: int, state: int, stream: int) -> None:
        self._check("se3mpc_controller_reset", self._dll.se3mpc_controller_reset(C.byref(cp), B, state, stream))

    def loop_call(self, base: str, suffix: str, *args) -> None:
        """se3mpc_control_<suffix> / se3mpc_closed_loop_<suffix>; struct arguments are passed by reference here."""
        a = [C.byref(x) if isinstance(x, (ControllerParams, SimulatorParams, Params)) else x for x in args]
        self._check(f"se3mpc_{base}_{suffix}", getattr(self._dll, f"se3mpc_{base}_{suffix}")(*a))

    def loop_status(self, base: str, suffix: str, *args) -> int:
        a = [C.byref(x) if isinstance(x, (ControllerParams, SimulatorParams, Params)) else x for x in args]
        return getattr(self._dll, f"se3mpc_{base}_{suffix}")(*a)

    # -- voxel map ----------------------------------------------------------------------------
    def voxel(self, name: str, desc: VoxelMapDesc, *args) -> None:
        """Call se3mpc_voxel_<name>(&desc, *args) (name = 'clear', 'insert', 'export', 'query_f32', ...)."""
        self._check(f"se3mpc_voxel_{name}", getattr(self._dll, f"se3mpc_voxel_{name}")(C.byref(desc), *args))

    def voxel_status(self, name: str, desc: Optional[VoxelMapDesc], *args) -> int:
        return getattr(self._dll, f"se3mpc_voxel_{name}")(C.byref(desc) if desc is not None else None, *args)

    def voxel_local_workspace(self, cells_per_axis: int) -> int:
        return self._dll.se3mpc_voxel_local_workspace(cells_per_axis)

    def voxel_update_row_words(self, M: int, max_len: int) -> int:
        return self._dll.se3mpc_voxel_update_row_words(M, max_len)

    def _check(self, name: str, rc: int) -> None:
        if rc != 0:
            raise Se3mpcError(name, rc, self.last_error() if rc == -6 else "")


_default: Optional[Library] = None


def get_library() -> Library:
    """The process-wide library handle (loads on first use; raises if it is not built)."""
    global _default
    if _default is None:
        _default = Library()
    return _default
