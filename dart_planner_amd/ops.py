"""Array front-end of the C ABI: shapes, allocation and pointer hand-over.

``Ops`` is written against a tiny array-backend protocol so that the *same* host logic serves
PyTorch-ROCm tensors in production (:class:`TorchBackend`: CUDA tensors only, current HIP
stream, libse3mpc.so) and plain host buffers in the CPU test-suite (``tests/emu``).  PyTorch is
plumbing here -- device memory and streams -- every number is produced by the HIP kernels.

Layouts (include/se3mpc.h): evaluation ops use the lane layout ``[row][b]`` (one trajectory per
lane), the solver uses the problem layout ``[b][row]`` (the reference's decision-vector packing,
src/dart_planner/planning/se3_mpc_planner.py:361-376).
"""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes as _C

import numpy as np

from .capi import (CONTROLLER_STATE_WORDS, ControllerParams, Library, Params, SimulatorParams, SolveInfo, get_library,
                   SE3MPC_MAX_SPHERES)

INFO_DTYPE = np.dtype([("fun", "<f8"), ("nit", "<i4"), ("nfev", "<i4"), ("status", "<i4"), ("task", "<i4")])
assert INFO_DTYPE.itemsize == 24


class TorchBackend:
    """PyTorch-ROCm tensors on a HIP device.  Refuses CPU tensors: there is no CPU path."""
    graph_capable = True          # launches go to torch's current stream: a torch.cuda.graph capture records them

    def __init__(self, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("dart_planner_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._dt = {"f32": torch.float32, "f64": torch.float64, "i32": torch.int32, "i64": torch.int64,
                    "u8": torch.uint8}

    def empty(self, shape, kind):
        return self.torch.empty(shape, dtype=self._dt[kind], device=self.device)

    def suffix(self, a) -> str:
        if a.dtype == self.torch.float32:
            return "f32"
        if a.dtype == self.torch.float64:
            return "f64"
        raise TypeError(f"se3mpc kernels compute in float32 or float64, got {a.dtype}")

    def check(self, a, name: str):
        if not self.torch.is_tensor(a) or not a.is_cuda:
            raise TypeError(f"{name}: expected a CUDA (HIP) tensor; there is no CPU path")
        if a.device != self.device:
            raise ValueError(f"{name}: tensor on {a.device}, backend on {self.device}")
        if not a.is_contiguous():
            raise ValueError(f"{name}: tensor must be contiguous")
        return a

    def ptr(self, a) -> int:
        return 0 if a is None else a.data_ptr()

    def is_host_mapped(self, a) -> bool:
        return self.torch.is_tensor(a) and not a.is_cuda and a.is_pinned() and a.is_contiguous()

    def stream(self) -> int:
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def to_host(self, a) -> np.ndarray:
        return a.detach().cpu().numpy()

    def from_host(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def elements_from(self, a) -> int:
        """How many elements of a's dtype lie between a's first element and the end of its storage (views into a larger tensor reach
        past their own numel): what a kernel that strides from a.data_ptr() may legally touch."""
        return (a.untyped_storage().nbytes() - a.storage_offset() * a.element_size()) // a.element_size()


class Ops:
    """Shape-checked calls into libse3mpc for one array backend."""

    def __init__(self, backend=None, library: Optional[Library] = None):
        self.lib = library if library is not None else get_library()
        self.be = backend if backend is not None else TorchBackend()

    # ------------------------------------------------------------------ helpers
    def _lane(self, a, rows: int, name: str):
        self.be.check(a, name)
        if a.ndim != 2 or a.shape[0] != rows:
            raise ValueError(f"{name}: expected shape ({rows}, ld), got {tuple(a.shape)}")
        return a

    def _same(self, ref, *others):
        suf = self.be.suffix(ref)
        for o in others:
            if o is not None and (self.be.suffix(o) != suf or o.shape[-1] != ref.shape[-1]):
                raise ValueError("all lane-layout operands must share dtype and leading dimension")
        return suf

    @staticmethod
    def _B(ld: int, B: Optional[int]) -> int:
        B = ld if B is None else int(B)
        if B < 0 or B > ld:
            raise ValueError(f"B={B} outside [0, ld={ld}]")
        return B

    # ------------------------------------------------------------------ lane layout
    def init(self, params: Params, p0, v0, goal, project: bool = False, B: Optional[int] = None):
        """a3 (+a4 projection): -> X0 (9N, ld)."""
        N = params.horizon
        self._lane(p0, 3, "p0"); self._lane(v0, 3, "v0")
        if params.has_goal:
            self._lane(goal, 3, "goal")
        suf = self._same(p0, v0, goal if params.has_goal else None)
        ld = p0.shape[1]
        X0 = self.be.empty((9 * N, ld), suf)
        self.lib.call("init", suf, self._B(ld, B), ld, self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), int(bool(project)), self.be.ptr(X0),
                      self.be.stream(), params=params)
        return X0

    def cost_grad(self, params: Params, X, goal, want_grad: bool = True, B: Optional[int] = None):
        """a5 + a6: -> (f (ld,), g (9N, ld) | None)."""
        N = params.horizon
        self._lane(X, 9 * N, "X")
        if params.has_goal:
            self._lane(goal, 3, "goal")
        suf = self._same(X, goal if params.has_goal else None)
        ld = X.shape[1]
        f = self.be.empty((ld,), suf)
        g = self.be.empty((9 * N, ld), suf) if want_grad else None
        self.lib.call("cost_grad", suf, self._B(ld, B), ld, self.be.ptr(X),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(f), self.be.ptr(g),
                      self.be.stream(), params=params)
        return f, g

    def dynamics_residual(self, params: Params, X, p0, v0, B: Optional[int] = None):
        """a8: -> R (6N, ld)."""
        N = params.horizon
        self._lane(X, 9 * N, "X"); self._lane(p0, 3, "p0"); self._lane(v0, 3, "v0")
        suf = self._same(X, p0, v0)
        ld = X.shape[1]
        R = self.be.empty((6 * N, ld), suf)
        self.lib.call("dynamics_residual", suf, self._B(ld, B), ld, self.be.ptr(X), self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(R), self.be.stream(), params=params)
        return R

    def obstacle_residual(self, params: Params, X, spheres, materialize: bool = True, reduce: bool = True,
                          B: Optional[int] = None):
        """a9: spheres (K, 4) = (cx, cy, cz, r) -> (C (N*K, ld) | None, cmin (ld,) | None, viol (ld,) | None)."""
        N = params.horizon
        self._lane(X, 9 * N, "X")
        self.be.check(spheres, "spheres")
        if spheres.ndim != 2 or spheres.shape[1] != 4 or spheres.shape[0] > SE3MPC_MAX_SPHERES:
            raise ValueError(f"spheres: expected (K<={SE3MPC_MAX_SPHERES}, 4), got {tuple(spheres.shape)}")
        suf = self.be.suffix(X)
        if self.be.suffix(spheres) != suf:
            raise ValueError("spheres dtype must match X")
        K, ld = spheres.shape[0], X.shape[1]
        Cm = self.be.empty((N * K, ld), suf) if materialize else None
        cmin = self.be.empty((ld,), suf) if reduce else None
        viol = self.be.empty((ld,), suf) if reduce else None
        self.lib.call("obstacle_residual", suf, self._B(ld, B), ld, self.be.ptr(X), self.be.ptr(spheres), K,
                      self.be.ptr(Cm), self.be.ptr(cmin), self.be.ptr(viol), self.be.stream(), params=params)
        return Cm, cmin, viol

    def physical_constraints(self, params: Params, X, B: Optional[int] = None):
        """a10: -> C (4N, ld)."""
        N = params.horizon
        self._lane(X, 9 * N, "X")
        suf, ld = self.be.suffix(X), X.shape[1]
        Cm = self.be.empty((4 * N, ld), suf)
        self.lib.call("physical_constraints", suf, self._B(ld, B), ld, self.be.ptr(X), self.be.ptr(Cm),
                      self.be.stream(), params=params)
        return Cm

    def extract(self, params: Params, T, B: Optional[int] = None):
        """a11 + a12 on the T block (3N, ld): -> acc, att, rates (3N, ld), thrust (N, ld)."""
        N = params.horizon
        self._lane(T, 3 * N, "T")
        suf, ld = self.be.suffix(T), T.shape[1]
        acc, att, rates = (self.be.empty((3 * N, ld), suf) for _ in range(3))
        thrust = self.be.empty((N, ld), suf)
        self.lib.call("extract", suf, self._B(ld, B), ld, self.be.ptr(T), self.be.ptr(acc), self.be.ptr(att),
                      self.be.ptr(rates), self.be.ptr(thrust), self.be.stream(), params=params)
        return acc, att, rates, thrust

    def rollout_cost_grad(self, params: Params, p0, v0, goal, T, want_grad: bool = True, want_states: bool = False,
                          B: Optional[int] = None, out=None, key=None, index_base: int = 0, wave_keys=None):
        """Shooting-form rollout + cost (+ gradient wrt T, + rolled-out states).
        -> (cost (ld,), gradT (3N, ld) | None, P (3N, ld) | None, V (3N, ld) | None).
        ``out=(cost, gradT)`` reuses preallocated outputs (the benchmark's steady state).
        ``key`` (int64 (1,)) receives the fused batch argmin: min over b of (orderable cost bits << 32 |
        index_base + b) -- the kernel writes one partial key per wavefront (``wave_keys``, int64
        (ceil(ld/64),), allocated here unless given) and se3mpc_reduce_keys folds them.  Pass
        ``wave_keys`` alone to defer the fold (bucketed use, see bench.py)."""
        N = params.horizon
        self._lane(p0, 3, "p0"); self._lane(v0, 3, "v0"); self._lane(T, 3 * N, "T")
        if params.has_goal:
            self._lane(goal, 3, "goal")
        suf = self._same(T, p0, v0, goal if params.has_goal else None)
        ld = T.shape[1]
        if out is not None:
            cost, gradT = out
        else:
            cost = self.be.empty((ld,), suf)
            gradT = self.be.empty((3 * N, ld), suf) if want_grad else None
        P = self.be.empty((3 * N, ld), suf) if want_states else None
        V = self.be.empty((3 * N, ld), suf) if want_states else None
        nB = self._B(ld, B)
        if key is not None and wave_keys is None:
            wave_keys = self.be.empty(((nB + 63) // 64,), "i64")
        self.lib.call("rollout_cost_grad", suf, nB, ld, self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(cost),
                      self.be.ptr(gradT), self.be.ptr(P), self.be.ptr(V), self.be.ptr(wave_keys), int(index_base),
                      self.be.stream(), params=params)
        if key is not None and nB > 0:
            self.lib.reduce_keys(self.be.ptr(wave_keys), (nB + 63) // 64, 1, self.be.ptr(key), self.be.stream())
        return cost, gradT, P, V

    def reduce_keys(self, wave_keys, keys_out):
        """wave_keys int64 (nbatch, per_batch) -> keys_out int64 (nbatch,)."""
        nb, per = wave_keys.shape
        self.lib.reduce_keys(self.be.ptr(wave_keys), per, nb, self.be.ptr(keys_out), self.be.stream())
        return keys_out

    def shooting_finish(self, params: Params, T, wave_keys, state, out, key_out=None, spheres=None, obstacle_weight: float = 0.0, B: Optional[int] = None,
                        index_base: int = 0):
        """The tail of a shooting-form plan in one launch (``se3mpc_shooting_finish_*``): fold `wave_keys` (int64, flat), roll the winning column of
        T (3N, ld) out from state = float64 (9,) [p0 | v0 | goal], extract, and write float64 out (19 N + 3,) = [P | V | T | acc | att | rates | thrust |
        cost | penalty | cost + penalty]; key_out: int64 (1,) or None.  state, out and key_out may be pinned host tensors (read / written in place
        over the host link); spheres: float64 (K, 4) device tensor or None."""
        N = params.horizon
        self._lane(T, 3 * N, "T")
        suf = self.be.suffix(T)
        for nm, a, n in (("state", state, 9), ("out", out, 19 * N + 3), ("key_out", key_out, 1)):
            if a is None and nm == "key_out":
                continue
            if not (self.be.is_host_mapped(a) or (self.be.check(a, nm) is a)):
                raise TypeError(f"{nm}: expected a device tensor or a pinned host tensor")
            if int(np.prod(a.shape)) < n or str(a.dtype).rsplit(".", 1)[-1] != ("int64" if nm == "key_out" else "float64"):
                raise ValueError(f"{nm}: expected at least {n} {'int64' if nm == 'key_out' else 'float64'} values")
        K = 0
        if spheres is not None:
            self.be.check(spheres, "spheres")
            if spheres.ndim != 2 or spheres.shape[1] != 4 or spheres.shape[0] > SE3MPC_MAX_SPHERES or str(spheres.dtype).rsplit(".", 1)[-1] != "float64":
                raise ValueError(f"spheres: expected float64 (K<={SE3MPC_MAX_SPHERES}, 4), got {tuple(spheres.shape)} {spheres.dtype}")
            K = spheres.shape[0]
        self.be.check(wave_keys, "wave_keys")
        ld = T.shape[1]
        self.lib.call("shooting_finish", suf, self._B(ld, B), ld, self.be.ptr(T), self.be.ptr(wave_keys), int(np.prod(wave_keys.shape)), int(index_base),
                      self.be.ptr(state), self.be.ptr(spheres if K else None), K, float(obstacle_weight), self.be.ptr(out), self.be.ptr(key_out),
                      self.be.stream(), params=params)
        return out

    def rollout_cost_grad_batched(self, params: Params, p0, v0, goal, T, cost, gradT, wave_keys=None, index_base: int = 0):
        """`nbatch` independent batches in one launch.  p0, v0, goal: (nbatch, 3, ld); T, gradT:
        (nbatch, 3N, ld); cost: (nbatch, ld); wave_keys: int64 (nbatch, ceil(ld/64)) or None (fold with
        :meth:`reduce_keys`).  Outputs are caller-allocated (steady-state use)."""
        N = params.horizon
        nb, _, ld = T.shape
        for a, shp, nm in ((p0, (nb, 3, ld), "p0"), (v0, (nb, 3, ld), "v0"), (T, (nb, 3 * N, ld), "T"),
                           (cost, (nb, ld), "cost"), (gradT, (nb, 3 * N, ld), "gradT")):
            self.be.check(a, nm)
            if tuple(a.shape) != shp:
                raise ValueError(f"{nm}: expected {shp}, got {tuple(a.shape)}")
        if params.has_goal:
            self.be.check(goal, "goal")
        suf = self.be.suffix(T)
        self.lib.call("rollout_cost_grad_batched", suf, ld, ld, nb, self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(cost),
                      self.be.ptr(gradT), self.be.ptr(wave_keys), int(index_base), self.be.stream(), params=params)
        return cost, gradT

    def rollout_obstacles(self, params: Params, p0, v0, goal, T, spheres, want_grad: bool = True, B: Optional[int] = None,
                          wave_keys=None, index_base: int = 0, out=None):
        """Rollout + cost (+ gradient) fused with the sphere-obstacle residuals of the rolled-out positions.
        -> (cost (ld,), gradT (3N, ld) | None, cmin (ld,), viol (ld,)).  ``out=(cost, gradT, cmin, viol)`` reuses
        preallocated outputs."""
        N = params.horizon
        self._lane(p0, 3, "p0"); self._lane(v0, 3, "v0"); self._lane(T, 3 * N, "T")
        if params.has_goal:
            self._lane(goal, 3, "goal")
        suf = self._same(T, p0, v0, goal if params.has_goal else None)
        self.be.check(spheres, "spheres")
        if spheres.ndim != 2 or spheres.shape[1] != 4 or spheres.shape[0] > SE3MPC_MAX_SPHERES or self.be.suffix(spheres) != suf:
            raise ValueError(f"spheres: expected (K<={SE3MPC_MAX_SPHERES}, 4) {suf}, got {tuple(spheres.shape)}")
        ld = T.shape[1]
        if out is not None:
            cost, gradT, cmin, viol = out
        else:
            cost = self.be.empty((ld,), suf)
            gradT = self.be.empty((3 * N, ld), suf) if want_grad else None
            cmin, viol = self.be.empty((ld,), suf), self.be.empty((ld,), suf)
        self.lib.call("rollout_obstacles", suf, self._B(ld, B), ld, self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(cost), self.be.ptr(gradT),
                      self.be.ptr(spheres), spheres.shape[0], self.be.ptr(cmin), self.be.ptr(viol), self.be.ptr(wave_keys),
                      int(index_base), self.be.stream(), params=params)
        return cost, gradT, cmin, viol

    def rollout_obstacles_batched(self, params: Params, p0, v0, goal, T, spheres, cost, gradT, cmin, viol, wave_keys=None,
                                  index_base: int = 0):
        """`nbatch` independent batches of the fused rollout + obstacle kernel in one launch (one shared sphere table).
        p0, v0, goal: (nbatch, 3, ld); T, gradT: (nbatch, 3N, ld); cost, cmin, viol: (nbatch, ld); caller-allocated."""
        N = params.horizon
        nb, _, ld = T.shape
        for a, shp, nm in ((p0, (nb, 3, ld), "p0"), (v0, (nb, 3, ld), "v0"), (T, (nb, 3 * N, ld), "T"), (cost, (nb, ld), "cost"),
                           (gradT, (nb, 3 * N, ld), "gradT"), (cmin, (nb, ld), "cmin"), (viol, (nb, ld), "viol")):
            if a is None and nm in ("gradT", "cmin", "viol"):
                continue
            self.be.check(a, nm)
            if tuple(a.shape) != shp:
                raise ValueError(f"{nm}: expected {shp}, got {tuple(a.shape)}")
        if params.has_goal:
            self.be.check(goal, "goal")
        suf = self.be.suffix(T)
        self.be.check(spheres, "spheres")
        if spheres.ndim != 2 or spheres.shape[1] != 4 or spheres.shape[0] > SE3MPC_MAX_SPHERES or self.be.suffix(spheres) != suf:
            raise ValueError(f"spheres: expected (K<={SE3MPC_MAX_SPHERES}, 4) {suf}, got {tuple(spheres.shape)}")
        self.lib.call("rollout_obstacles_batched", suf, ld, ld, nb, self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(cost), self.be.ptr(gradT),
                      self.be.ptr(spheres), spheres.shape[0], self.be.ptr(cmin), self.be.ptr(viol), self.be.ptr(wave_keys),
                      int(index_base), self.be.stream(), params=params)
        return cost, gradT, cmin, viol

    def rollout_iterate(self, params: Params, p0, v0, goal, T, iters: int, step: float, T_out=None, want_grad: bool = True,
                        want_first_cost: bool = False, B: Optional[int] = None, out=None, wave_keys=None, index_base: int = 0,
                        spheres=None, obstacle_weight: float = 1000.0, want_penalty: bool = True):
        """`iters` projected-gradient iterations of the shooting form in ONE launch (thrust sequences stay in registers), then one
        last evaluation.  Single batch: p0, v0, goal (3, ld), T (3N, ld); multi-batch (grid.y): leading batch axis on every operand.
        -> dict(T (like T), cost, gradT | None, cost_first | None).  ``out=(T_out, cost, gradT)`` reuses preallocated outputs;
        T_out may be T itself (in place).
        spheres (K, 4) rows (cx, cy, cz, r): the obstacle-aware loop (se3mpc_rollout_iterate_obstacles_*: the build's extension) --
        the objective gains obstacle_weight * sum max(0, -(|P_k - c_j|^2 - (r_j + safety_margin)^2))^2 and the dict a ``penalty`` entry."""
        N = params.horizon
        batched = T.ndim == 3
        nb = T.shape[0] if batched else 1
        ld = T.shape[-1]
        lead = (nb,) if batched else ()
        for a, rows, nm in ((p0, 3, "p0"), (v0, 3, "v0"), (T, 3 * N, "T")) + (((goal, 3, "goal"),) if params.has_goal else ()):
            self.be.check(a, nm)
            if tuple(a.shape) != lead + (rows, ld):
                raise ValueError(f"{nm}: expected {lead + (rows, ld)}, got {tuple(a.shape)}")
        suf = self.be.suffix(T)
        if out is not None:
            T_out, cost, gradT = out
        else:
            T_out = T_out if T_out is not None else self.be.empty(lead + (3 * N, ld), suf)
            cost = self.be.empty(lead + (ld,), suf)
            gradT = self.be.empty(lead + (3 * N, ld), suf) if want_grad else None
        cost_first = self.be.empty(lead + (ld,), suf) if want_first_cost else None
        nB = self._B(ld, B)
        if spheres is not None:
            self.be.check(spheres, "spheres")
            if spheres.ndim != 2 or spheres.shape[1] != 4 or spheres.shape[0] > SE3MPC_MAX_SPHERES or self.be.suffix(spheres) != suf:
                raise ValueError(f"spheres: expected (K<={SE3MPC_MAX_SPHERES}, 4) {suf}, got {tuple(spheres.shape)}")
            penalty = self.be.empty(lead + (ld,), suf) if want_penalty else None
            self.lib.call("rollout_iterate_obstacles", suf, nB, ld, nb, int(iters), float(step), self.be.ptr(p0), self.be.ptr(v0),
                          self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(T_out), self.be.ptr(cost_first),
                          self.be.ptr(cost), self.be.ptr(gradT), self.be.ptr(spheres if spheres.shape[0] else None), spheres.shape[0],
                          float(obstacle_weight), self.be.ptr(penalty), self.be.ptr(wave_keys), int(index_base), self.be.stream(),
                          params=params)
            return dict(T=T_out, cost=cost, gradT=gradT, cost_first=cost_first, penalty=penalty)
        self.lib.call("rollout_iterate", suf, nB, ld, nb, int(iters), float(step), self.be.ptr(p0), self.be.ptr(v0),
                      self.be.ptr(goal if params.has_goal else None), self.be.ptr(T), self.be.ptr(T_out), self.be.ptr(cost_first),
                      self.be.ptr(cost), self.be.ptr(gradT), self.be.ptr(wave_keys), int(index_base), self.be.stream(), params=params)
        return dict(T=T_out, cost=cost, gradT=gradT, cost_first=cost_first)

    def projected_step(self, params: Params, T, gradT, step: float, out=None, B: Optional[int] = None):
        """One descent step as its own launch: clip(T - step * gradT, thrust box) -> (3N, ld)."""
        N = params.horizon
        self._lane(T, 3 * N, "T"); self._lane(gradT, 3 * N, "gradT")
        suf = self._same(T, gradT)
        ld = T.shape[1]
        T_out = out if out is not None else self.be.empty((3 * N, ld), suf)
        self.lib.call("projected_step", suf, self._B(ld, B), ld, float(step), self.be.ptr(T), self.be.ptr(gradT), self.be.ptr(T_out),
                      self.be.stream(), params=params)
        return T_out

    def is_plan_valid(self, params: Params, P, V=None, B: Optional[int] = None):
        """a16: -> int32 (ld,)."""
        N = params.horizon
        self._lane(P, 3 * N, "P")
        if V is not None:
            self._lane(V, 3 * N, "V")
        suf = self._same(P, V)
        ld = P.shape[1]
        valid = self.be.empty((ld,), "i32")
        self.lib.call("is_plan_valid", suf, self._B(ld, B), ld, self.be.ptr(P), self.be.ptr(V), self.be.ptr(valid),
                      self.be.stream(), params=params)
        return valid

    def argmin(self, cost, index_base: int = 0, out=None):
        """-> int64 (1,) holding the packed unsigned key (cost bits << 32 | index)."""
        self.be.check(cost, "cost")
        suf = self.be.suffix(cost)
        key = out if out is not None else self.be.empty((1,), "i64")
        self.lib.call("argmin", suf, int(cost.shape[0]), self.be.ptr(cost), int(index_base), self.be.ptr(key),
                      self.be.stream())
        return key

    def decode_key(self, key) -> Tuple[int, float]:
        k = int(self.be.to_host(key).reshape(-1)[0]) & 0xFFFFFFFFFFFFFFFF
        return self.lib.key_index(k), self.lib.key_cost(k)

    def population_sums(self, X, cost=None, temperature: float = 1.0, cost_ref: float = 0.0, ref_key=None, B: Optional[int] = None):
        """out[r] = sum_b w_b X[r][b] (r < rows), out[rows] = sum_b w_b in float64; w = 1, or the MPPI weight
        exp(-(cost - cost_ref) / temperature) with cost_ref taken from the packed argmin key `ref_key` (device) when
        given.  X: (rows, ld) lane layout.  -> device float64 (rows + 1,)."""
        self.be.check(X, "X")
        if X.ndim != 2:
            raise ValueError("X: expected (rows, ld)")
        rows, ld = X.shape
        B = self._B(ld, B)
        suf = self.be.suffix(X)
        if cost is not None and (self.be.suffix(self.be.check(cost, "cost")) != suf or cost.shape[-1] < B):
            raise ValueError("cost: same dtype as X and at least B entries")
        out = self.be.empty((rows + 1,), "f64")
        work = self.be.empty((self.lib.population_workspace(rows, B),), "f64")
        self.lib.call("population_sums", suf, rows, B, ld, self.be.ptr(X), self.be.ptr(cost), float(cost_ref), self.be.ptr(ref_key),
                      float(temperature), self.be.ptr(out), self.be.ptr(work), self.be.stream())
        return out

    def spheres_from_grid(self, positions, occupancy, threshold: float = 0.6, target: int = 20, radius: float = 1.0,
                          cap: int = 64):
        """Occupancy grid -> sphere table on the device (f-2).  positions (M, 3), occupancy (M,) in the
        mapper's grid order.  -> (spheres (cap, 4), count int32 (1,)); rows >= count are unspecified."""
        self.be.check(positions, "positions"); self.be.check(occupancy, "occupancy")
        M = occupancy.shape[0]
        if tuple(positions.shape) != (M, 3):
            raise ValueError(f"positions: expected ({M}, 3), got {tuple(positions.shape)}")
        suf = self.be.suffix(positions)
        if self.be.suffix(occupancy) != suf:
            raise ValueError("positions and occupancy must share a dtype")
        spheres = self.be.empty((cap, 4), suf)
        count = self.be.empty((1,), "i32")
        self.lib.call("spheres_from_grid", suf, self.be.ptr(positions), self.be.ptr(occupancy), M, float(threshold),
                      int(target), float(radius), self.be.ptr(spheres), cap, self.be.ptr(count), self.be.stream())
        return spheres, count

    def transpose(self, a):
        """(rows, cols) -> (cols, rows), through the LDS-tiled kernel."""
        self.be.check(a, "a")
        if a.ndim != 2:
            raise ValueError("transpose: 2-D only")
        suf = self.be.suffix(a)
        rows, cols = a.shape
        out = self.be.empty((cols, rows), suf)
        self.lib.call("transpose", suf, rows, cols, self.be.ptr(a), cols, self.be.ptr(out), rows, self.be.stream())
        return out

    # ------------------------------------------------------------------ consumer side of the contract (per-drone rows)
    def controller_state(self, cp: ControllerParams, B: int):
        """Fresh controller members for B drones (GeometricController.__init__ / reset): float64 (B, 12)."""
        st = self.be.empty((B, CONTROLLER_STATE_WORDS), "f64")
        self.lib.controller_reset(cp, B, self.be.ptr(st), self.be.stream())
        return st

    def _rows3(self, a, B, name, suf=None):
        self.be.check(a, name)
        if tuple(a.shape) != (B, 3) or (suf is not None and self.be.suffix(a) != suf):
            raise ValueError(f"{name}: expected ({B}, 3) {suf or ''}, got {tuple(a.shape)}")
        return a

    def control(self, cp: ControllerParams, state, time, pos, vel, att, omega, dpos, dvel, dacc=None, yaw=None, yaw_rate=None,
                want_body_rate: bool = False):
        """compute_control (+ compute_body_rate_command) for B drones.  time: float64 (B,); pos .. dvel, dacc: (B, 3);
        yaw, yaw_rate: (B,) or None (0).  `state` (B, 12) float64 is updated in place.
        -> dict(thrust (B,), torque (B,3), flags int32 (B,)[, body_thrust (B,), body_rates (B,3)])."""
        B = time.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega"), (dpos, "dpos"), (dvel, "dvel")):
            self._rows3(a, B, nm, suf)
        if dacc is not None:
            self._rows3(dacc, B, "dacc", suf)
        self.be.check(time, "time"); self.be.check(state, "state")
        if self.be.suffix(time) != "f64" or tuple(state.shape) != (B, CONTROLLER_STATE_WORDS) or self.be.suffix(state) != "f64":
            raise ValueError("time: float64 (B,); state: float64 (B, 12)")
        thrust, torque = self.be.empty((B,), suf), self.be.empty((B, 3), suf)
        flags = self.be.empty((B,), "i32")
        bt = self.be.empty((B,), suf) if want_body_rate else None
        br = self.be.empty((B, 3), suf) if want_body_rate else None
        self.lib.loop_call("control", suf, cp, B, self.be.ptr(time), self.be.ptr(pos), self.be.ptr(vel), self.be.ptr(att),
                           self.be.ptr(omega), self.be.ptr(dpos), self.be.ptr(dvel), self.be.ptr(dacc), self.be.ptr(yaw),
                           self.be.ptr(yaw_rate), self.be.ptr(state), self.be.ptr(thrust), self.be.ptr(torque), self.be.ptr(bt),
                           self.be.ptr(br), self.be.ptr(flags), self.be.stream())
        out = dict(thrust=thrust, torque=torque, flags=flags)
        if want_body_rate:
            out.update(body_thrust=bt, body_rates=br)
        return out

    def control_fast(self, cp: ControllerParams, state, dt: float, pos, vel, att, omega, dpos, dvel, dacc=None, yaw=None, yaw_rate=None,
                     vehicle_mass: float = 1.0, vehicle_gravity: float = 9.80665):
        """compute_control_fast / compute_control_from_fast_state (controller.py:253-411, :728-768) for B drones: dt in seconds is given;
        vehicle_mass / vehicle_gravity = get_control_constants() (common/vehicle_params.py:19-23 by default).  `state` (B, 12) float64
        is the record se3mpc_control_* uses, updated in place.  -> dict(thrust (B,), torque (B,3), flags int32 (B,))."""
        B = pos.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega"), (dpos, "dpos"), (dvel, "dvel")):
            self._rows3(a, B, nm, suf)
        if dacc is not None:
            self._rows3(dacc, B, "dacc", suf)
        for a, nm in ((yaw, "yaw"), (yaw_rate, "yaw_rate")):
            if a is not None:
                self.be.check(a, nm)
                if tuple(a.shape) != (B,) or self.be.suffix(a) != suf:
                    raise ValueError(f"{nm}: expected ({B},) {suf}")
        self.be.check(state, "state")
        if tuple(state.shape) != (B, CONTROLLER_STATE_WORDS) or self.be.suffix(state) != "f64":
            raise ValueError("state: float64 (B, 12)")
        thrust, torque = self.be.empty((B,), suf), self.be.empty((B, 3), suf)
        flags = self.be.empty((B,), "i32")
        self.lib.loop_call("control_fast", suf, cp, float(vehicle_mass), float(vehicle_gravity), B, float(dt), self.be.ptr(pos), self.be.ptr(vel),
                           self.be.ptr(att), self.be.ptr(omega), self.be.ptr(dpos), self.be.ptr(dvel), self.be.ptr(dacc), self.be.ptr(yaw),
                           self.be.ptr(yaw_rate), self.be.ptr(state), self.be.ptr(thrust), self.be.ptr(torque), self.be.ptr(flags),
                           self.be.stream())
        return dict(thrust=thrust, torque=torque, flags=flags)

    def monte_carlo(self, params: Params, cp: ControllerParams, sp: SimulatorParams, state, time, pos, vel, att, omega, goal, cycles: int,
                    substeps: int, sim_dt: float, wind=None, want_last_plan: bool = False):
        """se3mpc_monte_carlo_*: `cycles` x (solve from the drone's own state, `substeps` control + simulator steps against the fresh plan) for B
        drones in ONE launch, in place on (state, time, pos, vel, att, omega).  -> dict(overflowed: int32 (1,) device array -- not 0 means
        a solve needed more L-BFGS pairs than the launch's LDS image holds and the run must be repeated with solve + closed_loop launches;
        x, accelerations, info of the last cycle's plan if asked for)."""
        B = pos.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega"), (goal, "goal")):
            self._rows3(a, B, nm, suf)
        self.be.check(state, "state"); self.be.check(time, "time")
        if tuple(state.shape) != (B, CONTROLLER_STATE_WORDS) or self.be.suffix(state) != "f64" or tuple(time.shape) != (B,) or self.be.suffix(time) != "f64":
            raise ValueError("state: float64 (B, 12); time: float64 (B,)")
        w_stride = 0
        if wind is not None:
            self.be.check(wind, "wind")
            if self.be.suffix(wind) != suf or wind.shape[-1] != 3 or (wind.ndim == 2 and wind.shape[0] != B):
                raise ValueError("wind: (3,) or (B, 3)")
            w_stride = 3 if wind.ndim == 2 else 0
        N = params.horizon
        over = self.be.empty((1,), "i32")
        X = self.be.empty((B, 9 * N), suf) if want_last_plan else None
        acc = self.be.empty((B, N, 3), suf) if want_last_plan else None
        info = self.be.empty((B * INFO_DTYPE.itemsize,), "u8") if want_last_plan else None
        self.lib.loop_call("monte_carlo", suf, params, cp, sp, B, int(cycles), int(substeps), float(sim_dt), self.be.ptr(goal), self.be.ptr(wind), w_stride,
                           self.be.ptr(time), self.be.ptr(pos), self.be.ptr(vel), self.be.ptr(att), self.be.ptr(omega), self.be.ptr(state),
                           self.be.ptr(X), self.be.ptr(acc), self.be.ptr(info), self.be.ptr(over), self.be.stream())
        return dict(overflowed=over, x=X, accelerations=acc, info=info)

    def controller_integral_update(self, cp: ControllerParams, state, vel_error, dt: float, saturation=None) -> None:
        """_update_integral_error(vel_error, dt, thrust_saturated, torque_saturated) (controller.py:536-564) for B drones, in place on
        `state`.  saturation: int32 (B,) bit 0 thrust, bits 1..3 torque x/y/z, or None."""
        B = vel_error.shape[0]
        suf = self.be.suffix(vel_error)
        self._rows3(vel_error, B, "vel_error", suf)
        self.be.check(state, "state")
        if tuple(state.shape) != (B, CONTROLLER_STATE_WORDS) or self.be.suffix(state) != "f64":
            raise ValueError("state: float64 (B, 12)")
        if saturation is not None:
            self.be.check(saturation, "saturation")
            if tuple(saturation.shape) != (B,) or not str(saturation.dtype).endswith("int32"):     # (a float32 (B,) array would be read as flag bits)
                raise ValueError("saturation: int32 (B,)")
        self.lib.loop_call("controller_integral_update", suf, cp, B, self.be.ptr(vel_error), float(dt), self.be.ptr(saturation), self.be.ptr(state),
                           self.be.stream())

    def controller_attitude_torque(self, cp: ControllerParams, state, att, omega, b3_des, yaw=None, yaw_rate=None, inertia=None):
        """_geometric_attitude_control / _fast_geometric_attitude_control (controller.py:643-704, :348-411) for B drones.
        inertia: None (diag of cp.inertia) or a host 3x3.  -> dict(torque (B,3), flags int32 (B,)); `state` takes the unsaturated torques
        and the torque saturation flags."""
        B = att.shape[0]
        suf = self.be.suffix(att)
        for a, nm in ((att, "att"), (omega, "omega"), (b3_des, "b3_des")):
            self._rows3(a, B, nm, suf)
        self.be.check(state, "state")
        if tuple(state.shape) != (B, CONTROLLER_STATE_WORDS) or self.be.suffix(state) != "f64":
            raise ValueError("state: float64 (B, 12)")
        mat = None
        if inertia is not None:
            m = np.asarray(inertia, dtype=np.float64)
            if m.shape != (3, 3):
                raise ValueError("inertia: a 3x3 matrix")
            mat = (_C.c_double * 9)(*m.reshape(-1))
        torque, flags = self.be.empty((B, 3), suf), self.be.empty((B,), "i32")
        self.lib.loop_call("controller_attitude_torque", suf, cp, B, self.be.ptr(att), self.be.ptr(omega), self.be.ptr(b3_des), self.be.ptr(yaw),
                           self.be.ptr(yaw_rate), mat, self.be.ptr(state), self.be.ptr(torque), self.be.ptr(flags), self.be.stream())
        return dict(torque=torque, flags=flags)

    def controller_desired_frame(self, cp: ControllerParams, yaw_vector, b3_des, current_yaw=None, method: int = -1):
        """_detect_yaw_singularity (controller.py:160-189) + the desired frame: method -1 = as _geometric_attitude_control builds it,
        0..3 = _handle_yaw_singularity with that fallback (:191-252).  -> dict(frame (B,9) = b1|b2|b3, cos_angle (B,), singular int32 (B,))."""
        B = yaw_vector.shape[0]
        suf = self.be.suffix(yaw_vector)
        self._rows3(yaw_vector, B, "yaw_vector", suf); self._rows3(b3_des, B, "b3_des", suf)
        frame, ca, sg = self.be.empty((B, 9), suf), self.be.empty((B,), suf), self.be.empty((B,), "i32")
        self.lib.loop_call("controller_desired_frame", suf, cp, B, int(method), self.be.ptr(yaw_vector), self.be.ptr(b3_des), self.be.ptr(current_yaw),
                           self.be.ptr(frame), self.be.ptr(ca), self.be.ptr(sg), self.be.stream())
        return dict(frame=frame, cos_angle=ca, singular=sg)

    def _plan_args(self, B, suf, timestamps, P, V, A, strides):
        self.be.check(timestamps, "timestamps")
        if self.be.suffix(timestamps) != "f64":
            raise ValueError("timestamps are float64")
        N = timestamps.shape[-1]
        ts_stride = N if timestamps.ndim == 2 else 0
        if timestamps.ndim == 2 and timestamps.shape[0] != B:
            raise ValueError("timestamps: (N,) or (B, N)")
        if strides is None:
            def st_of(a, nm):
                if a is None:
                    return 0
                self.be.check(a, nm)
                if self.be.suffix(a) != suf or a.ndim not in (2, 3) or a.shape[-2:] != (N, 3) or (a.ndim == 3 and a.shape[0] != B):
                    raise ValueError(f"{nm}: expected ({N}, 3) or ({B}, {N}, 3) {suf}, got {tuple(a.shape)}")
                return 3 * N if a.ndim == 3 else 0
            sP, sV, sA = st_of(P, "P"), st_of(V, "V"), st_of(A, "A")
        else:
            sP, sV, sA = (int(x) for x in strides)
            if hasattr(self.be, "elements_from"):              # explicit strides read past the views' own shapes: check the storage instead
                for a, st, nm in ((P, sP, "P"), (V, sV, "V"), (A, sA, "A")):
                    if a is not None and (st < 0 or (B - 1) * st + 3 * N > self.be.elements_from(a)):
                        raise ValueError(f"{nm}: stride {st} x {B} plans of {N} rows runs past the tensor's storage")
        return N, ts_stride, sP, sV, sA

    def control_plan(self, cp: ControllerParams, state, time, sample_time, pos, vel, att, omega, timestamps, P, V=None, A=None,
                     strides=None, want_body_rate: bool = False, want_target: bool = False):
        """compute_control_from_trajectory / compute_body_rate_from_trajectory for B drones: the target is the plan sampled at
        sample_time (float64 (B,)).  Plans as in :meth:`closed_loop`.  -> dict like :meth:`control` (+ target (B, 9))."""
        B = time.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega")):
            self._rows3(a, B, nm, suf)
        N, ts_stride, sP, sV, sA = self._plan_args(B, suf, timestamps, P, V, A, strides)
        thrust, torque = self.be.empty((B,), suf), self.be.empty((B, 3), suf)
        flags = self.be.empty((B,), "i32")
        bt = self.be.empty((B,), suf) if want_body_rate else None
        br = self.be.empty((B, 3), suf) if want_body_rate else None
        tg = self.be.empty((B, 9), suf) if want_target else None
        self.lib.loop_call("control_plan", suf, cp, B, self.be.ptr(time), self.be.ptr(sample_time), self.be.ptr(pos), self.be.ptr(vel),
                           self.be.ptr(att), self.be.ptr(omega), N, self.be.ptr(timestamps), ts_stride, self.be.ptr(P), sP, self.be.ptr(V), sV,
                           self.be.ptr(A), sA, self.be.ptr(state), self.be.ptr(thrust), self.be.ptr(torque), self.be.ptr(bt), self.be.ptr(br),
                           self.be.ptr(flags), self.be.ptr(tg), self.be.stream())
        out = dict(thrust=thrust, torque=torque, flags=flags)
        if want_body_rate:
            out.update(body_thrust=bt, body_rates=br)
        if want_target:
            out["target"] = tg
        return out

    def simulator_step(self, sp: SimulatorParams, time, pos, vel, att, omega, thrust, torque, dt: float, wind=None):
        """DroneSimulator.step for B drones: advances time, pos, vel, att, omega in place."""
        B = time.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega"), (torque, "torque")):
            self._rows3(a, B, nm, suf)
        self.be.check(thrust, "thrust")
        w_stride = 0
        if wind is not None:
            self.be.check(wind, "wind")
            w_stride = 3 if wind.ndim == 2 else 0
        self.lib.loop_call("simulator_step", suf, sp, B, float(dt), self.be.ptr(thrust), self.be.ptr(torque), self.be.ptr(wind), w_stride,
                           self.be.ptr(time), self.be.ptr(pos), self.be.ptr(vel), self.be.ptr(att), self.be.ptr(omega), self.be.stream())

    def closed_loop(self, cp: ControllerParams, sp: SimulatorParams, state, time, pos, vel, att, omega, timestamps, P, V=None, A=None,
                    nsteps: int = 1, sim_dt: float = 0.01, strides=None, wind=None, gust=None, stop_at_plan_end: bool = True,
                    log: bool = False):
        """`nsteps` x (plan sample -> compute_control -> DroneSimulator.step) for B drones in ONE launch; time, pos, vel, att,
        omega and `state` are updated in place.  Plans: timestamps float64 (N,) shared or (B, N); P, V, A either (N, 3) shared,
        (B, N, 3) per drone, or -- with ``strides=(strideP, strideV, strideA)`` in elements -- views into the solver's outputs
        (P = X, V = X[:, 3N:], A = accelerations).  wind: None, (3,) or (B, 3) newtons; gust = (step, (wx, wy, wz)).
        -> dict(steps_taken int32 (B,)[, log_state (nsteps, B, 12), log_cmd (nsteps, B, 4), log_time (nsteps, B)])."""
        B = time.shape[0]
        suf = self.be.suffix(pos)
        for a, nm in ((pos, "pos"), (vel, "vel"), (att, "att"), (omega, "omega")):
            self._rows3(a, B, nm, suf)
        N, ts_stride, sP, sV, sA = self._plan_args(B, suf, timestamps, P, V, A, strides)
        w_stride = 0
        if wind is not None:
            self.be.check(wind, "wind")
            if self.be.suffix(wind) != suf or wind.shape[-1] != 3:
                raise ValueError("wind: (3,) or (B, 3)")
            w_stride = 3 if wind.ndim == 2 else 0
        gust_step, gust_vec = -1, None
        if gust is not None:
            gust_step = int(gust[0])
            gust_vec = (_C.c_double * 3)(*[float(x) for x in gust[1]])
        ls = self.be.empty((nsteps, B, 12), suf) if log else None
        lc = self.be.empty((nsteps, B, 4), suf) if log else None
        lt = self.be.empty((nsteps, B), "f64") if log else None
        taken = self.be.empty((B,), "i32")
        self.lib.loop_call("closed_loop", suf, cp, sp, B, int(nsteps), float(sim_dt), N, self.be.ptr(timestamps), ts_stride,
                           self.be.ptr(P), sP, self.be.ptr(V), sV, self.be.ptr(A), sA, self.be.ptr(time), self.be.ptr(pos),
                           self.be.ptr(vel), self.be.ptr(att), self.be.ptr(omega), self.be.ptr(state), self.be.ptr(wind), w_stride,
                           gust_step, gust_vec, int(bool(stop_at_plan_end)), self.be.ptr(ls), self.be.ptr(lc), self.be.ptr(lt),
                           self.be.ptr(taken), self.be.stream())
        out = dict(steps_taken=taken)
        if log:
            out.update(log_state=ls, log_cmd=lc, log_time=lt)
        return out

    # ------------------------------------------------------------------ problem layout
    def solve(self, params: Params, p0, v0, goal, x0=None, want_trajectory=True, out=None):
        """a7 (+a3, a4, a11, a12): batched L-BFGS-B solve, one wavefront per problem.
        p0, v0, goal: (B, 3);  x0: (B, 9N) or None (reference cold start).  want_trajectory: True (all outputs), False (x and info only:
        restart sweeps) or "accelerations" (x, info, accelerations: the kernel skips attitudes / rates / thrust magnitudes).
        out: a dict this method returned for the same shapes and output set -- its tensors are overwritten and returned (steady-state use).
        -> dict(x (B,9N), info (device bytes), acc/att/rates (B,N,3), thrust (B,N))."""
        N = params.horizon
        for a, nm in ((p0, "p0"), (v0, "v0")) + (((goal, "goal"),) if params.has_goal else ()):
            self.be.check(a, nm)
            if a.ndim != 2 or a.shape[1] != 3 or a.shape[0] != p0.shape[0]:
                raise ValueError(f"{nm}: expected (B, 3), got {tuple(a.shape)}")
        suf = self.be.suffix(p0)
        B = p0.shape[0]
        if x0 is not None:
            self.be.check(x0, "x0")
            if tuple(x0.shape) != (B, 9 * N) or self.be.suffix(x0) != suf:
                raise ValueError(f"x0: expected ({B}, {9 * N}) {suf}")
        if out is not None:
            # steady state: the dict a previous call with the same shapes returned is written again (no allocations)
            X, info, acc, att, rates, thrust = (out[k] for k in ("x", "info", "accelerations", "attitudes", "body_rates", "thrusts"))
            if tuple(X.shape) != (B, 9 * N) or self.be.suffix(X) != suf or info.shape[0] != B * INFO_DTYPE.itemsize:
                raise ValueError(f"out: buffers of another solve shape (expected x ({B}, {9 * N}) {suf})")
            if (acc is None) != (not want_trajectory) or (att is None) != (want_trajectory is not True):
                raise ValueError("out: buffers of another output set (want_trajectory)")
        else:
            X = self.be.empty((B, 9 * N), suf)
            info = self.be.empty((B * INFO_DTYPE.itemsize,), "u8")
            acc = att = rates = thrust = None
            if want_trajectory == "accelerations":            # what a closed loop reading the plan in place needs next to x (se3mpc_closed_loop_*)
                acc = self.be.empty((B, N, 3), suf)
            elif want_trajectory:
                acc, att, rates = (self.be.empty((B, N, 3), suf) for _ in range(3))
                thrust = self.be.empty((B, N), suf)
        self.lib.call("solve", suf, B, self.be.ptr(p0), self.be.ptr(v0), self.be.ptr(goal if params.has_goal else None),
                      self.be.ptr(x0), self.be.ptr(X), self.be.ptr(info), self.be.ptr(acc), self.be.ptr(att),
                      self.be.ptr(rates), self.be.ptr(thrust), self.be.stream(), params=params)
        return dict(x=X, info=info, accelerations=acc, attitudes=att, body_rates=rates, thrusts=thrust)

    def solve_packed(self, params: Params, inputs, x0=None, out=None, host_mapped=False, stream=None):
        """Latency-oriented form of :meth:`solve` for the planner: ONE device buffer in, ONE out.
        inputs: (3, B, 3) = stacked (p0, v0, goal) [goal ignored when has_goal == 0];
        out: uint8 buffer of :meth:`packed_size` bytes (allocated if None) laid out as
        [X | acc | att | rates | thrust | info]; decode on the host with :meth:`unpack_solution`.
        host_mapped=True: `inputs`, `x0` and `out` are PINNED host tensors; the kernel reads and writes them in
        place over the host link (hipHostMalloc memory is mapped into the device's address space under the
        same address), which for a handful of problems beats two staged copies.  The caller synchronises the
        stream before reading `out`."""
        N = params.horizon
        if host_mapped:
            for name, a in (("inputs", inputs), ("x0", x0), ("out", out)):
                if a is None and name == "x0":
                    continue
                if a is None or not self.be.is_host_mapped(a):
                    raise TypeError(f"{name}: host_mapped=True needs a pinned, contiguous host tensor")
        else:
            self.be.check(inputs, "inputs")
        if inputs.ndim != 3 or inputs.shape[0] != 3 or inputs.shape[2] != 3:
            raise ValueError(f"inputs: expected (3, B, 3), got {tuple(inputs.shape)}")
        suf = self.be.suffix(inputs)
        B = inputs.shape[1]
        esz = 4 if suf == "f32" else 8
        if out is None:
            out = self.be.empty((self.packed_size(B, N, suf),), "u8")
        base = self.be.ptr(out)
        o_x, o_acc, o_att, o_rates, o_thr, o_info, _ = self._packed_offsets(B, N, esz)
        step = B * 3 * esz
        pin = self.be.ptr(inputs)
        self.lib.call("solve", suf, B, pin, pin + step, (pin + 2 * step) if params.has_goal else 0, self.be.ptr(x0),
                      base + o_x, base + o_info, base + o_acc, base + o_att, base + o_rates, base + o_thr,
                      self.be.stream() if stream is None else stream, params=params)
        return out

    @staticmethod
    def _packed_offsets(B, N, esz):
        o_x = 0
        o_acc = o_x + B * 9 * N * esz
        o_att = o_acc + B * 3 * N * esz
        o_rates = o_att + B * 3 * N * esz
        o_thr = o_rates + B * 3 * N * esz
        o_info = (o_thr + B * N * esz + 7) // 8 * 8
        return o_x, o_acc, o_att, o_rates, o_thr, o_info, o_info + B * INFO_DTYPE.itemsize

    def packed_size(self, B, N, suffix) -> int:
        return self._packed_offsets(B, N, 4 if suffix == "f32" else 8)[-1]

    def unpack_solution(self, host_bytes: np.ndarray, B: int, N: int, suffix: str):
        """Decode a packed result into float64 arrays that do not alias `host_bytes` (which the caller reuses):
        f64 = ONE copy of the buffer and views into it, f32 = one widening cast per field."""
        esz, dt = (4, np.float32) if suffix == "f32" else (8, np.float64)
        o_x, o_acc, o_att, o_rates, o_thr, o_info, end = self._packed_offsets(B, N, esz)
        raw = host_bytes[:end].copy()
        # the five float fields are one contiguous run: ONE view (f64) or ONE widening cast (f32), then slices of it
        nfl = (o_thr + B * N * esz) // esz
        allf = np.frombuffer(raw, dtype=dt, count=nfl)
        if esz == 4:
            allf = allf.astype(np.float64)
        f = lambda lo, cnt, shape: allf[lo // esz:lo // esz + cnt].reshape(shape)
        return dict(x=f(o_x, B * 9 * N, (B, 9 * N)), accelerations=f(o_acc, B * 3 * N, (B, N, 3)),
                    attitudes=f(o_att, B * 3 * N, (B, N, 3)), body_rates=f(o_rates, B * 3 * N, (B, N, 3)),
                    thrusts=f(o_thr, B * N, (B, N)), info=np.frombuffer(raw, dtype=INFO_DTYPE, count=B, offset=o_info))

    def info_to_host(self, info) -> np.ndarray:
        """Device bytes of se3mpc_solve_info[B] -> NumPy structured array (synchronises)."""
        return np.frombuffer(self.be.to_host(info).tobytes(), dtype=INFO_DTYPE)
