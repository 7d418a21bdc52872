"""CPU oracle for the SE(3) MPC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the arithmetic of DART-Planner's
``src/dart_planner/planning/se3_mpc_planner.py`` (called ``planner.py`` below) under the
unit-stripped semantics of SURVEY.md section 0-6 ("every quantity is an SI magnitude").
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``dart_planner_amd`` never does, and has no CPU fallback.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against vectors
produced by running the reference's own code in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), and against the known-answer
vector of SURVEY.md Appendix B.  The one third-party algorithm on the path, SciPy's
L-BFGS-B (``scipy.optimize.minimize``; reference pins scipy==1.16.0, this image has 1.15.3,
both the C translation of L-BFGS-B 3.0), is *called*, exactly as the reference calls it
(planner.py:256-268); ``oracle/lbfgsb_port.py`` is the independent restatement of that
algorithm that the HIP solver is modelled on.

Batched functions take arrays with a leading batch axis; decision vectors use the
reference's packing x = [P(N,3).ravel() | V(N,3).ravel() | T(N,3).ravel()] (planner.py:361-376).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class OracleConfig:
    """SE3MPCConfig (planner.py:36-79) + ctor constants (planner.py:149-151), unit-stripped."""
    prediction_horizon: int = 6
    dt: float = 1.0 / 400.0            # planner.py:99-105 forces the timing manager's dt
    max_velocity: float = 10.0
    max_acceleration: float = 15.0
    max_thrust: float = 25.0
    min_thrust: float = 2.0
    max_tilt_angle: float = math.pi / 4
    position_weight: float = 100.0
    velocity_weight: float = 10.0
    acceleration_weight: float = 1.0
    thrust_weight: float = 0.1
    safety_margin: float = 1.5
    max_iterations: int = 15
    convergence_tolerance: float = 5e-2
    mass: float = 1.5
    gravity: float = 9.81
    position_bound: float = 100.0      # planner.py:383-384

    @property
    def hover_thrust(self) -> float:   # planner.py:151
        return self.mass * self.gravity


# --------------------------------------------------------------------------- packing
def pack(P, V, T):
    """planner.py:361-367 (batched: (...,N,3) x3 -> (...,9N))."""
    P, V, T = np.asarray(P, float), np.asarray(V, float), np.asarray(T, float)
    lead = P.shape[:-2]
    return np.concatenate([P.reshape(lead + (-1,)), V.reshape(lead + (-1,)), T.reshape(lead + (-1,))], axis=-1)


def unpack(x, N):
    """planner.py:369-376 (batched)."""
    x = np.asarray(x, float)
    lead = x.shape[:-1]
    return (x[..., :3 * N].reshape(lead + (N, 3)), x[..., 3 * N:6 * N].reshape(lead + (N, 3)),
            x[..., 6 * N:9 * N].reshape(lead + (N, 3)))


# --------------------------------------------------------------------------- a3 / a4
def straight_line_init(p0, v0, goal, cfg: OracleConfig):
    """Cold start, planner.py:329-359.  goal=None -> hover-in-place branch (:354-357).
    Batched over leading axes of p0/v0/goal."""
    N, dt = cfg.prediction_horizon, cfg.dt
    p0, v0 = np.asarray(p0, float), np.asarray(v0, float)
    lead = p0.shape[:-1]
    P = np.zeros(lead + (N, 3)); V = np.zeros(lead + (N, 3)); T = np.zeros(lead + (N, 3))
    if goal is None:
        P[...] = p0[..., None, :]
        V[..., 0, :] = v0
    else:
        goal = np.asarray(goal, float)
        for i in range(N):
            alpha = i / max(N - 1, 1)
            P[..., i, :] = (1 - alpha) * p0 + alpha * goal
        V[..., 0, :] = v0
        for i in range(1, N):
            V[..., i, :] = (P[..., i, :] - P[..., i - 1, :]) / dt
    T[..., 2] = cfg.hover_thrust
    return pack(P, V, T)


def bounds(cfg: OracleConfig) -> np.ndarray:
    """Box, planner.py:378-402 -> (9N, 2)."""
    N = cfg.prediction_horizon
    b = np.zeros((9 * N, 2))
    b[:3 * N] = (-cfg.position_bound, cfg.position_bound)
    b[3 * N:6 * N] = (-cfg.max_velocity, cfg.max_velocity)
    tilt = cfg.max_thrust * np.sin(cfg.max_tilt_angle)
    tb = b[6 * N:].reshape(N, 3, 2)
    tb[:, 0] = (-tilt, tilt)
    tb[:, 1] = (-tilt, tilt)
    tb[:, 2] = (cfg.min_thrust, cfg.max_thrust)
    return b


def warm_start(prev: Dict[str, np.ndarray], p0, v0, goal, cfg: OracleConfig):
    """planner.py:294-327 (dead code in the reference: last_solution is never assigned)."""
    N = cfg.prediction_horizon
    P = np.zeros((N, 3)); V = np.zeros((N, 3)); T = np.zeros((N, 3))
    P[0], V[0] = p0, v0
    plen = len(prev["positions"])
    shift = 0
    if plen > 1:
        shift = min(N - 1, plen - 1)
        P[1:shift + 1] = prev["positions"][1:shift + 1]
        V[1:shift + 1] = prev["velocities"][1:shift + 1]
        T[:shift] = prev["thrust_vectors"][1:shift + 1]
    if goal is not None:
        shift = min(N - 1, plen - 1)
        for i in range(shift + 1, N):
            alpha = (i - shift) / max(N - shift, 1)
            P[i] = (1 - alpha) * P[shift] + alpha * np.asarray(goal, float)
            T[i] = (0, 0, cfg.hover_thrust)
    return pack(P, V, T)


# --------------------------------------------------------------------------- a5 / a6
def objective(x, goal, cfg: OracleConfig):
    """planner.py:516-550, batched: x (...,9N), goal (...,3) or None -> (...)."""
    N = cfg.prediction_horizon
    P, V, T = unpack(x, N)
    e3 = np.array([0.0, 0.0, 1.0])
    cost = np.zeros(P.shape[:-2])
    if goal is not None:
        g = np.asarray(goal, float)[..., None, :]
        cost = cost + cfg.position_weight * np.sum((P - g) ** 2, axis=(-1, -2))
    cost = cost + cfg.velocity_weight * np.sum(V ** 2, axis=(-1, -2))
    acc = T / cfg.mass - cfg.gravity * e3
    cost = cost + cfg.acceleration_weight * np.sum(acc ** 2, axis=(-1, -2))
    cost = cost + cfg.thrust_weight * np.sum((T - cfg.hover_thrust * e3) ** 2, axis=(-1, -2))
    if goal is not None:
        cost = cost + 10 * cfg.position_weight * np.sum((P[..., -1, :] - np.asarray(goal, float)) ** 2, axis=-1)
    return cost


def gradient(x, goal, cfg: OracleConfig):
    """planner.py:552-580 -- the reference's "analytic gradient", reproduced AS IS: no
    acceleration term, no hover offset, no x10 terminal term."""
    N = cfg.prediction_horizon
    P, V, T = unpack(x, N)
    gP = np.zeros_like(P)
    if goal is not None:
        gP = 2 * cfg.position_weight * (P - np.asarray(goal, float)[..., None, :])
    gV = 2 * cfg.velocity_weight * V
    gT = 2 * cfg.thrust_weight * T
    return pack(gP, gV, gT)


def objective_loops(x, goal, cfg: OracleConfig) -> float:
    """planner.py:516-550 in the reference's own shape (Python loops over k, one problem):
    used for the 'reference-shaped' CPU baseline and as a bit-level cross-check."""
    N = cfg.prediction_horizon
    P, V, T = unpack(x, N)
    cost = 0.0
    if goal is not None:
        for k in range(N):
            e = P[k] - goal
            cost += cfg.position_weight * np.sum(e ** 2)
    for k in range(N):
        cost += cfg.velocity_weight * np.sum(V[k] ** 2)
    for k in range(N):
        a = T[k] / cfg.mass - np.array([0, 0, cfg.gravity])
        cost += cfg.acceleration_weight * np.sum(a ** 2)
    for k in range(N):
        d = T[k] - np.array([0, 0, cfg.hover_thrust])
        cost += cfg.thrust_weight * np.sum(d ** 2)
    if goal is not None:
        e = P[-1] - goal
        cost += 10 * cfg.position_weight * np.sum(e ** 2)
    return cost


def objective_ordered(x, goal, cfg: OracleConfig) -> float:
    """planner.py:516-550 for ONE problem with the reference's ACCUMULATION ORDER and without its Python loops: the per-step terms are
    formed vectorised (each `w * np.sum(e_k ** 2)` is the same three-element sum), then added left to right exactly as the reference's
    `cost += ...` statements run (np.cumsum is sequential).  Bit-identical to :func:`objective_loops` (tests/test_oracle_golden.py), ~20x
    faster.  This is what :func:`solve` hands to SciPy: L-BFGS-B's line search can turn a last-bit difference of f into another
    outcome (tests/golden/bifurcation_case: batched np.sum order -> (nit, nfev) = (3, 24), f = 2728.05; the reference's order ->
    (3, 27), f = 2680.18), so the oracle must not merely equal the reference's objective to 1e-16, it must round like it."""
    N = cfg.prediction_horizon
    P, V, T = unpack(np.asarray(x, float), N)
    terms = []
    if goal is not None:
        g = np.asarray(goal, float)
        terms.append(cfg.position_weight * np.sum((P - g) ** 2, axis=-1))
    terms.append(cfg.velocity_weight * np.sum(V ** 2, axis=-1))
    a = T / cfg.mass - np.array([0, 0, cfg.gravity])
    terms.append(cfg.acceleration_weight * np.sum(a ** 2, axis=-1))
    d = T - np.array([0, 0, cfg.hover_thrust])
    terms.append(cfg.thrust_weight * np.sum(d ** 2, axis=-1))
    if goal is not None:
        terms.append(np.atleast_1d(10 * cfg.position_weight * np.sum((P[-1] - g) ** 2)))
    return float(np.cumsum(np.concatenate(terms))[-1])


def gradient_loops(x, goal, cfg: OracleConfig) -> np.ndarray:
    """planner.py:552-580 with the reference's loops."""
    N = cfg.prediction_horizon
    P, V, T = unpack(x, N)
    g = np.zeros_like(x)
    gp, gv, gt = g[:3 * N].reshape(N, 3), g[3 * N:6 * N].reshape(N, 3), g[6 * N:].reshape(N, 3)
    if goal is not None:
        for i in range(N):
            gp[i] = 2 * cfg.position_weight * (P[i] - goal)
    for i in range(N):
        gv[i] = 2 * cfg.velocity_weight * V[i]
    for i in range(N):
        gt[i] = 2 * cfg.thrust_weight * T[i]
    return g


# --------------------------------------------------------------------------- a8 / a9 / a10
def dynamics_residual(x, p0, v0, cfg: OracleConfig):
    """planner.py:426-462, batched -> (...,6N).  Row order: P0-p0, V0-v0, then per k the
    position residual (3) followed by the velocity residual (3)."""
    N, dt = cfg.prediction_horizon, cfg.dt
    P, V, T = unpack(x, N)
    e3 = np.array([0.0, 0.0, 1.0])
    A = T / cfg.mass - cfg.gravity * e3
    lead = P.shape[:-2]
    r = np.zeros(lead + (2 * N, 3))
    r[..., 0, :] = P[..., 0, :] - np.asarray(p0, float)
    r[..., 1, :] = V[..., 0, :] - np.asarray(v0, float)
    if N > 1:
        rp = P[..., 1:, :] - P[..., :-1, :] - V[..., :-1, :] * dt - 0.5 * A[..., :-1, :] * dt ** 2
        rv = V[..., 1:, :] - V[..., :-1, :] - A[..., :-1, :] * dt
        r[..., 2::2, :] = rp
        r[..., 3::2, :] = rv
    return r.reshape(lead + (6 * N,))


def obstacle_residual(x, centres, radii, cfg: OracleConfig):
    """planner.py:499-514, batched -> (...,N*K), k-major then obstacle."""
    N = cfg.prediction_horizon
    P, _, _ = unpack(x, N)
    c = np.asarray(centres, float); r = np.asarray(radii, float)
    d2 = np.sum((P[..., :, None, :] - c[None, :, :]) ** 2, axis=-1)      # (...,N,K)
    safe = (r + cfg.safety_margin) ** 2
    return (d2 - safe).reshape(P.shape[:-2] + (N * len(r),))


def physical_constraints(x, cfg: OracleConfig):
    """planner.py:472-497 (no caller in the reference), batched -> (...,4N):
    N velocity, N acceleration, then N pairs (thrust max, thrust min) interleaved."""
    N = cfg.prediction_horizon
    _, V, T = unpack(x, N)
    e3 = np.array([0.0, 0.0, 1.0])
    A = T / cfg.mass - cfg.gravity * e3
    vel = cfg.max_velocity ** 2 - np.sum(V ** 2, axis=-1)
    acc = cfg.max_acceleration ** 2 - np.sum(A ** 2, axis=-1)
    t2 = np.sum(T ** 2, axis=-1)
    thr = np.stack([cfg.max_thrust ** 2 - t2, t2 - cfg.min_thrust ** 2], axis=-1).reshape(t2.shape[:-1] + (2 * N,))
    return np.concatenate([vel, acc, thr], axis=-1)


# --------------------------------------------------------------------------- a11 / a12
def attitudes_and_rates(T, cfg: OracleConfig):
    """planner.py:604-654 for ONE problem: T (N,3) -> attitudes (N,3), body_rates (N,3).
    Sequential because of prev_R (rows with |T|<=1e-6 are skipped and keep prev_R)."""
    T = np.asarray(T, float)
    N = len(T)
    att = np.zeros((N, 3)); rates = np.zeros((N, 3))
    prev_R = None
    for i in range(N):
        tv = T[i]
        mag = np.linalg.norm(tv)
        if mag > 1e-6:
            b3 = tv / mag
            yaw_vec = np.array([1.0, 0.0, 0.0])          # desired_yaw = 0 (planner.py:623-625)
            b1 = np.cross(yaw_vec, b3)
            n1 = np.linalg.norm(b1)
            b1 = b1 / n1 if n1 > 1e-6 else np.array([1.0, 0.0, 0.0])
            b2 = np.cross(b3, b1)
            R = np.column_stack([b1, b2, b3])
            att[i] = (np.arctan2(R[2, 1], R[2, 2]), np.arcsin(-R[2, 0]), np.arctan2(R[1, 0], R[0, 0]))
            if prev_R is not None:
                om = R.T @ ((R - prev_R) / cfg.dt)
                rates[i] = (om[2, 1], om[0, 2], om[1, 0])
            prev_R = R
    return att, rates


def extract_solution(x, cfg: OracleConfig) -> Dict[str, np.ndarray]:
    """planner.py:582-602 for one problem."""
    N = cfg.prediction_horizon
    P, V, T = unpack(x, N)
    acc = T / cfg.mass - np.array([0, 0, cfg.gravity])
    att, rates = attitudes_and_rates(T, cfg)
    return dict(positions=P, velocities=V, thrust_vectors=T, accelerations=acc, attitudes=att,
                body_rates=rates, thrusts=np.linalg.norm(T, axis=1))


def extract_solution_batch(X, cfg: OracleConfig) -> Dict[str, np.ndarray]:
    """Batched extraction (loops over problems; the recurrence on prev_R is per problem)."""
    outs = [extract_solution(x, cfg) for x in np.atleast_2d(X)]
    return {k: np.stack([o[k] for o in outs]) for k in outs[0]}


# --------------------------------------------------------------------------- a7
def solve(p0, v0, goal, cfg: OracleConfig, x0: Optional[np.ndarray] = None):
    """planner.py:230-280 for one problem: cold start, box, SciPy L-BFGS-B with the
    reference's options.  Returns (x, info)."""
    from scipy.optimize import minimize
    if x0 is None:
        x0 = straight_line_init(p0, v0, goal, cfg)
    b = bounds(cfg)
    g_ = None if goal is None else np.asarray(goal, float)
    res = minimize(fun=lambda x: objective_ordered(x, g_, cfg), x0=x0, method="L-BFGS-B",
                   jac=lambda x: gradient(x, g_, cfg), bounds=[(lo, hi) for lo, hi in b],
                   options={"maxiter": cfg.max_iterations, "gtol": cfg.convergence_tolerance,
                            "ftol": cfg.convergence_tolerance * 10, "disp": False})
    return np.asarray(res.x, float), dict(nit=int(res.nit), nfev=int(res.nfev), status=int(res.status),
                                          fun=float(res.fun), success=bool(res.success), message=str(res.message))


def plan(p0, v0, goal, cfg: OracleConfig) -> Tuple[Dict[str, np.ndarray], dict]:
    """plan_trajectory minus the wall clock (planner.py:215-228, 656-675): the 8 Trajectory
    arrays with timestamps relative to start_time."""
    x, info = solve(p0, v0, goal, cfg)
    sol = extract_solution(x, cfg)
    N = cfg.prediction_horizon
    tr = dict(timestamps_rel=np.arange(N) * cfg.dt, positions=sol["positions"], velocities=sol["velocities"],
              accelerations=sol["accelerations"], attitudes=sol["attitudes"], body_rates=sol["body_rates"],
              thrusts=sol["thrusts"], yaws=sol["attitudes"][:, 2], yaw_rates=sol["body_rates"][:, 2], x=x)
    return tr, info


def plan_reference_shaped(p0, v0, goal, cfg: OracleConfig):
    """Same result as :func:`plan`, with the reference's per-k Python loops in f and g --
    the 'reference-shaped, 1 core' CPU baseline of BASELINE.md section 3."""
    from scipy.optimize import minimize
    g_ = np.asarray(goal, float)
    x0 = straight_line_init(p0, v0, g_, cfg)
    b = bounds(cfg)
    res = minimize(fun=lambda x: objective_loops(x, g_, cfg), x0=x0, method="L-BFGS-B",
                   jac=lambda x: gradient_loops(x, g_, cfg), bounds=[(lo, hi) for lo, hi in b],
                   options={"maxiter": cfg.max_iterations, "gtol": cfg.convergence_tolerance,
                            "ftol": cfg.convergence_tolerance * 10, "disp": False})
    return extract_solution(res.x, cfg)


# --------------------------------------------------------------------------- a15 / a16
def emergency_trajectory(p0, t0, cfg: OracleConfig):
    """planner.py:677-694."""
    N = cfg.prediction_horizon
    return dict(timestamps=t0 + np.arange(N) * cfg.dt, positions=np.tile(np.asarray(p0, float), (N, 1)),
                velocities=np.zeros((N, 3)), accelerations=np.zeros((N, 3)))


def is_plan_valid(positions, velocities=None) -> bool:
    """planner.py:717-737."""
    if positions is None or len(positions) == 0:
        return False
    P = np.asarray(positions, float)
    if np.any(np.isnan(P)) or np.any(np.isinf(P)):
        return False
    if np.any(P[:, 2] < 0.1):
        return False
    if velocities is not None and np.any(np.abs(np.asarray(velocities, float)) > 20.0):
        return False
    return True


# --------------------------------------------------------------------------- shooting form
# The reference has no forward rollout; planner.py:426-462 (a8) DEFINES one: the trajectory
# whose dynamics residual is identically zero.  This is the build's "canonical rollout"
# (SURVEY.md section 8d): states from (p0, v0, T), cost = a5 on the rolled-out decision vector,
# gradient = d cost / d T by the reverse (adjoint) sweep.  The tests pin it through a8
# (residual == 0), a5 (same cost) and central finite differences of a5.
def rollout(p0, v0, T, cfg: OracleConfig):
    """Forward rollout, batched: p0,v0 (...,3), T (...,N,3) -> P,V (...,N,3)."""
    N, dt = cfg.prediction_horizon, cfg.dt
    T = np.asarray(T, float)
    e3 = np.array([0.0, 0.0, 1.0])
    A = T / cfg.mass - cfg.gravity * e3
    P = np.zeros_like(T); V = np.zeros_like(T)
    P[..., 0, :] = p0; V[..., 0, :] = v0
    for k in range(N - 1):
        P[..., k + 1, :] = P[..., k, :] + V[..., k, :] * dt + 0.5 * A[..., k, :] * dt ** 2
        V[..., k + 1, :] = V[..., k, :] + A[..., k, :] * dt
    return P, V


def rollout_cost(p0, v0, goal, T, cfg: OracleConfig):
    P, V = rollout(p0, v0, T, cfg)
    return objective(pack(P, V, T), goal, cfg)


def rollout_cost_grad(p0, v0, goal, T, cfg: OracleConfig):
    """Cost and exact gradient wrt T by the adjoint sweep; batched."""
    N, dt, m = cfg.prediction_horizon, cfg.dt, cfg.mass
    T = np.asarray(T, float)
    goal = np.asarray(goal, float)
    P, V = rollout(p0, v0, T, cfg)
    cost = objective(pack(P, V, T), goal, cfg)
    e3 = np.array([0.0, 0.0, 1.0])
    A = T / m - cfg.gravity * e3
    wp, wv, wa, wT = cfg.position_weight, cfg.velocity_weight, cfg.acceleration_weight, cfg.thrust_weight
    G = 2 * wa * A / m + 2 * wT * (T - cfg.hover_thrust * e3)
    lamP = 2 * wp * 11.0 * (P[..., N - 1, :] - goal)
    lamV = 2 * wv * V[..., N - 1, :]
    for k in range(N - 2, -1, -1):
        G[..., k, :] += (0.5 * dt * dt * lamP + dt * lamV) / m
        lamV = 2 * wv * V[..., k, :] + dt * lamP + lamV
        lamP = 2 * wp * (P[..., k, :] - goal) + lamP
    return cost, G


def obstacle_penalty_grad(p0, v0, T, spheres, cfg: OracleConfig, obstacle_weight: float):
    """The build's obstacle-aware shooting objective (its EXTENSION: the reference builds the sphere residuals
    c_kj = |P_k - c_j|^2 - (r_j + safety_margin)^2 of planner.py:499-514 and then drops them, :250 vs :256-268; obstacle_weight,
    :63, is never read):  penalty = obstacle_weight * sum_k sum_j max(0, -c_kj)^2  on the ROLLED-OUT positions, and its exact
    gradient wrt the thrust sequence.  The gradient is written in closed form -- dP_k/dT_j = dt^2/m (k - j - 1/2) for j < k for
    this double integrator -- i.e. not with the adjoint recurrences the kernel runs, so the two check each other.
    spheres: (K, 4) rows (cx, cy, cz, r).  Batched over leading axes.  Returns (penalty (...), dpenalty/dT (..., N, 3))."""
    N, dt, m = cfg.prediction_horizon, cfg.dt, cfg.mass
    T = np.asarray(T, float)
    P, _ = rollout(p0, v0, T, cfg)
    sp = np.asarray(spheres, float).reshape(-1, 4)
    if sp.shape[0] == 0:
        return np.zeros(T.shape[:-2]), np.zeros_like(T)
    d = P[..., :, None, :] - sp[:, :3]                                   # (..., N, K, 3)
    c = np.sum(d * d, axis=-1) - (sp[:, 3] + cfg.safety_margin) ** 2      # (..., N, K)   planner.py:505-512
    h = np.maximum(0.0, -c)
    pen = obstacle_weight * np.sum(h * h, axis=(-1, -2))
    q = -4.0 * obstacle_weight * np.sum(h[..., None] * d, axis=-2)        # dpen/dP_k   (..., N, 3)
    kk, jj = np.arange(N)[:, None], np.arange(N)[None, :]
    J = np.where(kk > jj, dt * dt / m * (kk - jj - 0.5), 0.0)             # dP_k/dT_j, (k, j)
    return pen, np.einsum("kj,...ka->...ja", J, q)


# --------------------------------------------------------------------------- f-2: obstacle source
def local_grid_positions(centre, size: float, resolution: float) -> np.ndarray:
    """Grid cell centres in the order of ExplicitGeometricMapper.get_local_occupancy_grid
    (perception/explicit_geometric_mapper.py:221-248), flattened to (M, 3)."""
    centre = np.asarray(centre, float)
    half = size / 2
    n = int(size / resolution)
    x, y, z = (np.linspace(centre[a] - half, centre[a] + half, n) for a in range(3))
    return np.array(np.meshgrid(x, y, z)).T.reshape(-1, 3)


def spheres_from_grid(grid_positions, occupancy, threshold: float = 0.6, target: int = 20, radius: float = 1.0) -> np.ndarray:
    """cloud/main_improved_threelayer.py:387-398 (target 20) / tests/test_se3_mpc_with_mapper.py:29-33
    (target 10): every step-th occupied cell becomes a sphere of fixed radius.  -> (K, 4)."""
    pts = np.asarray(grid_positions, float).reshape(-1, 3)
    occ = np.asarray(occupancy, float).reshape(-1)
    occupied = pts[occ > threshold]
    if occupied.size == 0:
        return np.zeros((0, 4))
    step = max(1, occupied.shape[0] // target)
    chosen = occupied[::step]
    return np.concatenate([chosen, np.full((len(chosen), 1), radius)], axis=1)
