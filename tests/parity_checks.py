"""Parity checks shared by the CPU suite (product kernels compiled for the host, tests/emu) and the
GPU suite (libse3mpc.so on an MI355X).  Every check drives the C ABI through
dart_planner_amd.ops.Ops and compares with the oracle (oracle/se3mpc_oracle.py) or with the golden
vectors produced by the reference itself (tests/golden).

Tolerances (stated once, used everywhere):
  f64 kernels:  1e-9 absolute/relative on every output (they differ from the reference only by
                summation order / FMA contraction) -- except the solver's thrust block and what is
                derived from it (accelerations, thrust magnitudes): 1e-7.  Reason: while the L-BFGS
                memory is empty the solver takes the generalized Cauchy point in closed form
                (t = 1/theta exactly), whereas SciPy accumulates f1, f2 over ~170 breakpoints; when almost
                all of sum(g^2) sits on variables that hit their bounds (starts outside the box: position
                gradients ~1e4, thrust gradients ~3) SciPy's dtm = -f1/f2 carries ~1e-9 relative
                cancellation noise, visible as 1e-8 N on the free thrust entries (golden cases s26, s29).
  f32 kernels:  positions (and every other trajectory array except body rates) <= 1e-4 absolute --
                the bound BASELINE.json's north_star states ("trajectory position error <= 1e-4 m");
                costs 5e-6 relative; gradients / residuals 5e-6 of the array's max magnitude;
                body rates 5e-2 absolute (they are differences of O(1) rotation entries divided by
                dt = 1/400 s: float32 resolution 6e-8 * 400 * a few terms).
"""
import numpy as np

from dart_planner_amd.capi import Params
from oracle import se3mpc_oracle as orc

F32 = dict(pos=1e-4, cost_rel=5e-6, vec_rel=5e-6, rates=5e-2, thrust=1e-4)
F64 = dict(pos=1e-9, cost_rel=1e-12, vec_rel=1e-12, rates=1e-7, thrust=1e-7)


class Harness:
    """Backend adaptor: host ndarray <-> backend array."""

    def __init__(self, ops, to_dev, to_host, np_dtype):
        self.ops, self.to_dev, self.to_host, self.dt = ops, to_dev, to_host, np_dtype
        self.tol = F32 if np_dtype == np.float32 else F64

    def lane(self, a, B):
        """(B, ...) host array -> lane layout [rows][B] on the backend."""
        return self.to_dev(np.ascontiguousarray(np.asarray(a, float).reshape(B, -1).T.astype(self.dt)))

    def prob(self, a):
        return self.to_dev(np.ascontiguousarray(np.asarray(a, float).astype(self.dt)))

    def unlane(self, a, shape):
        return self.to_host(a).T.reshape(shape).astype(float)


def oracle_cfg(prm: Params) -> orc.OracleConfig:
    """The oracle configuration that states the same problem as `prm`.  The reference couples ftol to gtol
    (planner.py:264-265: ftol = 10 * convergence_tolerance), so only such pairs can be expressed."""
    assert abs(prm.ftol - 10 * prm.pgtol) <= 1e-15 * max(1.0, abs(prm.ftol)), "the reference solves with ftol = 10 * gtol"
    return orc.OracleConfig(prediction_horizon=prm.horizon, dt=prm.dt, max_iterations=prm.max_iterations,
                            convergence_tolerance=prm.pgtol, max_velocity=prm.max_velocity, max_acceleration=prm.max_acceleration,
                            max_thrust=prm.max_thrust, min_thrust=prm.min_thrust, max_tilt_angle=prm.max_tilt_angle,
                            position_weight=prm.position_weight, velocity_weight=prm.velocity_weight,
                            acceleration_weight=prm.acceleration_weight, thrust_weight=prm.thrust_weight,
                            safety_margin=prm.safety_margin, mass=prm.mass, gravity=prm.gravity, position_bound=prm.position_bound)


def vec_close(a, b, rel, what=""):
    scale = max(1.0, float(np.max(np.abs(b)))) if np.size(b) else 1.0
    err = float(np.max(np.abs(np.asarray(a, float) - b))) if np.size(b) else 0.0
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} > {rel:.1e} * {scale:.3e}"


def random_batch(rng, B, N, spread=2.0):
    cfg = orc.OracleConfig(prediction_horizon=N)
    p0, v0, goal = rng.uniform(-20, 20, (B, 3)), rng.uniform(-5, 5, (B, 3)), rng.uniform(-20, 20, (B, 3))
    T = rng.normal(0, spread, (B, N, 3)) + [0, 0, cfg.hover_thrust]
    return p0, v0, goal, T


def check_wide_kernels(h: Harness, N: int, B: int, seed: int = 0, K: int = 7):
    """The 16-byte-per-lane forms of the write-heavy float32 lane kernels (four trajectories per lane; taken by default from 262144
    trajectories up, forced here with se3mpc_set_rollout_variant(+1024)) give the SAME BITS as the dword forms (+512), and fall back to them
    when the batch is not a multiple of 4."""
    assert h.dt == np.float32 and B % 4 == 0
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N)
    p0, v0, goal, T = random_batch(rng, B, N)
    X = np.concatenate([rng.uniform(-30, 30, (B, 3 * N)), rng.uniform(-15, 15, (B, 3 * N)), T.reshape(B, -1)], axis=1)
    S = np.concatenate([rng.uniform(-10, 10, (K, 3)), rng.uniform(0.3, 2.0, (K, 1))], axis=1)
    lib = h.ops.lib

    def run(Bx):
        sl = slice(0, Bx)
        a = [h.to_host(h.ops.init(prm, h.lane(p0[sl], Bx), h.lane(v0[sl], Bx), h.lane(goal[sl], Bx), project=pj)) for pj in (False, True)]
        a += [h.to_host(x) for x in h.ops.obstacle_residual(prm, h.lane(X[sl], Bx), h.to_dev(S.astype(h.dt)))]
        return a
    try:
        lib.set_rollout_variant(512)
        narrow = run(B)
        lib.set_rollout_variant(1024)
        wide = run(B)
        odd = run(B - 1)                                 # not a multiple of 4: the dword kernels, whatever the switch says
        lib.set_rollout_variant(512)
        odd_ref = run(B - 1)
    finally:
        lib.set_rollout_variant(0)
    for w, n in zip(wide, narrow):
        assert np.array_equal(w, n)
    for w, n in zip(odd, odd_ref):
        assert np.array_equal(w, n)
    # a batch shorter than the leading dimension (B < ld, both multiples of 4): columns B.. of the outputs are not written
    ld, Bs = B, B - 8
    try:
        outs = []
        for var in (512, 1024):
            lib.set_rollout_variant(var)
            pl, vl, gl = h.lane(p0, ld), h.lane(v0, ld), h.lane(goal, ld)
            X0 = h.ops.init(prm, pl, vl, gl, B=Bs)
            Cm, cm, vi = h.ops.obstacle_residual(prm, h.lane(X, ld), h.to_dev(S.astype(h.dt)), B=Bs)
            outs.append([h.to_host(a)[..., :Bs] for a in (X0, Cm, cm, vi)])
    finally:
        lib.set_rollout_variant(0)
    for w, n in zip(outs[1], outs[0]):
        assert np.array_equal(w, n)


# --------------------------------------------------------------------------------------- lane kernels
def check_lane_kernels(h: Harness, N: int, B: int, seed: int = 0, variants=(0,), dt=None):
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N)
    if dt is not None:
        prm.dt = dt
    cfg = oracle_cfg(prm)
    t = h.tol
    p0, v0, goal, T = random_batch(rng, B, N)
    # a3 / a4
    X0 = h.unlane(h.ops.init(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B)), (B, 9 * N))
    ref0 = orc.straight_line_init(p0, v0, goal, cfg)
    vec_close(X0, ref0, t["vec_rel"] * 10, "init")
    X0p = h.unlane(h.ops.init(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), project=True), (B, 9 * N))
    b = orc.bounds(cfg)
    vec_close(X0p, np.clip(ref0, b[:, 0], b[:, 1]), t["vec_rel"] * 10, "init+project")
    # a5 / a6 on vectors inside and outside the box
    X = np.concatenate([rng.uniform(-120, 120, (B, 3 * N)), rng.uniform(-15, 15, (B, 3 * N)), T.reshape(B, -1)], axis=1)
    f, g = h.ops.cost_grad(prm, h.lane(X, B), h.lane(goal, B))
    fr = orc.objective(X, goal, cfg)
    assert np.max(np.abs(h.to_host(f) - fr) / np.abs(fr)) <= t["cost_rel"], "objective"
    vec_close(h.unlane(g, (B, 9 * N)), orc.gradient(X, goal, cfg), t["vec_rel"], "gradient")
    # goal_position None (planner.py:524, :546, :567)
    prm_ng = prm.copy(has_goal=0)
    f2, g2 = h.ops.cost_grad(prm_ng, h.lane(X, B), None)
    fr2 = orc.objective(X, None, cfg)
    assert np.max(np.abs(h.to_host(f2) - fr2) / np.abs(fr2)) <= t["cost_rel"], "objective(no goal)"
    vec_close(h.unlane(g2, (B, 9 * N)), orc.gradient(X, None, cfg), t["vec_rel"], "gradient(no goal)")
    # a8
    R = h.ops.dynamics_residual(prm, h.lane(X, B), h.lane(p0, B), h.lane(v0, B))
    vec_close(h.unlane(R, (B, 6 * N)), orc.dynamics_residual(X, p0, v0, cfg), t["vec_rel"], "dynamics residual")
    # a9 (K = 16 as config 3, and the empty table)
    sph = np.concatenate([np.round(rng.uniform(0, 15, (16, 3)) * 2) / 2, np.ones((16, 1))], axis=1)
    Cm, cmin, viol = h.ops.obstacle_residual(prm, h.lane(X, B), h.prob(sph))
    Cref = orc.obstacle_residual(X, sph[:, :3], sph[:, 3], cfg)
    vec_close(h.unlane(Cm, (B, N * 16)), Cref, t["vec_rel"], "obstacle residual")
    vec_close(h.to_host(cmin), Cref.min(1), t["vec_rel"], "obstacle min")
    vec_close(h.to_host(viol), np.maximum(0, -Cref).sum(1), t["vec_rel"] * 4, "obstacle violation")
    # reduced in-kernel (no residuals written): its own kernel; odd sphere counts exercise the padding row
    for Ks in (16, 7, 1):
        _, cmin_r, viol_r = h.ops.obstacle_residual(prm, h.lane(X, B), h.prob(sph[:Ks]), materialize=False)
        Cr_k = Cref.reshape(B, N, 16)[:, :, :Ks].reshape(B, -1)
        vec_close(h.to_host(cmin_r), Cr_k.min(1), t["vec_rel"], f"obstacle min (reduced, K={Ks})")
        vec_close(h.to_host(viol_r), np.maximum(0, -Cr_k).sum(1), t["vec_rel"] * 4, f"obstacle violation (reduced, K={Ks})")
    _, cmin_e, viol_e = h.ops.obstacle_residual(prm, h.lane(X, B), h.prob(np.zeros((0, 4))), materialize=False)
    assert np.all(np.isinf(h.to_host(cmin_e))) and np.all(h.to_host(viol_e) == 0)
    C0, cmin0, viol0 = h.ops.obstacle_residual(prm, h.lane(X, B), h.prob(np.zeros((0, 4))))
    assert tuple(C0.shape) == (0, B) and np.all(np.isinf(h.to_host(cmin0))) and np.all(h.to_host(viol0) == 0)
    # a10
    Cp = h.ops.physical_constraints(prm, h.lane(X, B))
    vec_close(h.unlane(Cp, (B, 4 * N)), orc.physical_constraints(X, cfg), t["vec_rel"], "physical constraints")
    # a11 / a12 incl. degenerate thrust rows
    Te = T.copy()
    if N >= 6:
        Te[0, 1] = 0.0; Te[0, 3] = [7.0, 0, 0]; Te[1, 0] = 0.0; Te[1, 2] = [-3.0, 0, 0]; Te[2, 4] = [0, 5.0, 0]
        Te[3, 2] = [1e-7, 0, 0]
    acc, att, rates, thr = h.ops.extract(prm, h.lane(Te, B))
    Xe = X.copy(); Xe[:, 6 * N:] = Te.reshape(B, -1)
    ex = orc.extract_solution_batch(Xe, cfg)
    vec_close(h.unlane(acc, (B, N, 3)), ex["accelerations"], t["vec_rel"], "accelerations")
    assert np.max(np.abs(h.unlane(att, (B, N, 3)) - ex["attitudes"])) <= t["pos"], "attitudes"
    assert np.max(np.abs(h.unlane(rates, (B, N, 3)) - ex["body_rates"])) <= t["rates"], "body rates"
    vec_close(h.unlane(thr, (B, N)), ex["thrusts"], t["vec_rel"], "thrusts")
    # shooting form, every requested variant, with the fused argmin key
    c_ref, g_ref = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
    P_ref, V_ref = orc.rollout(p0, v0, T, cfg)
    for var in variants:
        h.ops.lib.set_rollout_variant(var)
        try:
            key = h.to_dev(np.array([-1], dtype=np.int64))
            cost, gT, P, V = h.ops.rollout_cost_grad(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B),
                                                     want_states=True, key=key, index_base=7000)
            ch = h.to_host(cost).astype(float)
            assert np.max(np.abs(ch - c_ref) / np.abs(c_ref)) <= t["cost_rel"], f"rollout cost (variant {var})"
            vec_close(h.unlane(gT, (B, N, 3)), g_ref, t["vec_rel"], f"rollout gradient (variant {var})")
            assert np.max(np.abs(h.unlane(P, (B, N, 3)) - P_ref)) <= t["pos"], f"rollout positions (variant {var})"
            assert np.max(np.abs(h.unlane(V, (B, N, 3)) - V_ref)) <= t["pos"], f"rollout velocities (variant {var})"
            idx, kc = h.ops.decode_key(key)
            assert idx == 7000 + int(np.argmin(h.to_host(cost))), f"fused argmin (variant {var})"
            assert kc == np.float32(h.to_host(cost).min())
            c_only, g_none, _, _ = h.ops.rollout_cost_grad(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B),
                                                           want_grad=False)
            # (another template instantiation: same arithmetic, FMA contraction may differ in the last bit)
            assert g_none is None and np.allclose(h.to_host(c_only), h.to_host(cost), rtol=t["cost_rel"], atol=0)
        finally:
            h.ops.lib.set_rollout_variant(0)
    # multi-batch launch == the same batches launched one by one (incl. per-batch keys)
    nb = 3
    rngb = np.random.default_rng(seed + 100)
    bp0, bv0, bgoal, bT = (np.stack(x) for x in zip(*[random_batch(rngb, B, N) for _ in range(nb)]))
    st = lambda a, rows: h.to_dev(np.ascontiguousarray(np.transpose(a.reshape(nb, B, rows), (0, 2, 1)).astype(h.dt)))
    dp0, dv0, dgoal, dT = st(bp0, 3), st(bv0, 3), st(bgoal, 3), st(bT, 3 * N)
    costb = h.to_dev(np.zeros((nb, B), dtype=h.dt)); gradb = h.to_dev(np.zeros((nb, 3 * N, B), dtype=h.dt))
    wkb = h.to_dev(np.zeros((nb, (B + 63) // 64), dtype=np.int64))
    keysb = h.to_dev(np.zeros(nb, dtype=np.int64))
    h.ops.rollout_cost_grad_batched(prm, dp0, dv0, dgoal, dT, costb, gradb, wave_keys=wkb, index_base=11)
    h.ops.reduce_keys(wkb, keysb)
    for i in range(nb):
        ci, gi, _, _ = h.ops.rollout_cost_grad(prm, h.lane(bp0[i], B), h.lane(bv0[i], B), h.lane(bgoal[i], B), h.lane(bT[i], B))
        assert np.array_equal(h.to_host(costb)[i], h.to_host(ci)) and np.array_equal(h.to_host(gradb)[i], h.to_host(gi))
        ki = int(h.to_host(keysb)[i]) & 0xFFFFFFFFFFFFFFFF
        assert h.ops.lib.key_index(ki) == 11 + int(np.argmin(h.to_host(ci)))
    # rollout fused with the obstacle residuals of the rolled-out positions (config 3) == rollout, then a9 on its states
    # (the workgroup shapes: +128 = the 3 axis wavefronts do the evaluations, +256 / +384 = 5 / 1 helper wavefronts start on the position tile
    #  during the adjoint sweep; sphere counts that are not a multiple of the register chunk of 8, and few enough that the helpers take every step)
    Cr_all = orc.obstacle_residual(orc.pack(P_ref, V_ref, T), sph[:, :3], sph[:, 3], cfg).reshape(B, N, -1)
    try:
        for wsel, Ks in ((128, len(sph)), (256, len(sph)), (384, len(sph)), (128, 3), (256, 9), (384, 9), (384, 1), (256, 2), (0, len(sph)),
                         (2048 + 128, len(sph)), (2048 + 256, 9), (2048 + 384, len(sph))):     # + 2048: float32 residuals on the matrix core (expanded form), not the packed-VALU difference form
            h.ops.lib.set_rollout_variant(wsel)
            co, go, cmn, vio = h.ops.rollout_obstacles(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B),
                                                       h.prob(sph[:Ks]))
            assert np.max(np.abs(h.to_host(co).astype(float) - c_ref) / np.abs(c_ref)) <= t["cost_rel"], "fused rollout cost"
            vec_close(h.unlane(go, (B, N, 3)), g_ref, t["vec_rel"], "fused rollout gradient")
            Cr = Cr_all[:, :, :Ks].reshape(B, -1)
            vec_close(h.to_host(cmn), Cr.min(1), t["vec_rel"] * 4, f"fused obstacle min (variant {wsel}, K={Ks})")
            vec_close(h.to_host(vio), np.maximum(0, -Cr).sum(1), t["vec_rel"] * 8, f"fused obstacle violation (variant {wsel}, K={Ks})")
        # 40 spheres: more than the helper wavefronts' head start covers, so that both shapes also split steps over all wavefronts
        sph40 = np.concatenate([sph, sph + np.array([0.25, -0.25, 0.5, 0.0]), sph[:8] - np.array([0.5, 0.5, 0.25, 0.0])])
        Cr40 = orc.obstacle_residual(orc.pack(P_ref, V_ref, T), sph40[:, :3], sph40[:, 3], cfg).reshape(B, -1)
        for wsel in (128, 256, 384, 2048 + 256, 2048 + 384):
            h.ops.lib.set_rollout_variant(wsel)
            _, _, cmn, vio = h.ops.rollout_obstacles(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B), h.prob(sph40))
            vec_close(h.to_host(cmn), Cr40.min(1), t["vec_rel"] * 4, f"fused obstacle min (variant {wsel}, K=40)")
            vec_close(h.to_host(vio), np.maximum(0, -Cr40).sum(1), t["vec_rel"] * 8, f"fused obstacle violation (variant {wsel}, K=40)")
        # the largest table the entry point takes (SE3MPC_MAX_SPHERES = 256): sixteen shifted copies of the scene
        sph256 = np.concatenate([sph + np.array([0.5 * i, -0.25 * i, 0.125 * i, 0.0]) for i in range(16)])
        Cr256 = orc.obstacle_residual(orc.pack(P_ref, V_ref, T), sph256[:, :3], sph256[:, 3], cfg).reshape(B, -1)
        for wsel in (0, 384, 2048 + 384):
            h.ops.lib.set_rollout_variant(wsel)
            _, _, cmn, vio = h.ops.rollout_obstacles(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B), h.prob(sph256))
            vec_close(h.to_host(cmn), Cr256.min(1), t["vec_rel"] * 4, f"fused obstacle min (variant {wsel}, K=256)")
            vec_close(h.to_host(vio), np.maximum(0, -Cr256).sum(1), t["vec_rel"] * 8, f"fused obstacle violation (variant {wsel}, K=256)")
    finally:
        h.ops.lib.set_rollout_variant(0)
    co2, g_none, cmn2, _ = h.ops.rollout_obstacles(prm, h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B),
                                                   h.prob(np.zeros((0, 4))), want_grad=False)
    assert g_none is None and np.all(np.isinf(h.to_host(cmn2))) and np.allclose(h.to_host(co2), h.to_host(co), rtol=t["cost_rel"])
    # multi-batch launch of the fused kernel == the same batches one by one (shared sphere table, per-batch keys)
    cob = h.to_dev(np.zeros((nb, B), dtype=h.dt)); gob = h.to_dev(np.zeros((nb, 3 * N, B), dtype=h.dt))
    cmb = h.to_dev(np.zeros((nb, B), dtype=h.dt)); vib = h.to_dev(np.zeros((nb, B), dtype=h.dt))
    wkb = h.to_dev(np.zeros((nb, (B + 63) // 64), dtype=np.int64))
    h.ops.rollout_obstacles_batched(prm, dp0, dv0, dgoal, dT, h.prob(sph[:9]), cob, gob, cmb, vib, wave_keys=wkb, index_base=3)
    h.ops.reduce_keys(wkb, keysb)
    for i in range(nb):
        ci, gi, mi, vi = h.ops.rollout_obstacles(prm, h.lane(bp0[i], B), h.lane(bv0[i], B), h.lane(bgoal[i], B), h.lane(bT[i], B),
                                                 h.prob(sph[:9]))
        for got, one in ((cob, ci), (gob, gi), (cmb, mi), (vib, vi)):
            assert np.array_equal(h.to_host(got)[i], h.to_host(one)), "batched fused rollout"
        assert h.ops.lib.key_index(int(h.to_host(keysb)[i]) & 0xFFFFFFFFFFFFFFFF) == 3 + int(np.argmin(h.to_host(ci)))
    # the rolled-out states satisfy the reference's dynamics constraints (a8 == 0)
    Xr = orc.pack(P_ref, V_ref, T)
    Rr = h.ops.dynamics_residual(prm, h.lane(Xr, B), h.lane(p0, B), h.lane(v0, B))
    assert np.max(np.abs(h.to_host(Rr))) <= (1e-4 if h.dt == np.float32 else 1e-12)
    # a16
    Pv = rng.uniform(0.2, 5, (B, N, 3)); Vv = rng.uniform(-19, 19, (B, N, 3))
    Pv[1, 0, 2] = 0.05; Vv[2, N - 1, 1] = -20.5
    if B > 4:
        Pv[3, N // 2, 0] = np.nan; Pv[4, 0, 1] = np.inf
    valid = h.to_host(h.ops.is_plan_valid(prm, h.lane(Pv, B), h.lane(Vv, B)))
    expect = np.array([orc.is_plan_valid(Pv[i], Vv[i]) for i in range(B)], dtype=np.int32)
    assert np.array_equal(valid, expect), "is_plan_valid"
    # standalone argmin + transpose
    key = h.ops.argmin(f, index_base=5)
    assert h.ops.decode_key(key)[0] == 5 + int(np.argmin(h.to_host(f)))
    Xt = h.ops.transpose(h.prob(X))
    assert np.array_equal(h.to_host(Xt), X.astype(h.dt).T)


def thrust_box(cfg):
    txy = cfg.max_thrust * np.sin(cfg.max_tilt_angle)
    return np.array([-txy, -txy, cfg.min_thrust]), np.array([txy, txy, cfg.max_thrust])


def oracle_iterate(p0, v0, goal, T, cfg, iters, step):
    """Host-chained float64 reference of se3mpc_rollout_iterate_*: `iters` x (rollout + cost + gradient, projected step), one last
    evaluation.  -> T_final, cost_final, grad_final, cost_first"""
    lo, hi = thrust_box(cfg)
    T = np.array(T, dtype=float)
    c0 = None
    for _ in range(iters):
        c, g = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
        c0 = c if c0 is None else c0
        T = np.clip(T - step * g, lo, hi)
    c, g = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
    return T, c, g, (c if c0 is None else c0)


def check_rollout_iterate(h: Harness, N: int, B: int, seed: int = 0, iters: int = 5, step: float = 0.9):
    """The on-device iteration loop: (1) against the host-chained oracle; (2) K iterations in one launch == K launches of one
    iteration, bit for bit (the thrust sequence crosses HBM between launches, stays in registers inside one); (3) == the chain
    se3mpc_rollout_cost_grad + se3mpc_projected_step it replaces; (4) iters = 0 == the plain rollout; (5) multi-batch launch ==
    batch by batch; (6) the fused argmin key; (7) the descent really descends."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N)
    cfg = oracle_cfg(prm)
    t = h.tol
    p0, v0, goal, T = random_batch(rng, B, N, spread=6.0)          # wide spread: some thrusts start outside the box
    r = lambda a: np.asarray(a).astype(h.dt).astype(float)
    Tr, cr, gr, c0r = oracle_iterate(r(p0), r(v0), r(goal), r(T), cfg, iters, step)
    lp0, lv0, lg, lT = h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B)
    key = h.to_dev(np.zeros((B + 63) // 64, dtype=np.int64))
    out = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, iters, step, want_first_cost=True, wave_keys=key, index_base=40)
    Tk = h.unlane(out["T"], (B, N, 3))
    scale = 50.0 if h.dt == np.float32 else 1.0                    # f32: `iters` descent steps amplify one-ulp differences of the gradient
    vec_close(Tk, Tr, t["vec_rel"] * scale, "iterated T")
    ch = h.to_host(out["cost"]).astype(float)
    assert np.max(np.abs(ch - cr) / np.abs(cr)) <= t["cost_rel"] * scale, "iterated cost"
    vec_close(h.unlane(out["gradT"], (B, N, 3)), gr, t["vec_rel"] * scale, "gradient at the final T")
    assert np.max(np.abs(h.to_host(out["cost_first"]).astype(float) - c0r) / np.abs(c0r)) <= t["cost_rel"], "cost at T_in"
    lo, hi = thrust_box(cfg)
    assert np.all(Tk >= lo.astype(h.dt).astype(float) - 0) and np.all(Tk <= hi.astype(h.dt).astype(float) + 0), "box"
    assert np.all(ch <= h.to_host(out["cost_first"]).astype(float) * (1 + 1e-6)), "descent"                      # (7)
    kd = h.to_dev(np.zeros(1, dtype=np.int64))
    h.ops.reduce_keys(key.reshape(1, -1), kd)
    idx, kc = h.ops.decode_key(kd)
    assert idx == 40 + int(np.argmin(h.to_host(out["cost"]))) and kc == np.float32(h.to_host(out["cost"]).min())   # (6)
    # (2) one launch == `iters` launches of one iteration
    Tc = lT
    for _ in range(iters):
        Tc = h.ops.rollout_iterate(prm, lp0, lv0, lg, Tc, 1, step, want_grad=False)["T"]
    last = h.ops.rollout_iterate(prm, lp0, lv0, lg, Tc, 0, step)
    assert np.array_equal(h.to_host(Tc), h.to_host(out["T"])), "K iterations in one launch != K one-iteration launches"
    assert np.array_equal(h.to_host(last["cost"]), h.to_host(out["cost"])) and np.array_equal(h.to_host(last["gradT"]), h.to_host(out["gradT"]))
    assert np.array_equal(h.to_host(last["T"]), h.to_host(Tc))
    # (3) the chain of the two stand-alone launches this entry point replaces
    Th = lT
    for _ in range(iters):
        _, g1, _, _ = h.ops.rollout_cost_grad(prm, lp0, lv0, lg, Th)
        Th = h.ops.projected_step(prm, Th, g1, step)
    c2, g2, _, _ = h.ops.rollout_cost_grad(prm, lp0, lv0, lg, Th)
    bitwise = np.array_equal(h.to_host(Th), h.to_host(out["T"]))
    vec_close(h.to_host(Th), h.to_host(out["T"]).astype(float), 2e-6 if h.dt == np.float32 else 1e-13, "host chain vs one launch")
    assert np.allclose(h.to_host(c2), h.to_host(out["cost"]), rtol=2e-6 if h.dt == np.float32 else 1e-13, atol=0)
    # (4) iters = 0
    o0 = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, 0, step)
    c1, g1, _, _ = h.ops.rollout_cost_grad(prm, lp0, lv0, lg, lT)
    assert np.array_equal(h.to_host(o0["T"]), h.to_host(lT))
    assert np.allclose(h.to_host(o0["cost"]), h.to_host(c1), rtol=t["cost_rel"], atol=0)
    vec_close(h.to_host(o0["gradT"]), h.to_host(g1).astype(float), t["vec_rel"], "iters = 0 gradient")
    # (5) multi-batch launch, in place (T_out = T)
    nb = 3
    rngb = np.random.default_rng(seed + 7)
    bp0, bv0, bgoal, bT = (np.stack(x) for x in zip(*[random_batch(rngb, B, N) for _ in range(nb)]))
    st = lambda a, rows: h.to_dev(np.ascontiguousarray(np.transpose(a.reshape(nb, B, rows), (0, 2, 1)).astype(h.dt)))
    dp0, dv0, dgoal, dT = st(bp0, 3), st(bv0, 3), st(bgoal, 3), st(bT, 3 * N)
    ob = h.ops.rollout_iterate(prm, dp0, dv0, dgoal, dT, iters, step, T_out=dT)
    for i in range(nb):
        oi = h.ops.rollout_iterate(prm, h.lane(bp0[i], B), h.lane(bv0[i], B), h.lane(bgoal[i], B), h.lane(bT[i], B), iters, step)
        for nm in ("T", "cost", "gradT"):
            assert np.array_equal(h.to_host(ob[nm])[i], h.to_host(oi[nm])), ("batched iterate", nm)
    return bitwise


def oracle_iterate_obstacles(p0, v0, goal, T, spheres, cfg, iters, step, w_obs):
    """Host-chained float64 reference of se3mpc_rollout_iterate_obstacles_*: running cost + obstacle penalty (oracle.obstacle_penalty_grad,
    closed-form chain rule), projected steps.  -> T_final, cost_final, grad_final, cost_first, penalty_final"""
    lo, hi = thrust_box(cfg)
    T = np.array(T, dtype=float)
    c0 = None
    for _ in range(iters):
        c, g = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
        pen, gp = orc.obstacle_penalty_grad(p0, v0, T, spheres, cfg, w_obs)
        c0 = (c + pen) if c0 is None else c0
        T = np.clip(T - step * (g + gp), lo, hi)
    c, g = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
    pen, gp = orc.obstacle_penalty_grad(p0, v0, T, spheres, cfg, w_obs)
    return T, c + pen, g + gp, ((c + pen) if c0 is None else c0), pen


def check_rollout_iterate_obstacles(h: Harness, N: int, B: int, seed: int = 0, iters: int = 4, K: int = 5, dt: float = 0.1,
                                    step: float = 2e-3, w_obs: float = 40.0):
    """The obstacle-aware iteration loop (the build's extension: running cost + w_obs * sum max(0, -c_kj)^2 on the rolled-out positions):
    (1) against the host-chained oracle whose penalty gradient is a closed-form chain rule, not an adjoint sweep; (2) K iterations in
    one launch == K one-iteration launches, bit for bit; (3) iters = 0: cost = rollout cost + penalty, the penalty output, cost_first;
    (4) no spheres == the plain loop; (5) the narrow (3 wavefronts x 64 trajectories, table in LDS) and the wide (7 x 32, table in registers) workgroup shapes give the same bits; (6) multi-batch == batch by batch;
    (7) spheres far away: zero penalty, the plain loop's numbers.  dt = 0.1 by default so that the horizon covers metres (at the
    reference's 1/400 s a 30-step plan spans 7 cm and nothing but a sphere on top of the start would matter)."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, dt=dt)
    cfg = oracle_cfg(prm)
    t = h.tol
    p0, v0, goal, T = random_batch(rng, B, N, spread=3.0)
    p0 = p0 * 0.1; goal = goal * 0.2                                   # starts within +-2 m, goals within +-4 m: the spheres are in the way
    sph = np.concatenate([rng.uniform(-3, 3, (K, 3)), rng.uniform(0.3, 1.0, (K, 1))], axis=1)
    r = lambda a: np.asarray(a).astype(h.dt).astype(float)
    Tr, cr, gr, c0r, penr = oracle_iterate_obstacles(r(p0), r(v0), r(goal), r(T), r(sph), cfg, iters, step, w_obs)
    assert np.any(penr > 0), "the check wants trajectories inside the spheres' margins"
    lp0, lv0, lg, lT = h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B)
    dsph = h.to_dev(np.ascontiguousarray(sph.astype(h.dt)))
    run = lambda Tin, it, **kw: h.ops.rollout_iterate(prm, lp0, lv0, lg, Tin, it, step, spheres=dsph, obstacle_weight=w_obs, **kw)
    out = run(lT, iters, want_first_cost=True)
    scale = 50.0 if h.dt == np.float32 else 1.0
    Tk = h.unlane(out["T"], (B, N, 3))
    vec_close(Tk, Tr, t["vec_rel"] * scale, "iterated T (obstacles)")
    ch = h.to_host(out["cost"]).astype(float)
    assert np.max(np.abs(ch - cr) / np.abs(cr)) <= t["cost_rel"] * scale, "iterated cost (obstacles)"
    vec_close(h.unlane(out["gradT"], (B, N, 3)), gr, t["vec_rel"] * scale, "gradient at the final T (obstacles)")
    assert np.max(np.abs(h.to_host(out["cost_first"]).astype(float) - c0r) / np.abs(c0r)) <= t["cost_rel"] * 4, "cost at T_in (obstacles)"
    ph = h.to_host(out["penalty"]).astype(float)
    assert np.max(np.abs(ph - penr)) <= t["cost_rel"] * scale * max(1.0, float(np.max(penr))), "penalty"
    # (2) one launch == `iters` launches of one iteration
    Tc = lT
    for _ in range(iters):
        Tc = run(Tc, 1, want_grad=False)["T"]
    last = run(Tc, 0)
    assert np.array_equal(h.to_host(Tc), h.to_host(out["T"])), "K iterations in one launch != K one-iteration launches (obstacles)"
    assert np.array_equal(h.to_host(last["cost"]), h.to_host(out["cost"])) and np.array_equal(h.to_host(last["gradT"]), h.to_host(out["gradT"]))
    # (3) iters = 0 against the oracle's two parts
    o0 = run(lT, 0, want_first_cost=True)
    c_run, g_run = orc.rollout_cost_grad(r(p0), r(v0), r(goal), r(T), cfg)
    pen0, gp0 = orc.obstacle_penalty_grad(r(p0), r(v0), r(T), r(sph), cfg, w_obs)
    assert np.allclose(h.to_host(o0["cost"]).astype(float), c_run + pen0, rtol=t["cost_rel"] * 4, atol=0)
    assert np.array_equal(h.to_host(o0["cost"]), h.to_host(o0["cost_first"])) and np.array_equal(h.to_host(o0["T"]), h.to_host(lT))
    vec_close(h.unlane(o0["gradT"], (B, N, 3)), g_run + gp0, t["vec_rel"] * 4, "iters = 0 gradient (obstacles)")
    # (4) no spheres, and (7) spheres out of reach: the plain loop
    plain = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, iters, step)
    none = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, iters, step, spheres=h.to_dev(np.zeros((0, 4), dtype=h.dt)), obstacle_weight=w_obs)
    far = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, iters, step, spheres=h.to_dev(np.array([[500.0, 500.0, 500.0, 1.0]], dtype=h.dt)), obstacle_weight=w_obs)
    for o in (none, far):
        assert np.all(h.to_host(o["penalty"]) == 0)
        vec_close(h.to_host(o["T"]).astype(float), h.to_host(plain["T"]).astype(float), 2e-6 if h.dt == np.float32 else 1e-13, "no obstacle in reach vs the plain loop")
        assert np.allclose(h.to_host(o["cost"]), h.to_host(plain["cost"]), rtol=2e-6 if h.dt == np.float32 else 1e-13, atol=0)
    # (5) narrow / wide workgroup shape
    res = {}
    for sel in (128, 256):
        h.ops.lib.set_rollout_variant(sel)
        try:
            res[sel] = run(lT, iters)
        finally:
            h.ops.lib.set_rollout_variant(0)
    for nm in ("T", "gradT"):
        assert np.array_equal(h.to_host(res[128][nm]), h.to_host(res[256][nm])), ("narrow vs wide workgroup shape", nm)
    assert np.allclose(h.to_host(res[128]["cost"]), h.to_host(res[256]["cost"]), rtol=t["cost_rel"], atol=0)      # (the penalty shares are summed in another order)
    # (6) multi-batch launch, in place
    nb = 2
    st = lambda a, rows: h.to_dev(np.ascontiguousarray(np.transpose(np.stack([a, a[::-1]]).reshape(nb, B, rows), (0, 2, 1)).astype(h.dt)))
    dp0, dv0, dgoal, dT = st(p0, 3), st(v0, 3), st(goal, 3), st(T.reshape(B, -1), 3 * N)
    ob = h.ops.rollout_iterate(prm, dp0, dv0, dgoal, dT, iters, step, T_out=dT, spheres=dsph, obstacle_weight=w_obs)
    for nm in ("T", "cost", "gradT", "penalty"):
        assert np.array_equal(h.to_host(ob[nm])[0], h.to_host(out[nm])), ("batched iterate (obstacles)", nm)
    return float(np.max(penr))


def check_iterate_keys(h: Harness, N: int, B: int, seed: int = 0, K: int = 5, iters: int = 2, index_base: int = 0):
    """The argmin keys fused into both iteration loops: min over the [ceil(B/64)] slots == (cost, index) of np.argmin over the launch's own costs (ties
    to the lowest index), for the plain loop, and for the obstacle-aware loop in its narrow shape (one workgroup per slot, plain stores) and its wide
    one (two workgroups of 32 trajectories fold into a slot with atomicMin; the launcher presets the slots), with slots pre-filled with garbage."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, dt=0.1)
    p0, v0, goal, T = random_batch(rng, B, N, spread=3.0)
    p0 = p0 * 0.1; goal = goal * 0.2
    sph = np.concatenate([rng.uniform(-3, 3, (K, 3)), rng.uniform(0.3, 1.0, (K, 1))], axis=1)
    lp0, lv0, lg, lT = h.lane(p0, B), h.lane(v0, B), h.lane(goal, B), h.lane(T, B)
    dsph = h.to_dev(np.ascontiguousarray(sph.astype(h.dt)))
    n_slots = (B + 63) // 64
    for name, kw, sel in (("plain", {}, 0), ("obstacles narrow", dict(spheres=dsph, obstacle_weight=40.0), 128), ("obstacles wide", dict(spheres=dsph, obstacle_weight=40.0), 0)):
        keys = h.to_dev(np.full(n_slots, 12345, dtype=np.int64))                 # garbage: every slot must be overwritten / preset by the launch
        h.ops.lib.set_rollout_variant(sel)
        try:
            out = h.ops.rollout_iterate(prm, lp0, lv0, lg, lT, iters, 2e-3, want_grad=False, wave_keys=keys, index_base=index_base, **kw)
        finally:
            h.ops.lib.set_rollout_variant(0)
        cost = h.to_host(out["cost"]).astype(np.float32)
        kh = h.to_host(keys).astype(np.int64).view(np.uint64)
        best = int(kh.min())
        idx, kc = h.ops.lib.key_index(best), h.ops.lib.key_cost(best)
        win = int(np.argmin(cost))
        assert idx - index_base == win and np.float32(kc) == cost[win], (name, idx, win, kc, cost[win])
        # every slot holds the minimum of its own 64 trajectories
        for sl in range(n_slots):
            lo, hi = 64 * sl, min(64 * sl + 64, B)
            w = lo + int(np.argmin(cost[lo:hi]))
            assert h.ops.lib.key_index(int(kh[sl])) - index_base == w, (name, sl)


def check_shooting_finish(h: Harness, N: int, B: int, seed: int = 0, K: int = 3, dt: float = 0.1):
    """se3mpc_shooting_finish_* (the tail of a shooting-form plan in one launch) against the chain it replaces AND the oracle: the winner is the
    argmin of the rollout's own keys; positions / velocities / cost = the float64 rollout of that thrust column (oracle, 1e-12 relative);
    accelerations, attitudes, body rates, thrust magnitudes = the oracle's extraction (planner.py:582-654) of it, zero-thrust and x-axis-thrust rows
    included; the penalty = the oracle's sphere penalty; has_goal = 0 ignores the goal."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, dt=dt)
    cfg = oracle_cfg(prm)
    p0, v0, goal, T = random_batch(rng, B, N, spread=3.0)
    p0, v0, goal = p0[0] * 0.1, v0[0], goal[0] * 0.2                  # ONE state: a plan's samples share it
    T = T.copy()
    if N >= 3 and B >= 2:
        T[:, 1] = 0.0                                                 # a zero-thrust step: skipped by the frame recurrence (planner.py:618)
        T[:, 2] = [3.0, 0.0, 0.0]                                     # thrust along x: the b1 fallback (planner.py:626-627)
    r = lambda a: np.asarray(a).astype(h.dt).astype(float)
    Tq = r(T)
    sph = np.concatenate([rng.uniform(-3, 3, (K, 3)), rng.uniform(0.3, 1.0, (K, 1))], axis=1)
    wide = lambda a: np.tile(np.asarray(a, float).reshape(1, 3), (B, 1))
    lane = lambda a: h.lane(a, B)
    n_slots = (B + 63) // 64
    for has_goal in (1, 0):
        prm_g = prm.copy(has_goal=has_goal)
        cfg_g = oracle_cfg(prm_g)
        keys = h.to_dev(np.zeros(n_slots, dtype=np.int64))
        cost, *_ = h.ops.rollout_cost_grad(prm_g, lane(wide(p0)), lane(wide(v0)), lane(wide(goal)), lane(T), want_grad=False, wave_keys=keys)
        ch = h.to_host(cost).astype(float)
        win = int(np.argmin(ch))
        state = h.to_dev(np.concatenate([p0, v0, goal]).astype(np.float64))
        out = h.to_dev(np.full(19 * N + 3, np.nan))
        key_out = h.to_dev(np.zeros(1, dtype=np.int64))
        dsph = h.to_dev(np.ascontiguousarray(sph, dtype=np.float64))
        h.ops.shooting_finish(prm_g, lane(T), keys, state, out, key_out=key_out, spheres=dsph, obstacle_weight=40.0)
        o = h.to_host(out).astype(float)
        assert int(h.ops.lib.key_index(int(h.to_host(key_out)[0]) & 0xFFFFFFFFFFFFFFFF)) == win
        blk = lambda i: o[3 * N * i:3 * N * (i + 1)].reshape(N, 3)
        assert np.array_equal(blk(2), Tq[win])
        P_ref, V_ref = orc.rollout(p0[None], v0[None], Tq[win][None], cfg_g)
        c_ref = orc.objective(orc.pack(P_ref, V_ref, Tq[win][None]), goal[None] if has_goal else None, cfg_g)
        vec_close(blk(0), P_ref[0], 1e-12, "finish: positions")
        vec_close(blk(1), V_ref[0], 1e-12, "finish: velocities")
        assert abs(o[19 * N] - c_ref[0]) <= 1e-12 * abs(c_ref[0]), "finish: cost"
        ex = orc.extract_solution(orc.pack(P_ref, V_ref, Tq[win][None])[0], cfg_g)
        vec_close(blk(3), ex["accelerations"], 1e-12, "finish: accelerations")
        vec_close(blk(4), ex["attitudes"], 1e-12, "finish: attitudes")
        vec_close(blk(5), ex["body_rates"], 1e-9, "finish: body rates")
        vec_close(o[18 * N:19 * N], ex["thrusts"], 1e-12, "finish: thrust magnitudes")
        pen_ref, _ = orc.obstacle_penalty_grad(p0[None], v0[None], Tq[win][None], sph, cfg_g, 40.0)
        assert abs(o[19 * N + 1] - pen_ref[0]) <= 1e-11 * max(1.0, pen_ref[0]), "finish: penalty"
        assert o[19 * N + 2] == o[19 * N] + o[19 * N + 1]
        # no spheres: penalty 0
        h.ops.shooting_finish(prm_g, lane(T), keys, state, out)
        o2 = h.to_host(out).astype(float)
        assert o2[19 * N + 1] == 0.0 and np.array_equal(o2[:19 * N + 1], o[:19 * N + 1])


def check_key_nonfinite(h: Harness):
    """Packed argmin keys with non-finite costs: NaN of either sign must never win (a negative NaN would sort below
    -inf in a plain sign-magnitude map) nor pass for the dead-lane sentinel; -inf wins over everything real; +inf
    loses to everything real.  Through se3mpc_argmin_*, the key fused into the rollout and se3mpc_reduce_keys."""
    B = 130
    nan_pos = np.float32(np.nan)
    nan_neg = np.frombuffer(np.uint32(0xFFC00001).tobytes(), dtype=np.float32)[0]
    nan_full = np.frombuffer(np.uint32(0x7FFFFFFF).tobytes(), dtype=np.float32)[0]
    for dt in (np.float32, np.float64):
        base = np.linspace(5.0, 9.0, B).astype(dt)
        for name, fill, expect in (("nan", [nan_pos, nan_neg, nan_full], 3), ("+inf", [np.inf], 1), ("-inf", [-np.inf], 0)):
            c = base.copy()
            c[:len(fill)] = np.asarray(fill, dtype=dt)
            c[64] = dt(4.0)                                    # the finite minimum sits in the second wavefront
            key = h.ops.argmin(h.to_dev(c), index_base=100)
            idx, kc = h.ops.decode_key(key)
            if name == "-inf":
                assert idx == 100 and kc == -np.inf, (name, dt, idx, kc)
            else:
                assert idx == 164 and kc == 4.0, (name, dt, idx, kc)
        # every cost NaN: the key is the NaN code with a REAL index (not the sentinel), so a caller can tell
        # "all diverged" from "no live lane"
        key = h.ops.argmin(h.to_dev(np.full(B, np.nan, dtype=dt)), index_base=0)
        k = int(h.to_host(key)[0]) & 0xFFFFFFFFFFFFFFFF
        assert (k >> 32) == 0xFFFFFFFE and (k & 0xFFFFFFFF) < B
    # fused into the rollout: a NaN goal makes trajectory 5's cost NaN; another trajectory must win
    N = 6
    prm = Params.reference_defaults(horizon=N)
    rng = np.random.default_rng(4)
    p0, v0, goal, T = random_batch(rng, B, N)
    goal[5, 1] = np.nan
    goal[70] = p0[70]; v0[70] = 0.0                            # cheap trajectory
    key = h.to_dev(np.array([-1], dtype=np.int64))
    hh = Harness(h.ops, h.to_dev, h.to_host, np.float32)
    cost, *_ = h.ops.rollout_cost_grad(prm, hh.lane(p0, B), hh.lane(v0, B), hh.lane(goal, B), hh.lane(T, B), want_grad=False, key=key)
    ch = h.to_host(cost)
    assert np.isnan(ch[5]) and np.isfinite(np.delete(ch, 5)).all()
    idx, kc = h.ops.decode_key(key)
    assert idx == int(np.nanargmin(ch)) and kc == np.nanmin(ch)


# --------------------------------------------------------------------------------------- solver
def solve_params(c) -> Params:
    return Params.reference_defaults(horizon=c["N"], dt=c["dt"], max_iterations=c["maxiter"], pgtol=c["tol"],
                                     ftol=10 * c["tol"])


def check_solver_golden(h: Harness, data, meta, keys=None, thrust_tol=None, group=None):
    """The batched solve against what the reference itself returned (tests/golden/solve_cases).
    group: force the solver's lanes-per-problem (cases whose horizon does not fit the group run at the smallest group that holds it)."""
    if group is not None:
        h.ops.lib.set_solver_variant(int(group) << 8)
        try:
            return check_solver_golden(h, data, meta, keys=keys, thrust_tol=thrust_tol)
        finally:
            h.ops.lib.set_solver_variant(0)
    t = dict(h.tol)
    if thrust_tol is not None:
        t["thrust"] = thrust_tol
    worst = 0.0
    for c in meta["cases"]:
        k = c["key"]
        if keys is not None and k not in keys:
            continue
        prm = solve_params(c)
        N = c["N"]
        out = h.ops.solve(prm, h.prob(data[k + "p0"][None]), h.prob(data[k + "v0"][None]), h.prob(data[k + "goal"][None]))
        info = h.ops.info_to_host(out["info"])[0]
        assert (int(info["nit"]), int(info["nfev"]), int(info["status"])) == (c["nit"], c["nfev"], c["status"]), (k, info)
        x = h.to_host(out["x"])[0].astype(float)
        assert np.max(np.abs(x[:6 * N] - data[k + "x"][:6 * N])) <= t["pos"], (k, "x[P,V]")
        assert np.max(np.abs(x[6 * N:] - data[k + "x"][6 * N:])) <= t["thrust"], (k, "x[T]")
        worst = max(worst, float(np.max(np.abs(x[:3 * N] - data[k + "positions"].ravel()))))
        assert abs(float(info["fun"]) - float(data[k + "fun"])) <= 1e-5 * abs(float(data[k + "fun"])), (k, "fun")
        for name, tl in (("accelerations", t["thrust"]), ("attitudes", t["pos"] * 100), ("thrusts", t["thrust"])):
            assert np.max(np.abs(h.to_host(out[name])[0] - data[k + name])) <= tl, (k, name)
        assert np.max(np.abs(h.to_host(out["body_rates"])[0] - data[k + "body_rates"])) <= t["rates"], (k, "body_rates")
    return worst


def check_solver_bifurcation_case(h: Harness, group=None):
    """tests/golden/bifurcation_case (make_golden_bifurcation.py): a problem whose outcome hangs on the last bit of an objective
    value.  The reference ends at (nit, nfev) = (3, 27), f = 2680.18; the other branch, 0.376 m away, at (3, 24), f = 2728.05.  Which
    side a build of the solver lands on is decided by its own last bits (summation tree, FMA contraction: the MI355X build takes the
    reference's branch, the host emulation -- sequential sums, no FMA -- the other); what is asserted is that the result IS one of the
    two outcomes, counts and positions to the usual tolerance.  Returns (branch, position error)."""
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data, meta = np.load(os.path.join(gold, "bifurcation_case.npz")), json.load(open(os.path.join(gold, "bifurcation_case.json")))
    N = meta["N"]
    prm = Params.reference_defaults(horizon=N, dt=meta["dt"], **meta["weights"])
    if group is not None:
        h.ops.lib.set_solver_variant(int(group) << 8)
    try:
        out = h.ops.solve(prm, h.prob(data["p0"][None]), h.prob(data["v0"][None]), h.prob(data["goal"][None]))
        info = h.ops.info_to_host(out["info"])[0]
        x = h.to_host(out["x"])[0].astype(float)
    finally:
        if group is not None:
            h.ops.lib.set_solver_variant(0)
    got = (int(info["nit"]), int(info["nfev"]), int(info["status"]))
    for branch, xkey in (("reference", "x_reference"), ("other_branch", "x_other_branch")):
        ref = meta[branch]
        if got == (ref["nit"], ref["nfev"], ref["status"]):
            assert abs(float(info["fun"]) - ref["fun"]) <= 1e-5 * ref["fun"], (branch, info)
            err = float(np.max(np.abs(x[:3 * N] - data[xkey][:3 * N])))
            assert err <= h.tol["pos"], (branch, err)
            return branch, err
    raise AssertionError(f"neither of the two published outcomes: {info}")


def cfg1_params(meta) -> Params:
    return Params.reference_defaults(horizon=meta["N"], dt=meta["dt"], max_velocity=meta["max_velocity"],
                                     max_acceleration=meta["max_acceleration"], max_iterations=meta["maxiter"], pgtol=meta["tol"],
                                     ftol=10 * meta["tol"])


def check_solver_cfg1(h: Harness, data, meta, rows=None):
    """BASELINE.json config 1's exact problem set (tests/golden/make_golden_cfg1.py: horizon 20, |v| <= 8 box, state
    (0,0,1), the contract goal + 100 x U(-5,5)^3) in ONE batched launch against what the reference returned: iteration
    and evaluation counts and status exactly, positions to the stated tolerance, trajectory arrays."""
    t = h.tol
    prm = cfg1_params(meta)
    N = meta["N"]
    rows = np.arange(meta["n"]) if rows is None else np.asarray(rows)
    B = len(rows)
    goals = data["goals_used"][rows]
    out = h.ops.solve(prm, h.prob(np.tile(data["p0"], (B, 1))), h.prob(np.tile(data["v0"], (B, 1))), h.prob(goals))
    info = h.ops.info_to_host(out["info"])
    got = np.stack([info["nit"], info["nfev"], info["status"]], axis=1)
    assert np.array_equal(got, data["info"][rows]), "nit / nfev / status"
    x = h.to_host(out["x"]).astype(float)
    assert np.max(np.abs(x[:, :6 * N] - data["x"][rows][:, :6 * N])) <= t["pos"], "x[P,V]"
    assert np.max(np.abs(x[:, 6 * N:] - data["x"][rows][:, 6 * N:])) <= t["thrust"], "x[T]"
    assert np.max(np.abs(x[:, 3 * N:6 * N])) <= meta["max_velocity"] + 1e-12, "velocity box of this configuration"
    assert np.max(np.abs(info["fun"] - data["fun"][rows]) / np.abs(data["fun"][rows])) <= 1e-5
    for name, tl in (("accelerations", t["thrust"]), ("attitudes", t["pos"] * 100), ("thrusts", t["thrust"]), ("body_rates", t["rates"])):
        assert np.max(np.abs(h.to_host(out[name]).astype(float) - data[name][rows])) <= tl, name
    return float(np.max(np.abs(x[:, :3 * N] - data["positions"][rows].reshape(B, -1))))


def check_solver_extraction(h: Harness, N: int, B: int, seed: int = 3):
    """pgtol = inf makes L-BFGS-B return the projected x0 untouched, which isolates the solver's
    own cold-start / projection / extraction code on arbitrary thrust sequences."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, pgtol=1e300)
    cfg = oracle_cfg(Params.reference_defaults(horizon=N))
    p0, v0, goal, T = random_batch(rng, B, N, spread=6.0)
    if N >= 6:
        T[0, 1] = 0.0; T[0, 3] = [7.0, 0, 0]; T[1, 0] = 0.0; T[1, 2] = [-3.0, 0, 0]; T[2, 4] = [0, 5.0, 0]
    X0 = np.concatenate([rng.uniform(-120, 120, (B, 3 * N)), rng.uniform(-15, 15, (B, 3 * N)), T.reshape(B, -1)], axis=1)
    out = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal), x0=h.prob(X0))
    b = orc.bounds(cfg)
    Xc = np.clip(X0.astype(h.dt).astype(float), b[:, 0], b[:, 1])
    info = h.ops.info_to_host(out["info"])
    assert np.all(info["nit"] == 0) and np.all(info["nfev"] == 1) and np.all(info["task"] == 1)
    vec_close(h.to_host(out["x"]).astype(float), Xc, h.tol["vec_rel"], "projected x0")
    ex = orc.extract_solution_batch(Xc, cfg)
    t = h.tol
    vec_close(h.to_host(out["accelerations"]), ex["accelerations"], t["vec_rel"], "accelerations")
    assert np.max(np.abs(h.to_host(out["attitudes"]) - ex["attitudes"])) <= t["pos"]
    assert np.max(np.abs(h.to_host(out["body_rates"]) - ex["body_rates"])) <= t["rates"]
    vec_close(h.to_host(out["thrusts"]), ex["thrusts"], t["vec_rel"], "thrusts")
    # the reduced output sets give the same bits for what they do return
    oa = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal), x0=h.prob(X0), want_trajectory="accelerations")
    ox = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal), x0=h.prob(X0), want_trajectory=False)
    assert oa["attitudes"] is None and oa["body_rates"] is None and oa["thrusts"] is None and ox["accelerations"] is None
    assert np.array_equal(h.to_host(oa["x"]), h.to_host(out["x"])) and np.array_equal(h.to_host(ox["x"]), h.to_host(out["x"]))
    assert np.array_equal(h.to_host(oa["accelerations"]), h.to_host(out["accelerations"]))
    # steady state: writing a previous call's output tensors again gives the same bits, and another output set is refused
    again = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal), x0=h.prob(X0), out=oa, want_trajectory="accelerations")
    assert again["x"] is oa["x"] and np.array_equal(h.to_host(again["x"]), h.to_host(out["x"]))
    try:
        h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal), x0=h.prob(X0), out=oa)
        raise AssertionError("an output dict of another output set was accepted")
    except ValueError:
        pass
    # cold start of the solver == a3 projected into the box (x0 = None path)
    out2 = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal))
    ref0 = orc.straight_line_init(p0.astype(h.dt).astype(float), v0.astype(h.dt).astype(float),
                                  goal.astype(h.dt).astype(float), cfg)
    vec_close(h.to_host(out2["x"]).astype(float), np.clip(ref0, b[:, 0], b[:, 1]), 10 * h.tol["vec_rel"], "cold start")


def mismatch_budget(B: int, dt) -> float:
    """Fraction of a batch whose (nit, nfev, status) may differ from SciPy's: none in float64; in float32 0.2 % (measured on MI355X:
    0.03 % over 340 000 + 126 000 random problems), i.e. at most one problem in batches below 500.  Whatever
    differs is still held to check_solver_vs_oracle's unconditional rule (SciPy's point, or an objective no worse)."""
    return 0.0 if np.dtype(dt) == np.float64 else max(0.002, 1.0 / B)


def check_solver_vs_oracle(h: Harness, N: int, B: int, seed: int = 11, group=None, **overrides):
    """Random problems of the cfg-2 distribution: batched HIP solve vs the oracle (SciPy) one by one.
    Returns (max position error over ALL problems that end at SciPy's point, fraction of problems whose nit/nfev/status differ).
    The assertion is unconditional: a problem whose counts differ from SciPy's (a line-search branch decided by the last bits of a
    float32 input, say) must still end within the position tolerance of SciPy's result OR at an objective value no worse than
    SciPy's (f_gpu <= f_scipy * (1 + 1e-6)): it may take another path, it may not return something worse.
    group: force the solver's lanes-per-problem (8 / 16 / 32 / 64; None = the library's own choice for this batch)."""
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, **overrides)
    cfg = oracle_cfg(prm)
    p0, v0, goal, _ = random_batch(rng, B, N)
    if group is not None:
        h.ops.lib.set_solver_variant(int(group) << 8)
    try:
        out = h.ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal))
        info = h.ops.info_to_host(out["info"])
        X = h.to_host(out["x"]).astype(float)
    finally:
        if group is not None:
            h.ops.lib.set_solver_variant(0)
    worst, mism = 0.0, 0
    for i in range(B):
        xr, ir = orc.solve(p0[i].astype(h.dt).astype(float), v0[i].astype(h.dt).astype(float),
                           goal[i].astype(h.dt).astype(float), cfg)
        same = (int(info["nit"][i]), int(info["nfev"][i]), int(info["status"][i])) == (ir["nit"], ir["nfev"], ir["status"])
        err = float(np.max(np.abs(X[i, :3 * N] - xr[:3 * N])))
        mism += 0 if same else 1
        if same or err <= h.tol["pos"]:
            worst = max(worst, err)
        else:
            assert float(info["fun"][i]) <= ir["fun"] * (1.0 + 1e-6) + 1e-12, (
                f"problem {i} (N={N}, seed={seed}): counts {int(info['nit'][i]), int(info['nfev'][i]), int(info['status'][i])} vs SciPy "
                f"{ir['nit'], ir['nfev'], ir['status']}, {err:.3e} m from SciPy's positions and a worse objective "
                f"({float(info['fun'][i]):.9g} vs {ir['fun']:.9g})")
    if mism:
        print(f"[check_solver_vs_oracle] N={N} B={B} seed={seed} {np.dtype(h.dt).name}: {mism} of {B} problems count differently from SciPy")
    return worst, mism / B


# --------------------------------------------------------------------------------------- f-2
def check_spheres_from_grid(h: Harness, data, meta, only=None):
    """Occupancy grid (produced by the reference's mapper) -> sphere table, against the reference's selection."""
    for c in meta["cases"]:
        k = c["key"]
        if only is not None and k not in only:
            continue
        grid, occ, exp = data[k + "grid"], data[k + "occ"], data[k + "spheres"]
        assert np.allclose(orc.local_grid_positions(c["centre"], c["size"], c["resolution"]), grid, rtol=0, atol=1e-12), k
        assert np.array_equal(orc.spheres_from_grid(grid, occ, target=c["target"]), exp), k
        spheres, count = h.ops.spheres_from_grid(h.prob(grid), h.prob(occ), threshold=0.6, target=c["target"], radius=1.0, cap=64)
        n = int(h.to_host(count)[0])
        assert n == c["n_spheres"] == len(exp), (k, n)
        got = h.to_host(spheres)[:n].astype(float)
        assert np.array_equal(got, exp.astype(h.dt).astype(float)), k          # copies of grid points: exact
    # cap smaller than the selection: the first `cap` spheres, count == cap
    c = meta["cases"][0]
    spheres, count = h.ops.spheres_from_grid(h.prob(data[c["key"] + "grid"]), h.prob(data[c["key"] + "occ"]), target=c["target"], cap=5)
    assert int(h.to_host(count)[0]) == 5
    assert np.array_equal(h.to_host(spheres).astype(float), data[c["key"] + "spheres"][:5].astype(h.dt).astype(float))
    # ragged sizes: M not a multiple of the 256-cell stride, and M == 0
    rng = np.random.default_rng(1)
    for M in (0, 1, 63, 64, 65, 255, 257, 1000):
        grid = rng.uniform(-5, 5, (M, 3)); occ = rng.choice([0.5, 0.9], M, p=[0.7, 0.3])
        exp = orc.spheres_from_grid(grid, occ, target=7)
        spheres, count = h.ops.spheres_from_grid(h.prob(grid), h.prob(occ), target=7, cap=32)
        n = int(h.to_host(count)[0])
        assert n == len(exp), (M, n, len(exp))
        assert np.array_equal(h.to_host(spheres)[:n].astype(float), exp.astype(h.dt).astype(float)), M


# --------------------------------------------------------------------------------------- robustness
def check_solver_edge_inputs(h: Harness):
    """(1) goal_position None (planner.py:524/:546/:567 skip the goal terms) against the oracle;
    (2) NaN / Inf / denormal inputs: every wavefront must terminate with a SciPy-style status --
    a wavefront that never exits would hang the GPU."""
    prm = Params.reference_defaults(horizon=6, has_goal=0)
    p0, v0 = np.array([[1.0, 2.0, 3.0]]), np.array([[0.5, -0.2, 0.1]])
    out = h.ops.solve(prm, h.prob(p0), h.prob(v0), None)
    xr, ir = orc.solve(p0[0].astype(h.dt).astype(float), v0[0].astype(h.dt).astype(float), None, orc.OracleConfig())
    info = h.ops.info_to_host(out["info"])[0]
    assert (int(info["nit"]), int(info["nfev"]), int(info["status"])) == (ir["nit"], ir["nfev"], ir["status"])
    assert np.max(np.abs(h.to_host(out["x"])[0] - xr)) <= h.tol["pos"]
    prm = Params.reference_defaults(horizon=6)
    bad = [np.nan, np.inf, -np.inf, 1e30, 1e-320]
    rows = [(b, w) for b in bad for w in range(3)]
    P0 = np.tile([0.0, 0.0, 1.0], (len(rows), 1)); V0 = np.zeros((len(rows), 3)); G = np.tile([5.0, 3.0, 2.0], (len(rows), 1))
    with np.errstate(over="ignore"):
        for i, (b, w) in enumerate(rows):
            (P0, V0, G)[w][i, 1] = b
        out = h.ops.solve(prm, h.prob(P0), h.prob(V0), h.prob(G))
    info = h.ops.info_to_host(out["info"])
    assert set(np.unique(info["status"])) <= {0, 1, 2} and np.all(info["task"] >= 1) and np.all(info["task"] <= 5)
    assert np.all(info["nfev"] <= 2 + prm.max_iterations * (prm.max_linesearch + 1))
    # a non-finite state is projected into the box, a non-finite goal ends in ABNORMAL (f is NaN/Inf)
    finite_goal = np.isfinite(G).all(axis=1)
    assert np.all(np.isfinite(h.to_host(out["x"])[finite_goal]))
    assert np.all(info["status"][~finite_goal] == 2)


def check_population_sums(h: Harness, rows=90, B=1000, seed=0):
    """se3mpc_population_sums_* against float64 NumPy on the same (rounded) inputs: plain sums, MPPI weights with a
    host reference cost and with the reference taken from a device argmin key; ld > B; B = 0."""
    rng = np.random.default_rng(seed)
    ld = B + 7
    X = rng.normal(size=(rows, ld)).astype(h.dt)
    cost = rng.uniform(50.0, 80.0, ld).astype(h.dt)
    dX, dc = h.to_dev(X), h.to_dev(cost)
    Xd, cd = X[:, :B].astype(np.float64), cost[:B].astype(np.float64)
    out = h.to_host(h.ops.population_sums(dX, B=B))
    assert np.allclose(out[:rows], Xd.sum(1), rtol=1e-12, atol=1e-9) and out[rows] == B
    lam, ref = 2.5, float(cd.min())
    w = np.exp(-(cd - ref) / lam)
    out = h.to_host(h.ops.population_sums(dX, cost=dc, temperature=lam, cost_ref=ref, B=B))
    assert np.allclose(out[:rows], (Xd * w).sum(1), rtol=1e-11, atol=1e-9) and np.isclose(out[rows], w.sum(), rtol=1e-12)
    key = h.to_dev(np.zeros(1, dtype=np.int64))
    h.ops.argmin(h.to_dev(np.ascontiguousarray(cost[:B])), index_base=0, out=key)
    ref32 = float(np.float32(cd.min()))                            # the key carries the cost as float32
    w = np.exp(-(cd - ref32) / lam)
    out = h.to_host(h.ops.population_sums(dX, cost=dc, temperature=lam, ref_key=key, B=B))
    assert np.allclose(out[:rows], (Xd * w).sum(1), rtol=1e-11, atol=1e-9) and np.isclose(out[rows], w.sum(), rtol=1e-12)
    out = h.to_host(h.ops.population_sums(dX, B=0))
    assert np.all(out == 0)


def check_transpose(h: Harness, shapes=((270, 300), (300, 270), (54, 1000), (1000, 54), (576, 257), (257, 576), (5, 4096), (4096, 5),
                                        (70, 70), (1, 256), (256, 1), (90, 255))):
    """se3mpc_transpose_*: the strip kernel (one small dimension against >= 256) both ways, ragged last strips, and the
    generic 64 x 64 tile -- bit-exact."""
    rng = np.random.default_rng(12)
    for rows, cols in shapes:
        a = rng.normal(size=(rows, cols)).astype(h.dt)
        assert np.array_equal(h.to_host(h.ops.transpose(h.to_dev(a))), a.T), (rows, cols)
