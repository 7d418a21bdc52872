"""Parity checks of the consumer-side kernels (se3mpc_control_*, se3mpc_closed_loop_*), shared by the CPU suite (product
sources compiled for the host, tests/emu) and the GPU suite: against the vectors the reference's own classes produced
(tests/golden/controller_cases.npz) and against the oracle (oracle/controller_oracle.py) on random batches.

Tolerances: f64 1e-9 on every output per call (the kernel and NumPy differ only by the last bit of sin / cos / acos / sqrt and
of three-term sums); closed loops 1e-8 over <= 100 steps.  f32: the law has hard branches (saturation, singularity, tilt limit),
so f32 is compared call by call FROM THE SAME STATE with the oracle evaluated on the f32-rounded inputs: thrust 2e-5 relative,
torque 2e-4 absolute (torques are differences of O(1) rotation entries times gains of ~20) wherever both took the same
branches, and at most 2 % of the calls may sit on another side of a branch."""
import numpy as np

from dart_planner_amd.capi import ControllerParams, SimulatorParams
from oracle import controller_oracle as co


def oracle_config(seq=None) -> co.ControllerConfig:
    cfg = co.ControllerConfig()
    if seq is not None:
        cfg.anti_windup_method = seq["anti_windup"]
        cfg.yaw_singularity_fallback_method = seq["fallback"]
        cfg.default_heading_yaw = seq["default_heading_yaw"]
    return cfg


def state_to_oracle(st_rows: np.ndarray, cfg) -> co.ControllerState:
    """device controller state (B, 12) -> oracle ControllerState"""
    B = len(st_rows)
    s = co.ControllerState(B, cfg)
    s.integral = st_rows[:, 0:3].copy(); s.last_time = st_rows[:, 3].copy(); s.last_valid_thrust = st_rows[:, 4].copy()
    s.unsaturated_thrust = st_rows[:, 5].copy(); s.unsaturated_torque = st_rows[:, 6:9].copy()
    s.failsafe_count = st_rows[:, 9].astype(np.int64); s.halvings = st_rows[:, 10].astype(np.int64)
    fl = st_rows[:, 11].astype(np.int64)
    s.failsafe_active = (fl & 1) != 0; s.thrust_saturated = (fl & 2) != 0
    s.torque_saturated = np.stack([(fl & 4) != 0, (fl & 8) != 0, (fl & 16) != 0], axis=1)
    return s


def check_defaults(h):
    """se3mpc_controller_default_params / se3mpc_simulator_default_params == the values the reference instantiates
    (golden meta: GeometricController(tuning_profile="sitl_optimized").config, DroneSimulator())."""
    import json, os
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "controller_cases.json")))
    cp, sp = h.ops.lib.controller_default_params(), h.ops.lib.simulator_default_params()
    c = meta["config"]
    for k in ("kp_pos", "ki_pos", "kd_pos", "kp_att", "kd_att", "inertia", "max_torque_xyz", "max_integral_per_axis"):
        assert list(getattr(cp, k)) == c[k], k
    for k in ("max_integral_pos", "max_tilt_angle", "mass", "gravity", "max_thrust", "min_thrust", "tracking_error_threshold",
              "velocity_error_threshold", "back_calculation_gain", "integral_decay_factor", "saturation_threshold",
              "yaw_singularity_threshold", "default_heading_yaw"):
        assert getattr(cp, k) == c[k], k
    assert (cp.anti_windup_method, cp.yaw_fallback_method) == (0, 0) and c["anti_windup_method"] == "clamping" and c["yaw_singularity_fallback_method"] == "skip_yaw"
    s = meta["simulator"]
    assert (sp.mass, sp.gravity, list(sp.inertia), sp.max_thrust, sp.max_torque) == (s["mass"], s["gravity"], s["inertia"], s["max_thrust"], s["max_torque"])
    ref = ControllerParams.from_config(co.ControllerConfig())
    assert bytes(ref) == bytes(cp)
    assert bytes(SimulatorParams.reference_defaults()) == bytes(sp)


def check_control_sequences(h, data, meta):
    """Every call sequence the reference's controller produced, call by call, through se3mpc_control_* (B = 1 per sequence: the
    configurations differ), including the controller members after each call."""
    dt_ = h.dt
    f64 = dt_ == np.float64
    mism = calls = 0
    for seq in meta["sequences"]:
        k = seq["key"]
        cfg = oracle_config(seq)
        cp = ControllerParams.from_config(cfg)
        st = h.ops.controller_state(cp, 1)
        for i in range(seq["calls"]):
            a = lambda nm: h.to_dev(np.ascontiguousarray(data[k + nm][i][None].astype(dt_)))
            t = h.to_dev(np.array([data[k + "t"][i]], dtype=np.float64))
            before = h.to_host(st).copy()
            out = h.ops.control(cp, st, t, a("pos"), a("vel"), a("att"), a("omega"), a("dpos"), a("dvel"), a("dacc"),
                                h.to_dev(np.array([data[k + "yaw"][i]], dtype=dt_)), h.to_dev(np.array([data[k + "yaw_rate"][i]], dtype=dt_)),
                                want_body_rate=seq["kind"] == "body_rate")
            after = h.to_host(st)
            calls += 1
            if f64:
                if seq["kind"] == "body_rate":
                    assert abs(float(h.to_host(out["body_thrust"])[0]) - data[k + "br_thrust"][i]) <= 1e-9, (k, i)
                    assert np.max(np.abs(h.to_host(out["body_rates"])[0] - data[k + "br_rates"][i])) <= 1e-8, (k, i)
                else:
                    assert abs(float(h.to_host(out["thrust"])[0]) - data[k + "thrust"][i]) <= 1e-9, (k, i)
                    assert np.max(np.abs(h.to_host(out["torque"])[0] - data[k + "torque"][i])) <= 1e-9, (k, i)
                assert np.max(np.abs(after[0, 0:3] - data[k + "integral"][i])) <= 1e-11, (k, i)
                fl = int(after[0, 11])
                assert bool(fl & 1) == bool(data[k + "failsafe_active"][i]) and bool(fl & 2) == bool(data[k + "thrust_saturated"][i]), (k, i)
                assert [bool(fl & 4), bool(fl & 8), bool(fl & 16)] == [bool(x) for x in data[k + "torque_saturated"][i]], (k, i)
                assert int(after[0, 9]) == int(data[k + "failsafe_count"][i]) and int(after[0, 10]) == int(data[k + "halvings"][i]), (k, i)
                assert abs(after[0, 4] - data[k + "last_valid_thrust"][i]) <= 1e-9, (k, i)
                assert after[0, 3] == data[k + "t"][i]
            else:
                # f32: the oracle from the SAME controller state on the f32-rounded inputs
                r = lambda nm: data[k + nm][i][None].astype(dt_).astype(float)
                so = state_to_oracle(before, cfg)
                th, tq, fo = co.compute_control(so, cfg, np.array([data[k + "t"][i]]), r("pos"), r("vel"), r("att"), r("omega"), r("dpos"),
                                                r("dvel"), r("dacc"), np.array([float(dt_(data[k + "yaw"][i]))]), np.array([float(dt_(data[k + "yaw_rate"][i]))]))
                fl = int(h.to_host(out["flags"])[0])
                same = (bool(fl & 1) == bool(fo["failsafe"][0]) and bool(fl & 4) == bool(fo["thrust_saturated"][0]) and bool(fl & 8) == bool(fo["singular"][0])
                        and bool(fl & 16) == bool(fo["tilt_limited"][0]) and [bool(fl & 32), bool(fl & 64), bool(fl & 128)] == [bool(x) for x in so.torque_saturated[0] & ~fo["failsafe"][0]])
                if not same:
                    mism += 1
                    # put the device state on the oracle's side so the sequence continues from a common state
                    continue
                assert abs(float(h.to_host(out["thrust"])[0]) - th[0]) <= 2e-5 * max(1.0, abs(th[0])), (k, i)
                assert np.max(np.abs(h.to_host(out["torque"])[0] - tq[0])) <= 2e-4, (k, i, h.to_host(out["torque"])[0], tq[0])
                assert np.max(np.abs(after[0, 0:3] - so.integral[0])) <= 1e-5, (k, i)
    assert mism <= 0.02 * calls, (mism, calls)
    return mism, calls


def check_closed_loops_golden(h, data, meta):
    """The reference's closed loops (planner plan -> sampler -> controller -> simulator) in ONE launch each, per-step logs against
    the reference's own."""
    assert h.dt == np.float64
    cp = ControllerParams.from_config(co.ControllerConfig())
    worst = 0.0
    for lp in meta["loops"]:
        k = lp["key"]
        sp = SimulatorParams.reference_defaults(max_thrust=lp["max_thrust"], max_torque=lp["max_torque"])
        st = h.ops.controller_state(cp, 1)
        d = lambda a: h.to_dev(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
        time = d([meta["T0"]]); pos = d([lp["p0"]]); vel = d([lp["v0"]]); att = d(np.zeros((1, 3))); om = d(np.zeros((1, 3)))
        out = h.ops.closed_loop(cp, sp, st, time, pos, vel, att, om, d(data[k + "ts"]), d(data[k + "P"]), d(data[k + "V"]), d(data[k + "A"]),
                                nsteps=lp["nsteps"], sim_dt=lp["sim_dt"], wind=None if lp["wind"] is None else d(lp["wind"]),
                                gust=None if lp["gust"] is None else (lp["gust"][0], lp["gust"][1]), stop_at_plan_end=not lp["emergency"], log=True)
        act = data[k + "active"].astype(bool)
        assert int(h.to_host(out["steps_taken"])[0]) == int(act.sum()) == lp["steps_active"], k
        ls, lc, lt = h.to_host(out["log_state"])[:, 0], h.to_host(out["log_cmd"])[:, 0], h.to_host(out["log_time"])[:, 0]
        ref_state = np.concatenate([data[k + "pos"], data[k + "vel"], data[k + "att"], data[k + "omega"]], axis=1)
        e = float(np.max(np.abs(ls - ref_state)))
        worst = max(worst, e)
        assert e <= 1e-8, (k, e)
        assert np.max(np.abs(lt - data[k + "t"])) <= 1e-9, k
        assert np.max(np.abs(lc[act, 0] - data[k + "thrust"][act])) <= 1e-8 and np.max(np.abs(lc[act, 1:] - data[k + "torque"][act])) <= 1e-8, k
        assert np.all(np.isnan(lc[~act]))
        final = np.concatenate([h.to_host(pos)[0], h.to_host(vel)[0], h.to_host(att)[0], h.to_host(om)[0], h.to_host(time)])
        assert np.max(np.abs(final - data[k + "final"])) <= 1e-8, k
    return worst


def random_plans(rng, B, N, t0, dt):
    """Smooth random plans shaped like the planner's output: positions along a line p0 -> goal with noise, constant velocity blocks."""
    ts = t0 + np.arange(N) * dt
    a = rng.uniform(-3, 3, (B, 1, 3)); b = rng.uniform(-3, 3, (B, 1, 3))
    s = np.linspace(0, 1, N)[None, :, None]
    P = a + s * (b - a) + rng.normal(0, 0.02, (B, N, 3))
    V = np.tile(rng.uniform(-2, 2, (B, 1, 3)), (1, N, 1)) + rng.normal(0, 0.05, (B, N, 3))
    A = rng.normal(0, 0.3, (B, N, 3))
    return ts, P, V, A


def check_closed_loop_vs_oracle(h, B=96, N=12, nsteps=40, seed=0, per_drone_plans=True):
    """Random drones, plans, winds and a gust against the batched oracle; the plan ends inside the run for some drones
    (per-drone timestamps), so `stop_at_plan_end` is exercised lane by lane."""
    rng = np.random.default_rng(seed)
    cfg = co.ControllerConfig()
    cp, sp = ControllerParams.from_config(cfg), SimulatorParams.reference_defaults()
    sim = co.SimulatorConfig()
    t0 = 500.0
    ts, P, V, A = random_plans(rng, B, N, t0, 0.02)
    TS = np.tile(ts, (B, 1)) - rng.choice([0.0, 0.05, 0.3], (B, 1))                       # some plans start (and end) earlier
    pos = P[:, 0] + rng.normal(0, 0.3, (B, 3)); vel = V[:, 0] + rng.normal(0, 0.3, (B, 3))
    att = rng.normal(0, 0.1, (B, 3)); om = rng.normal(0, 0.3, (B, 3)); t = np.full(B, t0)
    wind = rng.normal(0, 1.5, (B, 3))
    gust = (nsteps // 2, [4.0, -1.0, 0.5])
    dt_ = h.dt
    r = lambda a: np.asarray(a).astype(dt_).astype(float)
    if not per_drone_plans:
        P, V, A, TS = P[0], V[0], A[0], TS[0]
    fin, log = co.closed_loop(cfg, sim, co.ControllerState(B, cfg), r(pos), r(vel), r(att), r(om), t, TS, r(P), r(V), r(A), nsteps, 0.01,
                              wind=r(wind), gust_step=gust[0], gust_wind=gust[1])
    d = lambda a, ty=dt_: h.to_dev(np.ascontiguousarray(np.asarray(a).astype(ty)))
    st = h.ops.controller_state(cp, B)
    dtime, dpos, dvel, datt, dom = d(t, np.float64), d(pos), d(vel), d(att), d(om)
    out = h.ops.closed_loop(cp, sp, st, dtime, dpos, dvel, datt, dom, d(TS, np.float64), d(P), d(V), d(A), nsteps=nsteps, sim_dt=0.01,
                            wind=d(wind), gust=gust, log=True)
    taken = h.to_host(out["steps_taken"])
    assert np.array_equal(taken, log["active"].sum(0)), "steps taken per drone"
    assert len(np.unique(taken)) >= 3 or not per_drone_plans          # plans end at different steps, lane by lane
    ls = h.to_host(out["log_state"]).astype(float)
    ref = np.concatenate([log["pos"], log["vel"], log["att"], log["omega"]], axis=2)
    if dt_ == np.float64:
        assert np.max(np.abs(ls - ref)) <= 1e-8
        assert np.max(np.abs(h.to_host(dpos) - fin["pos"])) <= 1e-8 and np.max(np.abs(h.to_host(dtime) - fin["t"])) <= 1e-9
    else:
        # f32 closed loop: drones that never sat near a branch track the f64 oracle to a few 1e-4 over 40 steps
        err = np.max(np.abs(ls - ref), axis=(0, 2))
        assert np.median(err) <= 5e-3 and np.mean(err <= 5e-2) >= 0.9, (np.median(err), np.mean(err <= 5e-2))
    return float(np.max(np.abs(ls - ref)))
