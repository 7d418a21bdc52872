#!/usr/bin/env python3
"""Golden vectors for the consumer side of the Planner->Controller contract, produced by
RUNNING THE REFERENCE'S OWN CLASSES in the build container:

* ``GeometricController.compute_control`` / ``compute_body_rate_command``
  (/root/reference/src/dart_planner/control/geometric_controller.py:413-512, :706-726; profile "sitl_optimized",
  control_config.py:95-111) -- call sequences on fresh controllers: normal tracking, thrust saturation (both ends), the
  tilt limit, the yaw-singularity branch and its three fallbacks, both anti-windup methods, invalid dt (repeated and
  distant timestamps -> failsafe, halved gains, recovery);
* ``OnboardController._interpolate_trajectory`` (control/onboard_controller.py:43-93) -- the reference's plan sampler;
* ``DroneSimulator.step`` (utils/drone_simulator.py:52-72);
* closed loops of those three around plans of the reference's ``SE3MPCPlanner``, shaped like the reference's contract
  tests (tests/test_planner_controller_contract.py:115-162, :255-316: nominal, constant wind, gust at step 50,
  actuator saturation, emergency hover), with the wall clock pinned so the run is reproducible.

* ``compute_control_fast`` / ``compute_control_from_fast_state`` (:253-411, :728-768), the unit-free path of the reference's
  400 Hz hardware loop (hardware/pixhawk_interface.py:401): own sequences, and sequences that interleave it with
  ``compute_control`` on one controller (shared integral, gains, saturation flags).

Same stand-ins as make_golden.py (identity units for the missing ``pint``).  Writes controller_cases.npz / .json.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.environ.get("SE3MPC_GOLDEN_OUT", HERE)      # tests/test_golden_reproducible.py writes to a scratch directory
sys.path.insert(0, HERE)
from make_golden import _install_standins  # noqa: E402


def main():
    tmp = _install_standins()
    try:
        import logging
        logging.disable(logging.CRITICAL)
        import dart_planner.planning.se3_mpc_planner as planner_mod
        from dart_planner.planning.se3_mpc_planner import SE3MPCPlanner
        from dart_planner.control.geometric_controller import GeometricController, GeometricControllerConfig
        from dart_planner.control.onboard_controller import OnboardController
        from dart_planner.utils.drone_simulator import DroneSimulator
        from dart_planner.common.types import DroneState, Trajectory, ControlCommand

        # compute_body_rate_command reads `control_cmd.torque.magnitude` (controller.py:719): under the identity-units stand-in a
        # quantity is a bare ndarray, so Q_ (the controller module's, and the one ControlCommand.__post_init__ reaches through
        # ensure_units, common/types.py:110-112) is given the one pint attribute that method touches
        import dart_planner.control.geometric_controller as ctrl_mod
        import dart_planner.common.units as units_mod

        class _Magnitude(np.ndarray):
            @property
            def magnitude(self):
                return np.asarray(self)

        ctrl_mod.Q_ = units_mod.Q_ = lambda value, unit=None: np.asarray(value, dtype=float).view(_Magnitude)

        out, meta = {}, {}
        c0 = GeometricController(tuning_profile="sitl_optimized")
        cfg = c0.config
        meta["config"] = {k: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else v) for k, v in vars(cfg).items()}
        meta["gravity_vector"] = np.asarray(c0._gravity_vector).tolist()
        sim0 = DroneSimulator()
        meta["simulator"] = dict(mass=sim0.mass, gravity=sim0.gravity, inertia=np.diag(sim0.inertia).tolist(), max_thrust=sim0.max_thrust,
                                 max_torque=sim0.max_torque)

        def mkstate(t, p, v, a, w):
            return DroneState(timestamp=float(t), position=np.array(p, float), velocity=np.array(v, float), attitude=np.array(a, float),
                              angular_velocity=np.array(w, float))

        def snapshot(c):
            return dict(integral=np.array(c.integral_vel_error, float), failsafe_active=bool(c.failsafe_active),
                        failsafe_count=int(c.failsafe_count), halvings=int(round(np.log2(20.0 / c.config.kp_pos[0]))),
                        last_valid_thrust=float(c.last_valid_thrust), torque_saturated=np.array(c.last_torque_saturated, bool),
                        thrust_saturated=bool(c.last_thrust_saturated))

        # ------------------------------------------------------------------ A. call sequences
        rng = np.random.default_rng(20261004)
        seqs = []
        kinds = ["small", "small", "medium", "medium", "large", "large", "descend", "attitude", "attitude", "mixed_dt", "mixed_dt", "mixed_dt",
                 "back_calculation", "back_calculation", "default_heading", "maintain_current", "maintain_current", "body_rate", "body_rate",
                 "vertical_singular", "unknown_method", "yawed"]
        for si, kind in enumerate(kinds):
            conf = GeometricControllerConfig()
            if kind == "back_calculation":
                conf.anti_windup_method = "back_calculation"
            if kind in ("default_heading", "maintain_current"):
                conf.yaw_singularity_fallback_method = kind
                conf.default_heading_yaw = 0.7
            if kind == "unknown_method":
                conf.yaw_singularity_fallback_method = "none_of_these"
            ctrl = GeometricController(config=conf, tuning_profile="sitl_optimized")
            T = 48
            spread = dict(small=0.05, medium=0.6, large=6.0, descend=2.0, attitude=0.3, mixed_dt=0.5, back_calculation=4.0, default_heading=0.5,
                          maintain_current=0.5, body_rate=0.4, vertical_singular=0.004, unknown_method=0.5, yawed=0.5)[kind]
            t = 50.0 + si
            rec = {k: [] for k in ("t", "pos", "vel", "att", "omega", "dpos", "dvel", "dacc", "yaw", "yaw_rate", "thrust", "torque", "integral",
                                   "failsafe_active", "failsafe_count", "halvings", "last_valid_thrust", "torque_saturated", "thrust_saturated",
                                   "br_thrust", "br_rates")}
            pos = rng.uniform(-5, 5, 3); vel = rng.uniform(-1, 1, 3)
            for k in range(T):
                if kind == "mixed_dt":
                    step = [0.0025, 0.01, 0.05, 0.0, 0.2, 0.1, 0.1000001, -0.01][rng.integers(0, 8)]
                else:
                    step = [0.0025, 0.01][si % 2] if k else 0.0
                if k == 17 and kind in ("medium", "large"):
                    step = 0.0                                   # one repeated timestamp -> "Invalid dt" failsafe -> gains halve
                t += step
                att = rng.normal(0, 0.05 if kind != "attitude" else 0.6, 3)
                if kind == "yawed":
                    att[2] = rng.uniform(-3, 3)
                omega = rng.normal(0, 0.2 if kind != "attitude" else 2.0, 3)
                pos = pos + rng.normal(0, 0.02, 3); vel = vel + rng.normal(0, 0.05, 3)
                dpos = pos + rng.normal(0, spread, 3)
                dvel = vel + rng.normal(0, spread, 3)
                dacc = rng.normal(0, 0.5 * spread, 3)
                if kind == "descend":
                    dacc[2] -= 9.0                              # desired free fall: thrust under the lower limit
                if kind == "vertical_singular":
                    dacc[:] = 0.0; dpos[0] = pos[0] + 0.055       # b3 ~ (0.11, 0, 0.994): singular with |b3z| >= 0.99
                yaw = float(rng.uniform(-3.1, 3.1)) if kind in ("attitude", "default_heading", "maintain_current", "yawed", "mixed_dt") else 0.0
                yaw_rate = float(rng.normal(0, 0.5)) if kind in ("attitude", "yawed") else 0.0
                st = mkstate(t, pos, vel, att, omega)
                if kind == "body_rate":
                    br = ctrl.compute_body_rate_command(st, dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate)
                    rec["br_thrust"].append(float(br.thrust)); rec["br_rates"].append(np.array(br.body_rates, float))
                    thrust, torque = np.nan, np.full(3, np.nan)
                else:
                    cmd = ctrl.compute_control(st, dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate)
                    thrust, torque = float(cmd.thrust), np.array(cmd.torque, float)
                    rec["br_thrust"].append(np.nan); rec["br_rates"].append(np.full(3, np.nan))
                snap = snapshot(ctrl)
                for nm, v in (("t", t), ("pos", pos), ("vel", vel), ("att", att), ("omega", omega), ("dpos", dpos), ("dvel", dvel), ("dacc", dacc),
                              ("yaw", yaw), ("yaw_rate", yaw_rate), ("thrust", thrust), ("torque", torque)):
                    rec[nm].append(np.array(v, float))
                for nm in ("integral", "failsafe_active", "failsafe_count", "halvings", "last_valid_thrust", "torque_saturated", "thrust_saturated"):
                    rec[nm].append(snap[nm])
            key = f"q{si:02d}_"
            for nm, v in rec.items():
                out[key + nm] = np.array(v)
            seqs.append(dict(key=key, kind=kind, calls=T, anti_windup=conf.anti_windup_method, fallback=conf.yaw_singularity_fallback_method,
                             default_heading_yaw=float(conf.default_heading_yaw)))
        meta["sequences"] = seqs

        # ------------------------------------------------------------------ B. the plan sampler
        N = 6
        ts = 100.0 + np.arange(N) * 0.0025
        P, V, A = rng.uniform(-5, 5, (N, 3)), rng.uniform(-2, 2, (N, 3)), rng.uniform(-1, 1, (N, 3))
        tq = np.concatenate([ts, ts[:-1] + 0.001, [99.0, 99.999999, 100.0 + 1e-12, ts[-1] + 1e-9, 200.0], rng.uniform(99.99, 100.02, 20)])
        tr = Trajectory(timestamps=ts, positions=P, velocities=V, accelerations=A)
        res = [OnboardController._interpolate_trajectory(None, float(t), tr) for t in tq]
        out["i_ts"], out["i_P"], out["i_V"], out["i_A"], out["i_tq"] = ts, P, V, A, tq
        out["i_pos"], out["i_vel"], out["i_acc"] = (np.array([np.array(r[j], float) for r in res]) for j in range(3))
        tr2 = Trajectory(timestamps=ts, positions=P)             # a plan without velocities / accelerations (the emergency trajectory has them)
        res2 = [OnboardController._interpolate_trajectory(None, float(t), tr2) for t in tq[:8]]
        out["i_pos_only"] = np.array([np.array(r[0], float) for r in res2])
        assert all(np.all(np.array(r[1]) == 0) and np.all(np.array(r[2]) == 0) for r in res2)

        # ------------------------------------------------------------------ C. simulator steps
        S = 24
        sp, sv, sa, sw = rng.uniform(-5, 5, (S, 3)), rng.uniform(-3, 3, (S, 3)), rng.normal(0, 0.3, (S, 3)), rng.normal(0, 1, (S, 3))
        sth = rng.uniform(-3, 30, S); stq = rng.normal(0, 6, (S, 3)); swind = rng.normal(0, 2, (S, 3)); sdt = rng.choice([0.01, 0.0025, 0.15], S)
        smax = [(20.0, 10.0), (5.0, 2.0)]
        rows = []
        for i in range(S):
            mt, mq = smax[i % 2]
            sim = DroneSimulator(wind=swind[i] if i % 3 else None, max_thrust=mt, max_torque=mq)
            ns = sim.step(mkstate(3.0 + i, sp[i], sv[i], sa[i], sw[i]), ControlCommand(thrust=float(sth[i]), torque=stq[i].copy()), float(sdt[i]))
            rows.append(np.concatenate([ns.position, ns.velocity, ns.attitude, ns.angular_velocity, [ns.timestamp]]))
        out.update(s_pos=sp, s_vel=sv, s_att=sa, s_omega=sw, s_thrust=sth, s_torque=stq, s_dt=sdt, s_out=np.array(rows),
                   s_wind=np.array([swind[i] if i % 3 else np.zeros(3) for i in range(S)]), s_max=np.array([smax[i % 2] for i in range(S)]))

        # ------------------------------------------------------------------ D. closed loops around the reference planner's plans
        T0 = 1000.0
        loops = []

        def closed_loop(tag, horizon, stamp_offset, nsteps, wind=None, gust=None, max_thrust=20.0, max_torque=10.0, emergency=False,
                        p0=(0.0, 0.0, 1.0), v0=(0.0, 0.0, 0.0), goal=(5.0, 3.0, 2.0), sim_dt=0.01):
            planner_mod.time.time = lambda: T0 + stamp_offset                 # the plan is stamped with the wall clock (planner.py:221)
            from dart_planner.planning.se3_mpc_planner import SE3MPCConfig
            planner = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=horizon))
            ctrl = GeometricController(tuning_profile="sitl_optimized")
            sim = DroneSimulator(wind=None if wind is None else np.array(wind, float), max_thrust=max_thrust, max_torque=max_torque)
            st = mkstate(T0, p0, v0, np.zeros(3), np.zeros(3))
            if emergency:
                planner.goal_position = None
                traj = planner._generate_emergency_trajectory(st)            # stamped from state.timestamp
            else:
                traj = planner.plan_trajectory(st, np.array(goal, float))
            log = {k: [] for k in ("pos", "vel", "att", "omega", "t", "thrust", "torque", "active")}
            active = True
            for i in range(nsteps):
                t_cur = st.timestamp
                if not emergency and t_cur > traj.timestamps[-1]:
                    active = False                                            # the reference loops `break` here (:130-131)
                if active:
                    tp, tv, ta = OnboardController._interpolate_trajectory(None, t_cur, traj)
                    cmd = ctrl.compute_control(st, np.array(tp, float), np.array(tv, float), np.array(ta, float))
                    thrust, torque = float(cmd.thrust), np.array(cmd.torque, float)
                else:
                    thrust, torque = np.nan, np.full(3, np.nan)
                for nm, v in (("pos", st.position), ("vel", st.velocity), ("att", st.attitude), ("omega", st.angular_velocity), ("t", st.timestamp),
                              ("thrust", thrust), ("torque", torque), ("active", active)):
                    log[nm].append(np.array(v))
                if not active:
                    continue
                if gust is not None and i == gust[0]:
                    sim.wind = np.array(gust[1], float)
                st = sim.step(st, cmd, sim_dt)
            key = f"l_{tag}_"
            for nm, v in log.items():
                out[key + nm] = np.array(v)
            out[key + "ts"] = np.array(traj.timestamps, float)
            out[key + "P"], out[key + "V"], out[key + "A"] = (np.array(x, float) for x in (traj.positions, traj.velocities, traj.accelerations))
            out[key + "final"] = np.concatenate([st.position, st.velocity, st.attitude, st.angular_velocity, [st.timestamp]])
            loops.append(dict(key=key, tag=tag, horizon=horizon, nsteps=nsteps, sim_dt=sim_dt, wind=wind, gust=gust, max_thrust=max_thrust,
                              max_torque=max_torque, emergency=emergency, p0=list(p0), v0=list(v0), goal=list(goal),
                              steps_active=int(np.sum(log["active"]))))

        # the contract test's own situation: the plan is stamped AFTER the state's clock (setUp's time.time() precedes the
        # planner's), so every sample is the plan's first point and the loop never leaves the plan
        closed_loop("contract_nominal", 6, 0.5, 100)
        closed_loop("contract_wind", 6, 0.5, 100, wind=[2.0, 0.0, 0.0])
        closed_loop("contract_gust", 6, 0.5, 100, gust=(50, [5.0, 0.0, 0.0]))
        closed_loop("contract_saturation", 6, 0.5, 100, max_thrust=5.0, max_torque=2.0)
        closed_loop("contract_emergency", 6, 0.0, 50, emergency=True)
        # plans stamped at the state's clock: the sampler walks the plan and the loop stops at its end
        closed_loop("walk_h6", 6, 0.0, 12, sim_dt=0.0025)
        closed_loop("walk_h30", 30, 0.0, 40, sim_dt=0.0025, goal=(1.0, -2.0, 1.5))
        closed_loop("walk_h50_fast", 50, 0.0, 30, sim_dt=0.01, p0=(2.0, 1.0, 3.0), v0=(1.0, -0.5, 0.2), goal=(-3.0, 4.0, 1.0))
        meta["loops"] = loops
        meta["T0"] = T0

        # ------------------------------------------------------------------ E. the unit-free "fast" path of the 400 Hz hardware loop
        # compute_control_fast / compute_control_from_fast_state (controller.py:253-411, :728-768; caller hardware/pixhawk_interface.py:401):
        # dt is an argument, an invalid dt returns the vehicle's hover thrust without touching the failsafe, mass and gravity come from
        # common/vehicle_params.py (not the controller config), saturations are counted.  Own random stream: sections A-D keep their bytes.
        from dart_planner.common.types import FastDroneState
        rng_f = np.random.default_rng(20261005)
        meta["vehicle_constants"] = dict(mass=float(c0._fast_mass), gravity=float(c0._fast_gravity_magnitude),
                                         gravity_vector=np.asarray(c0._fast_gravity_vector, float).tolist(), min_thrust=float(c0._fast_min_thrust))
        fast = []
        fkinds = ["small", "small", "large", "descend", "attitude", "bad_dt", "bad_dt", "back_calculation", "maintain_current", "default_heading",
                  "fast_state", "interleaved", "interleaved", "yawed"]
        for si, kind in enumerate(fkinds):
            conf = GeometricControllerConfig()
            if kind == "back_calculation":
                conf.anti_windup_method = "back_calculation"
            if kind in ("default_heading", "maintain_current"):
                conf.yaw_singularity_fallback_method = kind
                conf.default_heading_yaw = -1.1
            ctrl = GeometricController(config=conf, tuning_profile="sitl_optimized")
            T = 40
            spread = dict(small=0.05, large=6.0, descend=2.0, attitude=0.3, bad_dt=0.5, back_calculation=4.0, maintain_current=0.5, default_heading=0.5,
                          fast_state=0.4, interleaved=1.5, yawed=0.5)[kind]
            t = 300.0 + si
            rec = {k: [] for k in ("t", "dt", "path", "pos", "vel", "att", "omega", "dpos", "dvel", "dacc", "yaw", "yaw_rate", "thrust", "torque",
                                   "integral", "failsafe_active", "failsafe_count", "halvings", "last_valid_thrust", "torque_saturated",
                                   "thrust_saturated", "unsaturated_thrust", "thrust_saturation_count", "torque_saturation_count")}
            pos = rng_f.uniform(-5, 5, 3); vel = rng_f.uniform(-1, 1, 3)
            for k in range(T):
                dt = [0.0025, 0.001][si % 2]
                if kind == "bad_dt":
                    dt = [0.0025, 0.0, -0.01, 0.2, 0.1, 0.1000001, 0.05][rng_f.integers(0, 7)]
                path = 1                                        # 1 fast, 0 compute_control (same controller: shared integral, gains, flags)
                if kind == "interleaved":
                    path = int(rng_f.integers(0, 2)) if k not in (20, 21) else 0
                t += 0.0025 if not (kind == "interleaved" and k == 21) else 0.0     # two compute_control calls on one stamp: its failsafe halves the gains the fast path reads
                att = rng_f.normal(0, 0.05 if kind != "attitude" else 0.6, 3)
                if kind == "yawed":
                    att[2] = rng_f.uniform(-3, 3)
                omega = rng_f.normal(0, 0.2 if kind != "attitude" else 2.0, 3)
                pos = pos + rng_f.normal(0, 0.02, 3); vel = vel + rng_f.normal(0, 0.05, 3)
                dpos = pos + rng_f.normal(0, spread, 3); dvel = vel + rng_f.normal(0, spread, 3); dacc = rng_f.normal(0, 0.5 * spread, 3)
                if kind == "descend":
                    dacc[2] -= 9.0
                yaw = float(rng_f.uniform(-3.1, 3.1)) if kind in ("attitude", "default_heading", "maintain_current", "yawed") else 0.0
                yaw_rate = float(rng_f.normal(0, 0.5)) if kind in ("attitude", "yawed") else 0.0
                if path == 0:
                    cmd = ctrl.compute_control(mkstate(t, pos, vel, att, omega), dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate)
                    thrust, torque = float(cmd.thrust), np.array(cmd.torque, float)
                elif kind == "fast_state":
                    fs = FastDroneState(timestamp=float(t), position=pos.copy(), velocity=vel.copy(), attitude=att.copy(), angular_velocity=omega.copy())
                    thrust, torque = ctrl.compute_control_from_fast_state(fs, dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate, dt)
                else:
                    thrust, torque = ctrl.compute_control_fast(pos.copy(), vel.copy(), att.copy(), omega.copy(), dpos.copy(), dvel.copy(), dacc.copy(),
                                                               yaw, yaw_rate, dt)
                snap = snapshot(ctrl)
                snap.update(unsaturated_thrust=float(ctrl.unsaturated_thrust), thrust_saturation_count=int(ctrl._thrust_saturation_count),
                            torque_saturation_count=int(ctrl._torque_saturation_count))
                for nm, v in (("t", t), ("dt", dt), ("path", path), ("pos", pos), ("vel", vel), ("att", att), ("omega", omega), ("dpos", dpos),
                              ("dvel", dvel), ("dacc", dacc), ("yaw", yaw), ("yaw_rate", yaw_rate), ("thrust", float(thrust)),
                              ("torque", np.array(torque, float))):
                    rec[nm].append(np.array(v, float))
                for nm in ("integral", "failsafe_active", "failsafe_count", "halvings", "last_valid_thrust", "torque_saturated", "thrust_saturated",
                           "unsaturated_thrust", "thrust_saturation_count", "torque_saturation_count"):
                    rec[nm].append(snap[nm])
            key = f"f{si:02d}_"
            for nm, v in rec.items():
                out[key + nm] = np.array(v)
            fast.append(dict(key=key, kind=kind, calls=T, anti_windup=conf.anti_windup_method, fallback=conf.yaw_singularity_fallback_method,
                             default_heading_yaw=float(conf.default_heading_yaw)))
        meta["fast_sequences"] = fast

        # ------------------------------------------------------------------ F. quaternion attitudes (controller.py:770-803): a 4-vector attitude is a
        # (w, x, y, z) quaternion, normalised before use, the identity below a norm of 1e-6.  Single calls on fresh controllers (own generator: the
        # draws above keep their values), away from the yaw singularity (its maintain_current fallback reads att[2] of the raw 4-vector, :662).
        rng_q = np.random.default_rng(20261005)
        Q = 8
        rec = {k: [] for k in ("pos", "vel", "quat", "euler", "omega", "dpos", "dvel", "dacc", "yaw", "yaw_rate", "thrust", "torque", "thrust_euler", "torque_euler")}
        for qi in range(Q):
            roll, pitch, yaw_c = rng_q.normal(0, 0.004, 2).tolist() + [float(rng_q.uniform(-3.0, 3.0))]       # (small errors: unsaturated torques)
            cr, sr, cp_, sp_, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw_c / 2), np.sin(yaw_c / 2)
            quat = np.array([cr * cp_ * cy + sr * sp_ * sy, sr * cp_ * cy - cr * sp_ * sy, cr * sp_ * cy + sr * cp_ * sy, cr * cp_ * sy - sr * sp_ * cy])
            quat = quat * [1.0, 1.0, 3.0, 0.25, 1.0, 1e-3, 1e-9, 1.0][qi]                      # lengths 1, 3, 0.25, 1e-3; 1e-9: below the threshold -> identity
            pos, vel = rng_q.normal(0, 0.5, 3), rng_q.normal(0, 0.5, 3)
            omega = rng_q.normal(0, 0.01, 3)
            dpos, dvel, dacc = pos + rng_q.normal(0, 0.01, 3), vel + rng_q.normal(0, 0.01, 3), rng_q.normal(0, 0.02, 3)
            yaw, yaw_rate = (0.0 if qi == 6 else yaw_c) + float(rng_q.normal(0, 0.004)), float(rng_q.normal(0, 0.01))
            cmd = GeometricController(tuning_profile="sitl_optimized").compute_control(mkstate(5.0, pos, vel, quat, omega), dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate)
            euler = np.zeros(3) if qi == 6 else np.array([roll, pitch, yaw_c])
            cme = GeometricController(tuning_profile="sitl_optimized").compute_control(mkstate(5.0, pos, vel, euler, omega), dpos.copy(), dvel.copy(), dacc.copy(), yaw, yaw_rate)
            for nm, v in (("pos", pos), ("vel", vel), ("quat", quat), ("euler", euler), ("omega", omega), ("dpos", dpos), ("dvel", dvel), ("dacc", dacc), ("yaw", yaw),
                          ("yaw_rate", yaw_rate), ("thrust", float(cmd.thrust)), ("torque", np.array(cmd.torque, float)), ("thrust_euler", float(cme.thrust)),
                          ("torque_euler", np.array(cme.torque, float))):
                rec[nm].append(np.array(v, float))
        for nm, v in rec.items():
            out["quat_" + nm] = np.array(v)
        meta["quaternion_calls"] = Q

        np.savez_compressed(os.path.join(OUT_DIR, "controller_cases.npz"), **out)
        with open(os.path.join(OUT_DIR, "controller_cases.json"), "w") as f:
            json.dump(meta, f, indent=1)
        print("wrote controller_cases.npz / .json:", len(seqs), "sequences,", len(loops), "closed loops,", len(fast), "fast-path sequences")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
