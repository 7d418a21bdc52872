"""The controller / simulator / plan-sampler oracle (oracle/controller_oracle.py) against what the reference's own
classes returned (tests/golden/controller_cases.npz, written by tests/golden/make_golden_controller.py)."""
import json
import os

import numpy as np
import pytest

from oracle import controller_oracle as co

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden_controller():
    return np.load(os.path.join(GOLDEN, "controller_cases.npz")), json.load(open(os.path.join(GOLDEN, "controller_cases.json")))


def config_from_meta(meta, seq=None) -> co.ControllerConfig:
    c = meta["config"]
    cfg = co.ControllerConfig()
    for k in ("kp_pos", "ki_pos", "kd_pos", "kp_att", "kd_att", "inertia", "max_torque_xyz", "max_integral_per_axis"):
        assert np.array_equal(getattr(cfg, k), np.array(c[k], float)), k          # the oracle's defaults ARE the reference's values
    for k in ("max_integral_pos", "max_tilt_angle", "mass", "gravity", "max_thrust", "min_thrust", "tracking_error_threshold",
              "velocity_error_threshold", "back_calculation_gain", "integral_decay_factor", "saturation_threshold", "yaw_singularity_threshold"):
        assert getattr(cfg, k) == c[k], k
    if seq is not None:
        cfg.anti_windup_method = seq["anti_windup"]
        cfg.yaw_singularity_fallback_method = seq["fallback"]
        cfg.default_heading_yaw = seq["default_heading_yaw"]
    return cfg


def test_controller_call_sequences(golden_controller):
    data, meta = golden_controller
    assert meta["gravity_vector"] == [0.0, 0.0, -meta["config"]["gravity"]]
    seen = dict(failsafe=0, thrust_saturated=0, singular=0, tilt_limited=0, torque_saturated=0, calls=0)
    for seq in meta["sequences"]:
        k = seq["key"]
        cfg = config_from_meta(meta, seq)
        st = co.ControllerState(1, cfg)
        for i in range(seq["calls"]):
            a = lambda nm: data[k + nm][i][None].astype(float)
            args = (st, cfg, a("t"), a("pos"), a("vel"), a("att"), a("omega"), a("dpos"), a("dvel"), a("dacc"), a("yaw"), a("yaw_rate"))
            if seq["kind"] == "body_rate":
                nt, rates, thrust, torque = co.compute_body_rate_command(*args)
                assert abs(nt[0] - data[k + "br_thrust"][i]) <= 1e-12, (k, i)
                assert np.max(np.abs(rates[0] - data[k + "br_rates"][i])) <= 1e-10, (k, i)
            else:
                thrust, torque, fl = co.compute_control(*args)
                assert abs(thrust[0] - data[k + "thrust"][i]) <= 1e-10, (k, i, thrust, data[k + "thrust"][i])
                assert np.max(np.abs(torque[0] - data[k + "torque"][i])) <= 1e-10, (k, i)
                for nm in ("failsafe", "thrust_saturated", "singular", "tilt_limited"):
                    seen[nm] += int(fl[nm][0])
            assert np.max(np.abs(st.integral[0] - data[k + "integral"][i])) <= 1e-12, (k, i)
            assert bool(st.failsafe_active[0]) == bool(data[k + "failsafe_active"][i]), (k, i)
            assert int(st.failsafe_count[0]) == int(data[k + "failsafe_count"][i]), (k, i)
            assert int(st.halvings[0]) == int(data[k + "halvings"][i]), (k, i)
            assert abs(st.last_valid_thrust[0] - data[k + "last_valid_thrust"][i]) <= 1e-10, (k, i)
            assert np.array_equal(st.torque_saturated[0], data[k + "torque_saturated"][i]), (k, i)
            assert bool(st.thrust_saturated[0]) == bool(data[k + "thrust_saturated"][i]), (k, i)
            seen["torque_saturated"] += int(st.torque_saturated[0].any())
            seen["calls"] += 1
    # every branch of the control law is in the fixtures
    assert all(seen[nm] >= 10 for nm in ("failsafe", "thrust_saturated", "singular", "tilt_limited", "torque_saturated")), seen


def test_controller_fast_path_sequences(golden_controller):
    """compute_control_fast / compute_control_from_fast_state (the 400 Hz hardware loop's path), alone and interleaved with
    compute_control on one controller, call by call against the reference's own returns and members."""
    data, meta = golden_controller
    vc = meta["vehicle_constants"]
    veh = co.VehicleConstants()
    assert (veh.mass, veh.gravity) == (vc["mass"], vc["gravity"]) and vc["gravity_vector"] == [0.0, 0.0, -vc["gravity"]]
    assert vc["min_thrust"] == meta["config"]["min_thrust"] * vc["mass"] * vc["gravity"]
    seen = dict(bad_dt=0, thrust_saturated=0, singular=0, tilt_limited=0, unsaturated_axis=0, fast=0, normal=0, halved_fast=0)
    for seq in meta["fast_sequences"]:
        k = seq["key"]
        cfg = config_from_meta(meta, seq)
        st = co.ControllerState(1, cfg)
        n_thrust = n_torque = 0
        for i in range(seq["calls"]):
            a = lambda nm: data[k + nm][i][None].astype(float)
            if int(data[k + "path"][i]) == 0:
                thrust, torque, _ = co.compute_control(st, cfg, a("t"), a("pos"), a("vel"), a("att"), a("omega"), a("dpos"), a("dvel"), a("dacc"),
                                                       a("yaw"), a("yaw_rate"))
                seen["normal"] += 1
            else:
                thrust, torque, fl = co.compute_control_fast(st, cfg, veh, a("dt"), a("pos"), a("vel"), a("att"), a("omega"), a("dpos"), a("dvel"),
                                                             a("dacc"), a("yaw"), a("yaw_rate"))
                n_thrust += int(fl["thrust_saturated"][0]); n_torque += int(fl["torque_saturated"][0].sum())     # :315-320, :404
                for nm in ("bad_dt", "thrust_saturated", "singular", "tilt_limited"):
                    seen[nm] += int(fl[nm][0])
                seen["unsaturated_axis"] += int((~fl["torque_saturated"][0]).sum() if not fl["bad_dt"][0] else 0)
                seen["fast"] += 1
                seen["halved_fast"] += int(st.halvings[0] > 0)
            assert abs(thrust[0] - data[k + "thrust"][i]) <= 1e-10, (k, i, thrust, data[k + "thrust"][i])
            assert np.max(np.abs(torque[0] - data[k + "torque"][i])) <= 1e-10, (k, i)
            assert np.max(np.abs(st.integral[0] - data[k + "integral"][i])) <= 1e-12, (k, i)
            assert bool(st.failsafe_active[0]) == bool(data[k + "failsafe_active"][i]) and int(st.failsafe_count[0]) == int(data[k + "failsafe_count"][i]), (k, i)
            assert int(st.halvings[0]) == int(data[k + "halvings"][i]), (k, i)
            assert abs(st.last_valid_thrust[0] - data[k + "last_valid_thrust"][i]) <= 1e-10, (k, i)
            assert np.array_equal(st.torque_saturated[0], data[k + "torque_saturated"][i]) and bool(st.thrust_saturated[0]) == bool(data[k + "thrust_saturated"][i]), (k, i)
            assert abs(st.unsaturated_thrust[0] - data[k + "unsaturated_thrust"][i]) <= 1e-10, (k, i)
            assert (n_thrust, n_torque) == (int(data[k + "thrust_saturation_count"][i]), int(data[k + "torque_saturation_count"][i])), (k, i)
    assert all(seen[nm] >= 10 for nm in seen), seen


def test_controller_batched_equals_one_by_one(golden_controller):
    """The oracle is batched over drones: stacking all sequences as a batch gives the same numbers as one at a time."""
    data, meta = golden_controller
    seqs = [s for s in meta["sequences"] if s["kind"] not in ("body_rate", "back_calculation", "default_heading", "maintain_current", "unknown_method")]
    cfg = config_from_meta(meta)
    st = co.ControllerState(len(seqs), cfg)
    for i in range(seqs[0]["calls"]):
        a = lambda nm: np.stack([data[s["key"] + nm][i] for s in seqs]).astype(float)
        thrust, torque, _ = co.compute_control(st, cfg, a("t"), a("pos"), a("vel"), a("att"), a("omega"), a("dpos"), a("dvel"), a("dacc"), a("yaw"),
                                               a("yaw_rate"))
        assert np.max(np.abs(thrust - a("thrust"))) <= 1e-10 and np.max(np.abs(torque - a("torque"))) <= 1e-10, i


def test_plan_sampler(golden_controller):
    data, _ = golden_controller
    p, v, a = co.interpolate_trajectory(data["i_tq"], data["i_ts"], data["i_P"], data["i_V"], data["i_A"])
    assert np.array_equal(p, data["i_pos"]) and np.array_equal(v, data["i_vel"]) and np.array_equal(a, data["i_acc"])
    p, v, a = co.interpolate_trajectory(data["i_tq"][:8], data["i_ts"], data["i_P"])
    assert np.array_equal(p, data["i_pos_only"]) and not v.any() and not a.any()


def test_simulator_step(golden_controller):
    data, meta = golden_controller
    sm = meta["simulator"]
    assert (sm["mass"], sm["gravity"], sm["inertia"], sm["max_thrust"], sm["max_torque"]) == (1.5, 9.81, [0.1, 0.1, 0.2], 20.0, 10.0)
    for i in range(len(data["s_thrust"])):
        sim = co.SimulatorConfig(max_thrust=float(data["s_max"][i, 0]), max_torque=float(data["s_max"][i, 1]))
        r = lambda nm: data[nm][i][None].astype(float)
        out = co.simulator_step(sim, r("s_pos"), r("s_vel"), r("s_att"), r("s_omega"), np.array([3.0 + i]), r("s_thrust"), r("s_torque"),
                                float(data["s_dt"][i]), data["s_wind"][i])
        got = np.concatenate([out[0][0], out[1][0], out[2][0], out[3][0], out[4]])
        assert np.max(np.abs(got - data["s_out"][i])) <= 1e-13, i


def test_closed_loops(golden_controller):
    """Planner plan -> sampler -> controller -> simulator, step by step, against the reference's own loop."""
    data, meta = golden_controller
    cfg = config_from_meta(meta)
    for lp in meta["loops"]:
        k = lp["key"]
        sim = co.SimulatorConfig(max_thrust=lp["max_thrust"], max_torque=lp["max_torque"])
        st = co.ControllerState(1, cfg)
        z = np.zeros((1, 3))
        final, log = co.closed_loop(cfg, sim, st, np.array([lp["p0"]], float), np.array([lp["v0"]], float), z, z, np.array([meta["T0"]]),
                                    data[k + "ts"], data[k + "P"], data[k + "V"], data[k + "A"], lp["nsteps"], lp["sim_dt"], wind=lp["wind"],
                                    gust_step=None if lp["gust"] is None else lp["gust"][0], gust_wind=None if lp["gust"] is None else lp["gust"][1],
                                    stop_at_plan_end=not lp["emergency"])
        act = data[k + "active"].astype(bool)
        assert np.array_equal(log["active"][:, 0], act), k
        assert int(act.sum()) == lp["steps_active"]
        for nm in ("pos", "vel", "att", "omega"):
            assert np.max(np.abs(log[nm][:, 0] - data[k + nm])) <= 1e-9, (k, nm)
        assert np.max(np.abs(log["t"][:, 0] - data[k + "t"])) <= 1e-9
        assert np.max(np.abs(log["thrust"][act, 0] - data[k + "thrust"][act])) <= 1e-9, k
        assert np.max(np.abs(log["torque"][act, 0] - data[k + "torque"][act])) <= 1e-9, k
        got = np.concatenate([final["pos"][0], final["vel"][0], final["att"][0], final["omega"][0], final["t"]])
        assert np.max(np.abs(got - data[k + "final"])) <= 1e-9, k
