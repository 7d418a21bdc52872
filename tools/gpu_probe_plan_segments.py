"""Developer probe (GPU box): where the wall time of one SE3MPCPlanner.plan_trajectory goes, by re-running its steps with a clock between them
(median over 2000 plans; the clock reads themselves cost ~0.05 us each).  `python tools/gpu_probe_plan_segments.py`."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from dart_planner_amd.common.types import DroneState
from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner, TASK_MESSAGES
pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), precision="f32")
st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
rng = np.random.default_rng(0)
goals = rng.uniform(-5, 5, (2100, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
for g in goals[:50]:
    pl.plan_trajectory(st, g)
T = {k: [] for k in ("whole", "sense", "solve_se3_mpc", "act", "bookkeeping")}
clk = time.perf_counter
for g in goals[50:2050]:
    t0 = clk()
    cs, _, _ = pl.sense(st, g); t1 = clk()
    sol = pl.plan(cs); t2 = clk()
    tr = pl.act(sol, cs, time.time()); t3 = clk()
    ms = (clk() - t0) * 1e3
    pl.planning_times.append(ms); pl.plan_count += 1
    pl._update_planning_stats(ms, bool(pl.convergence_history[-1])); t4 = clk()
    for k, v in (("whole", t4 - t0), ("sense", t1 - t0), ("solve_se3_mpc", t2 - t1), ("act", t3 - t2), ("bookkeeping", t4 - t3)):
        T[k].append(v * 1e6)
for k, v in T.items():
    print(f"{k:16s} p50 {np.percentile(v, 50):6.2f} us   p95 {np.percentile(v, 95):6.2f} us")
# inside _solve_se3_mpc: the C call alone on the same buffers
io = pl._io
prm = pl._params(has_goal=1)
ts = []
for i in range(2000):
    io["ticket"] = tk = io["ticket"] + 1
    pin = io["ptr_in"]
    t0 = clk()
    io["plan_fn"](ctypes.byref(prm), 1, pin[0], pin[1], pin[2], 0, *io["ptr_out"], io["ptr_done"], tk, 2000.0, io["plan_stream_handle"])
    ts.append((clk() - t0) * 1e6)
print(f"{'C call alone':16s} p50 {np.percentile(ts, 50):6.2f} us   p95 {np.percentile(ts, 95):6.2f} us")
