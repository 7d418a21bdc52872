import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dart_planner_amd.common.types import DroneState
from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), precision="f64")
st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
rng = np.random.default_rng(0)
goals = rng.uniform(-5, 5, (600, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
for g in goals[:30]: pl.plan_shooting(st, g, n_samples=8192, iters=16, seed=0)
graph, io = next(iter(pl._shooting_graphs.values()))
clk = time.perf_counter
T = {"replay": [], "sync": [], "whole": []}
for g in goals[30:530]:
    t0 = clk(); pl.plan_shooting(st, g, n_samples=8192, iters=16, seed=0); T["whole"].append((clk() - t0) * 1e6)
for _ in range(500):
    t0 = clk(); graph.replay(); t1 = clk(); torch.cuda.current_stream().synchronize(); t2 = clk()
    T["replay"].append((t1 - t0) * 1e6); T["sync"].append((t2 - t1) * 1e6)
for k, v in T.items(): print(k, "p50 %.1f us p95 %.1f us" % (np.percentile(v, 50), np.percentile(v, 95)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): graph.replay()
e1.record(); torch.cuda.synchronize()
print("graph device time per replay (back to back): %.1f us" % (e0.elapsed_time(e1) * 1e3 / 200))
