import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from dart_planner_amd.common.types import DroneState
from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), precision="f64")
st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
rng = np.random.default_rng(0)
goals = rng.uniform(-5, 5, (240, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
ts = []
for i, g in enumerate(goals):
    t0 = time.perf_counter(); pl.plan_shooting(st, g, n_samples=8192, iters=16, seed=0); torch.cuda.synchronize()
    if i >= 20: ts.append((time.perf_counter() - t0) * 1e3)
print("captured plan_shooting p50 %.3f ms p95 %.3f ms" % (np.percentile(ts, 50), np.percentile(ts, 95)))
