import sys, numpy as np, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from dart_planner_amd.common.types import DroneState
from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
rng = np.random.default_rng(0)
goals = rng.uniform(-5, 5, (1220, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), precision="f32")
for g in goals[:20]: pl.plan_trajectory(st, g)
pr = cProfile.Profile(); pr.enable()
for g in goals[20:1020]: pl.plan_trajectory(st, g)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(24)
