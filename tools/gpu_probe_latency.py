"""Developer probe (GPU box): where the wall time of ONE plan_trajectory goes (staged copies vs host-mapped buffers)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.common.types import DroneState
from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner

def pct(ts):
    return f"p50 {np.percentile(ts, 50)*1e3:.1f} us  p95 {np.percentile(ts, 95)*1e3:.1f} us"

st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
rng = np.random.default_rng(0)
goals = rng.uniform(-5, 5, (420, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
for N in (30, 6):
    for prec in ("f64", "f32"):
        for mapped in (0, 16):
            pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=N), precision=prec)
            pl.host_mapped_max_problems = mapped
            ts, ts_solve = [], []
            for i, g in enumerate(goals):
                t0 = time.perf_counter()
                tr = pl.plan_trajectory(st, g)
                if i >= 20: ts.append((time.perf_counter() - t0) * 1e3)
            p0 = np.array([[0.0, 0.0, 1.0]]); v0 = np.zeros((1, 3))
            for i, g in enumerate(goals):
                t0 = time.perf_counter()
                pl._solve_batch(p0, v0, g.reshape(1, 3), None, prec)
                if i >= 20: ts_solve.append((time.perf_counter() - t0) * 1e3)
            print(f"N={N} {prec} mapped<={mapped}: plan_trajectory {pct(ts)} | _solve_batch {pct(ts_solve)}", flush=True)
# raw C call + synchronise (lower bound of the host path)
from dart_planner_amd.ops import Ops
ops = pl._get_ops()
for N in (30, 6):
    prm = pl._params(horizon=N)
    h_in = torch.zeros((3, 1, 3), dtype=torch.float64, pin_memory=True); h_in[0, 0, 2] = 1.0; h_in[2, 0] = torch.tensor([5.0, 3.0, 2.0])
    h_out = torch.empty((ops.packed_size(1, N, "f64"),), dtype=torch.uint8, pin_memory=True)
    ts = []
    for i in range(420):
        t0 = time.perf_counter()
        ops.solve_packed(prm, h_in, out=h_out, host_mapped=True)
        torch.cuda.current_stream().synchronize()
        if i >= 20: ts.append((time.perf_counter() - t0) * 1e3)
    print(f"N={N} raw solve_packed(host_mapped)+sync: {pct(ts)}")
# raw se3mpc_plan_host_* (launch + spin on the completion ticket: no hipStreamSynchronize)
import ctypes
for N in (30, 6):
    for suf, dt in (("f32", torch.float32), ("f64", torch.float64)):
        prm = pl._params(horizon=N)
        h_in = torch.zeros((3, 1, 3), dtype=dt, pin_memory=True); h_in[0, 0, 2] = 1.0; h_in[2, 0] = torch.tensor([5.0, 3.0, 2.0])
        h_out = torch.empty((ops.packed_size(1, N, suf),), dtype=torch.uint8, pin_memory=True)
        h_done = torch.zeros((8,), dtype=torch.int64, pin_memory=True)
        esz = 4 if suf == "f32" else 8
        o_x, o_acc, o_att, o_rates, o_thr, o_info, _ = ops._packed_offsets(1, N, esz)
        pin, base = h_in.data_ptr(), h_out.data_ptr()
        fn = getattr(ops.lib._dll, f"se3mpc_plan_host_{suf}")
        stream = torch.cuda.current_stream().cuda_stream
        ts = []
        for i in range(420):
            t0 = time.perf_counter()
            rc = fn(ctypes.byref(prm), 1, pin, pin + 3 * esz, pin + 6 * esz, 0, base + o_x, base + o_info, base + o_acc, base + o_att, base + o_rates, base + o_thr,
                    h_done.data_ptr(), i + 1, 2000.0, stream)
            assert rc == 0
            if i >= 20: ts.append((time.perf_counter() - t0) * 1e3)
        print(f"N={N} {suf} raw se3mpc_plan_host (launch + ticket spin): {pct(ts)}")
import cProfile, pstats
pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=30), precision="f64")
for g in goals[:20]: pl.plan_trajectory(st, g)
pr = cProfile.Profile(); pr.enable()
for g in goals[:200]: pl.plan_trajectory(st, g)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
