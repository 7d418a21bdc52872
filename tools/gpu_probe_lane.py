"""Developer probe (GPU box): streaming bandwidth of every lane-layout kernel at a saturating batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
N, B, K = 30, 1 << 20, 16
prm = Params.reference_defaults(horizon=N)
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn(9 * N, B, device=dev, generator=g)
p0 = torch.randn(3, B, device=dev, generator=g); v0 = torch.randn(3, B, device=dev, generator=g); goal = torch.randn(3, B, device=dev, generator=g)
T = X[6 * N:].contiguous()
sph = torch.rand(K, 4, device=dev, generator=g) * 10
def bench(name, fn, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:28s} {ms*1e3:9.1f} us  {bytes_/ms/1e9:8.3f} TB/s (algorithmic)", flush=True)
f4 = 4 * B
bench("init", lambda: ops.init(prm, p0, v0, goal), f4 * (9 + 9 * N))
bench("init+project", lambda: ops.init(prm, p0, v0, goal, project=True), f4 * (9 + 9 * N))
bench("cost_grad", lambda: ops.cost_grad(prm, X, goal), f4 * (18 * N + 4))
bench("cost only", lambda: ops.cost_grad(prm, X, goal, want_grad=False), f4 * (9 * N + 4))
bench("dynamics_residual", lambda: ops.dynamics_residual(prm, X, p0, v0), f4 * (9 * N + 6 + 6 * N))
bench("obstacle (materialised)", lambda: ops.obstacle_residual(prm, X, sph), f4 * (3 * N + N * K + 2))
bench("obstacle (reduced)", lambda: ops.obstacle_residual(prm, X, sph, materialize=False), f4 * (3 * N + 2))
bench("physical_constraints", lambda: ops.physical_constraints(prm, X), f4 * (6 * N + 4 * N))
bench("extract", lambda: ops.extract(prm, T), f4 * (3 * N + 10 * N))
bench("is_plan_valid", lambda: ops.is_plan_valid(prm, X[:3 * N], X[3 * N:6 * N]), f4 * (6 * N + 1))
bench("transpose 9N x B", lambda: ops.transpose(X), f4 * 18 * N)
cost = torch.rand(B, device=dev)
bench("argmin", lambda: ops.argmin(cost), f4)
# config 3: horizon 50, K = 16 spheres: fused rollout+obstacles vs rollout(states) followed by obstacle reduce
N3 = 50; prm3 = Params.reference_defaults(horizon=N3)
for B3 in (8192, 1 << 20):
    T3 = torch.randn(3 * N3, B3, device=dev, generator=g) * 2; T3[2::3] += 14.715
    q0 = torch.rand(3, B3, device=dev, generator=g) * 40 - 20; w0 = torch.rand(3, B3, device=dev, generator=g) * 10 - 5
    gl = torch.rand(3, B3, device=dev, generator=g) * 40 - 20
    bytes3 = 4 * B3 * (6 * N3 + 12)
    bench(f"cfg3 fused B={B3}", lambda: ops.rollout_obstacles(prm3, q0, w0, gl, T3, sph), bytes3)
    def unfused():
        c, gT, P, V = ops.rollout_cost_grad(prm3, q0, w0, gl, T3, want_states=True)
        X = torch.cat([P, V, T3], 0)
        return ops.obstacle_residual(prm3, X, sph, materialize=False)
    bench(f"cfg3 unfused B={B3}", unfused, bytes3)
