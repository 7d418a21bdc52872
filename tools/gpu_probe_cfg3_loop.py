"""Developer probe (GPU box, under rocprofv3): BASELINE config 3 INSIDE the iteration loop -- 8192 trajectories x horizon 50 x 16 spheres,
K = 16 obstacle-aware iterations per launch (se3mpc_rollout_iterate_obstacles_*), both workgroup shapes, beside the plain K = 16 loop at
the same horizon and the one-shot fused rollout + obstacle kernel.  `python3 tools/gpu_probe_cfg3_loop.py [B] [K] [time]`; with `time` (not
under the profiler) it prints HIP-event timings of both shapes instead."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
g = torch.Generator(device=dev); g.manual_seed(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = 50; prm = Params.reference_defaults(horizon=N)
sph = torch.cat([torch.round(torch.rand(16, 3, device=dev, generator=g) * 30) / 2 - 3.75, torch.ones(16, 1, device=dev)], dim=1)
T = torch.randn(3 * N, B, device=dev, generator=g) * 2; T[2::3] += 14.715
q0 = torch.rand(3, B, device=dev, generator=g) * 4 - 2; w0 = torch.rand(3, B, device=dev, generator=g) * 10 - 5
gl = torch.rand(3, B, device=dev, generator=g) * 4 - 2
Tout = torch.empty_like(T); cost = torch.empty(B, device=dev); grad = torch.empty_like(T)
for wsel in (128, 256):
    ops.lib.set_rollout_variant(wsel)
    for _ in range(20): ops.rollout_iterate(prm, q0, w0, gl, T, K, 1e-3, out=(Tout, cost, grad), spheres=sph, obstacle_weight=1000.0, want_penalty=False)
ops.lib.set_rollout_variant(0)
for _ in range(20): ops.rollout_iterate(prm, q0, w0, gl, T, K, 1e-3, out=(Tout, cost, grad))
for _ in range(20): ops.rollout_obstacles(prm, q0, w0, gl, T, sph)
torch.cuda.synchronize()
print("done", B, K)
if "time" not in sys.argv[3:]:
    sys.exit(0)


def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


# HIP-event timings (meaningful without the tracer): microseconds per launch at K = 0 and K, per pass in between
for name, wsel in (("narrow (3 wavefronts x 64 trajectories, table in LDS)", 128), ("wide (7 wavefronts x 32 trajectories, table in registers; default)", 0)):
    ops.lib.set_rollout_variant(wsel)
    t0 = timed(lambda: ops.rollout_iterate(prm, q0, w0, gl, T, 0, 1e-3, out=(Tout, cost, grad), spheres=sph, obstacle_weight=1000.0, want_penalty=False))
    tk = timed(lambda: ops.rollout_iterate(prm, q0, w0, gl, T, K, 1e-3, out=(Tout, cost, grad), spheres=sph, obstacle_weight=1000.0, want_penalty=False))
    print(f"B={B} obstacle-aware loop, {name}: K=0 {t0:.1f} us, K={K} {tk:.1f} us, {(tk - t0) / max(K, 1):.2f} us per pass")
ops.lib.set_rollout_variant(0)
tk = timed(lambda: ops.rollout_iterate(prm, q0, w0, gl, T, K, 1e-3, out=(Tout, cost, grad)))
print(f"plain loop K={K}: {tk:.1f} us")
