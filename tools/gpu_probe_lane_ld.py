"""Developer probe (GPU box): does the leading dimension matter?  At B = 1 M trajectories a lane-layout row is 4 MB, so the rows a
wavefront walks are 4 MB apart -- the same channel / bank / page offsets row after row?  The parity-form stream kernels with
ld = B + 64 k (k = 0, 1, 3, 17), against the bare copy / read / write of tools/probes/probe_rows at the same leading dimensions."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
N, B = 30, 1 << 20
prm = Params.reference_defaults(horizon=N)
f4 = 4 * B
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for k in (0, 1, 3, 17):
    ld = B + 64 * k
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn(9 * N, ld, device=dev, generator=g)
    p0 = torch.randn(3, ld, device=dev, generator=g); v0 = torch.randn(3, ld, device=dev, generator=g); goal = torch.randn(3, ld, device=dev, generator=g)
    rows = [("cost_grad", lambda: ops.cost_grad(prm, X, goal, B=B), f4 * (18 * N + 4)),
            ("cost only", lambda: ops.cost_grad(prm, X, goal, want_grad=False, B=B), f4 * (9 * N + 4)),
            ("dynamics_residual", lambda: ops.dynamics_residual(prm, X, p0, v0, B=B), f4 * (9 * N + 6 + 6 * N)),
            ("physical_constraints", lambda: ops.physical_constraints(prm, X, B=B), f4 * (6 * N + 4 * N)),
            ("is_plan_valid", lambda: ops.is_plan_valid(prm, X[:3 * N], X[3 * N:6 * N], B=B), f4 * (6 * N + 1))]
    for name, fn, nbytes in rows:
        ms = bench(fn)
        print(f"ld = B + {64 * k:5d}  {name:22s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.3f} TB/s (algorithmic)", flush=True)
    del X, p0, v0, goal
    torch.cuda.empty_cache()
for k in (0, 1, 3, 17):
    out = subprocess.run([os.path.join(ROOT, "tools", "probes", "probe_rows"), "270", "20", str(64 * k)], stdout=subprocess.PIPE, text=True).stdout
    print(out, flush=True)
