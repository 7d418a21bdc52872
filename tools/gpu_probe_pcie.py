"""Developer probe (GPU box): the PCIe-inclusive rate of the benchmarked step -- inputs start in pinned host memory, cost and
gradient end in pinned host memory (never bench.py's `value`, which starts with inputs resident in HBM)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
N = 30; prm = Params.reference_defaults(horizon=N)
out = {}
for B in (8192, 65536, 1 << 20):
    h = [torch.randn(r, B).pin_memory() for r in (3, 3, 3, 3 * N)]
    h[3][2::3] += 14.715
    d = [torch.empty_like(t, device=dev) for t in h]
    cost, grad = torch.empty(B, device=dev), torch.empty(3 * N, B, device=dev)
    hc, hg = torch.empty(B).pin_memory(), torch.empty(3 * N, B).pin_memory()
    def step():
        for a, b in zip(d, h): a.copy_(b, non_blocking=True)
        ops.rollout_cost_grad(prm, d[0], d[1], d[2], d[3], out=(cost, grad))
        hc.copy_(cost, non_blocking=True); hg.copy_(grad, non_blocking=True)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): step()
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / reps
    bytes_ = 4 * B * (6 * N + 10)
    out[str(B)] = dict(ms_per_step=el * 1e3, rollouts_per_s=B / el, host_link_GB_per_s=bytes_ / el / 1e9)
print(json.dumps(out, indent=1))
