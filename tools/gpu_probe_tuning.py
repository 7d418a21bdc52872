"""Developer probe (GPU box): A/B the rollout kernel's tuning flags, interleaved rounds in one process."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); lib = ops.lib; dev = ops.be.device
N = 30; prm = Params.reference_defaults(horizon=N)
res = {}
def mk(nb, B):
    g = torch.Generator(device=dev); g.manual_seed(1)
    p0 = torch.rand(nb, 3, B, device=dev, generator=g) * 40 - 20; v0 = torch.rand(nb, 3, B, device=dev, generator=g) * 10 - 5
    goal = torch.rand(nb, 3, B, device=dev, generator=g) * 40 - 20
    T = torch.randn(nb, 3 * N, B, device=dev, generator=g) * 2; T[:, 2::3] += 14.715
    return p0, v0, goal, T, torch.empty(nb, B, device=dev), torch.empty(nb, 3 * N, B, device=dev), torch.zeros(nb, (B + 63) // 64, dtype=torch.int64, device=dev)
for name, nb, B, reps in (("fused_87x8192", 87, 8192, 30), ("fused_348x8192", 348, 8192, 10), ("fused_1024x8192", 1024, 8192, 5), ("single_1M", 1, 1 << 20, 30), ("single_4M", 1, 1 << 22, 10)):
    bufs = mk(nb, B)
    times = {}
    for rnd in range(5):
        for flags in range(8):
            lib.set_rollout_variant(1 + 8 * (flags + 1))
            p0, v0, goal, T, cost, grad, keys = bufs
            ops.rollout_cost_grad_batched(prm, p0, v0, goal, T, cost, grad, wave_keys=keys)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.rollout_cost_grad_batched(prm, p0, v0, goal, T, cost, grad, wave_keys=keys)
            e1.record(); torch.cuda.synchronize()
            times.setdefault(flags, []).append(e0.elapsed_time(e1) / reps)
    lib.set_rollout_variant(0)
    byt = 4 * (6 * N + 10) * B * nb
    res[name] = {f: dict(ms_median=float(np.median(t)), ms_min=float(np.min(t)), TBps=byt / (np.median(t) * 1e-3) / 1e12, us_per_8192=float(np.median(t)) * 1e3 * 8192 / (B * nb)) for f, t in times.items()}
    for f in range(8):
        print(name, "flags", f, res[name][f], flush=True)
    del bufs
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe_tuning.json"), "w"), indent=1)
