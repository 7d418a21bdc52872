"""Developer probe (GPU box): the benchmarked launch (64 batches x 8192 rollouts x horizon 30, rollout + cost + gradient + keys) with the batch
stride `ld` padded beyond B -- does the 32 KiB row stride of ld = 8192 alias HBM channels for this kernel as it does for a bare copy
(profiles/r03_lane_ld_probe.txt)?  Ring of 128 batches (two launches' worth, > Infinity Cache), HIP events, warm card.
`python3 tools/gpu_probe_headline_ld.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device; be = ops.be
B, N, S, ring = 8192, 30, 64, 128
prm = Params.reference_defaults(horizon=N)
g = torch.Generator(device=dev); g.manual_seed(3)
for pad in (0, 16, 32, 64, 96, 128, 192, 1040, 0):
    ld = B + pad
    p0 = torch.rand(ring, 3, ld, device=dev, generator=g) * 40 - 20; v0 = torch.rand(ring, 3, ld, device=dev, generator=g) * 10 - 5
    goal = torch.rand(ring, 3, ld, device=dev, generator=g) * 40 - 20
    T = torch.randn(ring, 3 * N, ld, device=dev, generator=g) * 2; T[:, 2::3] += 14.715
    cost = torch.empty(ring, ld, device=dev); grad = torch.empty(ring, 3 * N, ld, device=dev)
    keys = torch.zeros(ring, (B + 63) // 64, dtype=torch.int64, device=dev)
    stream = be.stream()

    def launch(i):
        s0 = (i % (ring // S)) * S
        ops.lib.call("rollout_cost_grad_batched", "f32", B, ld, S, be.ptr(p0[s0]), be.ptr(v0[s0]), be.ptr(goal[s0]), be.ptr(T[s0]), be.ptr(cost[s0]),
                     be.ptr(grad[s0]), be.ptr(keys[s0]), 0, stream, params=prm)
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.08:
        for _ in range(8): launch(i); i += 1
        torch.cuda.synchronize()
    reps = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): launch(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"ld = B + {pad}: {us:.2f} us per 64-batch launch, {760 * B * S / us / 1e6:.2f} TB/s algorithmic ({760 * B * S / us / 1e6 / 8:.3f} of peak)", flush=True)
    del p0, v0, goal, T, cost, grad, keys
    torch.cuda.empty_cache()
