"""Developer probe (GPU box): BASELINE config 3 (horizon 50, K = 16 spheres) -- fused rollout+obstacle kernel,
both workgroup shapes (3 / 8 wavefronts), against the plain rollout of the same horizon."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
g = torch.Generator(device=dev); g.manual_seed(1)
def bench(name, fn, bytes_, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:44s} {ms*1e3:9.1f} us  {bytes_/ms/1e9:8.3f} TB/s (algorithmic)", flush=True)
N3 = 50; prm3 = Params.reference_defaults(horizon=N3)
for K in (16, 64):
    sph = torch.rand(K, 4, device=dev, generator=g) * 10
    for B3 in (1024, 8192, 65536, 1 << 20):
        T3 = torch.randn(3 * N3, B3, device=dev, generator=g) * 2; T3[2::3] += 14.715
        q0 = torch.rand(3, B3, device=dev, generator=g) * 40 - 20; w0 = torch.rand(3, B3, device=dev, generator=g) * 10 - 5
        gl = torch.rand(3, B3, device=dev, generator=g) * 40 - 20
        cost = torch.empty(B3, device=dev); grad = torch.empty(3 * N3, B3, device=dev)
        bytes3 = 4 * B3 * (6 * N3 + 12)
        for wsel, tag in ((128, "W=3"), (256, "W=8"), (0, "auto")):
            ops.lib.set_rollout_variant(wsel)
            bench(f"cfg3 fused K={K} B={B3} {tag}", lambda: ops.rollout_obstacles(prm3, q0, w0, gl, T3, sph), bytes3)
        ops.lib.set_rollout_variant(0)
        if K == 16:
            bench(f"plain rollout N=50 B={B3}", lambda: ops.rollout_cost_grad(prm3, q0, w0, gl, T3, out=(cost, grad)), 4 * B3 * (6 * N3 + 10))
# register-light (reversible) sweep vs exact-N register sweep inside the fused kernel: horizon 49 takes the former
for Nq in (49, 50):
    prmq = Params.reference_defaults(horizon=Nq)
    sph = torch.rand(16, 4, device=dev, generator=g) * 10
    for Bq in (65536, 1 << 20):
        Tq = torch.randn(3 * Nq, Bq, device=dev, generator=g) * 2; Tq[2::3] += 14.715
        q0 = torch.rand(3, Bq, device=dev, generator=g) * 40 - 20; w0 = torch.rand(3, Bq, device=dev, generator=g) * 10 - 5
        gl = torch.rand(3, Bq, device=dev, generator=g) * 40 - 20
        bench(f"cfg3 fused N={Nq} B={Bq} auto", lambda: ops.rollout_obstacles(prmq, q0, w0, gl, Tq, sph), 4 * Bq * (6 * Nq + 12))
