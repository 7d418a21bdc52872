"""Developer probe (GPU box): does a leg's device time depend on how long the card has been busy?  Runs one kernel back to back for ~0.4 s and
prints its mean HIP-event time per ~10 ms window (the one-shot config-3 launch of 64 batches, the 8192-problem batched solve, the benchmarked
64-batch rollout launch).  `python3 tools/gpu_probe_warmup.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
g = torch.Generator(device=dev); g.manual_seed(1)


def series(name, fn, per_window, windows=40, idle_s=0.5):
    fn(); torch.cuda.synchronize()
    time.sleep(idle_s)                                            # the card idles, as between two legs of bench.py
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(windows + 1)]
    ev[0].record()
    for w in range(windows):
        for _ in range(per_window): fn()
        ev[w + 1].record()
    torch.cuda.synchronize()
    t = [ev[w].elapsed_time(ev[w + 1]) * 1e3 / per_window for w in range(windows)]
    cum = 0.0; marks = []
    for w in range(windows):
        cum += t[w] * per_window / 1e3
        marks.append(f"{cum:.0f}ms:{t[w]:.1f}")
    print(f"{name}: us per launch by window (elapsed:mean) " + " ".join(marks), flush=True)


# config 3 one-shot, 64 x 8192
nb, B, N = 64, 8192, 50; prm = Params.reference_defaults(horizon=N)
sph = torch.cat([torch.round(torch.rand(16, 3, device=dev, generator=g) * 30) / 2, torch.ones(16, 1, device=dev)], dim=1)
T = torch.randn(nb, 3 * N, B, device=dev, generator=g) * 2; T[:, 2::3] += 14.715
q0 = torch.rand(nb, 3, B, device=dev, generator=g) * 4 - 2; w0 = torch.rand(nb, 3, B, device=dev, generator=g) * 10 - 5
gl = torch.rand(nb, 3, B, device=dev, generator=g) * 4 - 2
cost = torch.empty(nb, B, device=dev); grad = torch.empty_like(T); cmin = torch.empty(nb, B, device=dev); viol = torch.empty(nb, B, device=dev)
series("cfg3 one-shot 64 x 8192", lambda: ops.rollout_obstacles_batched(prm, q0, w0, gl, T, sph, cost, grad, cmin, viol), 60)
# plain rollout, 64 x 8192 x N = 30
N = 30; prm30 = Params.reference_defaults(horizon=N)
T3 = torch.randn(nb, 3 * N, B, device=dev, generator=g) * 2; T3[:, 2::3] += 14.715
grad3 = torch.empty_like(T3)
series("rollout 64 x 8192 x N=30", lambda: ops.rollout_cost_grad_batched(prm30, q0, w0, gl, T3, cost, grad3), 150)
# batched solve 8192 x N = 30
p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20; v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5; goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
out = ops.solve(prm30, p0, v0, goal)
series("solve 8192 x N=30 (both tiers)", lambda: ops.solve(prm30, p0, v0, goal, out=out), 100)
