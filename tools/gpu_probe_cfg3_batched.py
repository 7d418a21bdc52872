"""Developer probe (GPU box): BASELINE config 3 one-shot (se3mpc_rollout_obstacles_*: rollout + cost + gradient + sphere residuals) at the
batched size of bench.py's `configs.cfg3.batched` leg (64 x 8192), by workgroup shape (3 / 4 / 8 wavefronts), residual form (packed VALU / + 2048: matrix core) and rollout form (exact-N registers /
register-light reversible sweep).  `python3 tools/gpu_probe_cfg3_batched.py [nbatch] [B] [N]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50; prm = Params.reference_defaults(horizon=N)
g = torch.Generator(device=dev); g.manual_seed(1)
sph = torch.cat([torch.round(torch.rand(16, 3, device=dev, generator=g) * 30) / 2 - 3.75, torch.ones(16, 1, device=dev)], dim=1)
T = torch.randn(nb, 3 * N, B, device=dev, generator=g) * 2; T[:, 2::3] += 14.715
q0 = torch.rand(nb, 3, B, device=dev, generator=g) * 4 - 2; w0 = torch.rand(nb, 3, B, device=dev, generator=g) * 10 - 5
gl = torch.rand(nb, 3, B, device=dev, generator=g) * 4 - 2


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cost = torch.empty(nb, B, device=dev); grad = torch.empty_like(T); cmin = torch.empty(nb, B, device=dev); viol = torch.empty(nb, B, device=dev)
VARIANTS = (("registers, auto shape", 0), ("registers, 3 wavefronts", 128), ("registers, 8 wavefronts", 256), ("registers, 4 wavefronts", 384),
            ("registers, 3 wavefronts, matrix-core residuals", 128 + 2048), ("registers, 4 wavefronts, matrix-core residuals", 384 + 2048), ("registers, 8 wavefronts, matrix-core residuals", 256 + 2048),
            ("reversible sweep, 3 wavefronts", 3 + 128), ("reversible sweep, 8 wavefronts", 3 + 256), ("reversible sweep, 4 wavefronts", 3 + 384))
# three interleaved rounds (the first timing of a process runs on a cold clock / page state: the same kernel measured 85 and 110 us in one list)
times = {name: [] for name, _ in VARIANTS}
for _ in range(3):
    for name, var in VARIANTS:
        ops.lib.set_rollout_variant(var)
        times[name].append(timed(lambda: ops.rollout_obstacles_batched(prm, q0, w0, gl, T, sph, cost, grad, cmin, viol)))
bytes_ = 4 * (6 * N + 12) * B * nb
for name, _ in VARIANTS:
    t = sorted(times[name])[1]
    print(f"cfg3 one-shot {nb} x {B} x N={N}, {name}: {t:.1f} us per launch (median of 3: {' '.join(f'{x:.1f}' for x in times[name])}), {bytes_ / t / 1e6:.2f} TB/s algorithmic")
ops.lib.set_rollout_variant(0)
