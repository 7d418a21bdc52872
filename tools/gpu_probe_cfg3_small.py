"""Developer probe (GPU box, under `rocprofv3 --kernel-trace --stats`): kernel time of config 3 at its named batch
(8192 x horizon 50 x 16 spheres), both workgroup shapes, and the plain horizon-50 rollout."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
g = torch.Generator(device=dev); g.manual_seed(1)
B3 = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N3 = 50; prm3 = Params.reference_defaults(horizon=N3)
sph = torch.rand(16, 4, device=dev, generator=g) * 10
T3 = torch.randn(3 * N3, B3, device=dev, generator=g) * 2; T3[2::3] += 14.715
q0 = torch.rand(3, B3, device=dev, generator=g) * 40 - 20; w0 = torch.rand(3, B3, device=dev, generator=g) * 10 - 5
gl = torch.rand(3, B3, device=dev, generator=g) * 40 - 20
cost = torch.empty(B3, device=dev); grad = torch.empty(3 * N3, B3, device=dev)
for wsel in (128, 256):
    ops.lib.set_rollout_variant(wsel)
    for _ in range(50): ops.rollout_obstacles(prm3, q0, w0, gl, T3, sph)
ops.lib.set_rollout_variant(0)
for _ in range(50): ops.rollout_cost_grad(prm3, q0, w0, gl, T3, out=(cost, grad))
torch.cuda.synchronize()
print("done")
