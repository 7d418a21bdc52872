"""Developer probe (GPU box): se3mpc_transpose both ways at decision-vector shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.ops import Ops
ops = Ops(); dev = ops.be.device
for dt in (torch.float32, torch.float64):
    for rows, cols in ((270, 1 << 20), (1 << 20, 270), (270, 8192), (8192, 270), (54, 1 << 20), (450, 1 << 19)):
        a = torch.randn(rows, cols, device=dev, dtype=dt)
        for _ in range(3): ops.transpose(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.transpose(a)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{str(dt)[6:]:8s} {rows:8d} x {cols:8d}: {ms*1e3:9.1f} us  {2*a.numel()*a.element_size()/ms/1e9:7.3f} TB/s", flush=True)
