"""Developer probe (one-GPU box): the RCCL calls of the multi-GPU path on a world of ONE rank -- backend "nccl" initialises with
device_id, int64 all-reduce(MIN) of sign-flipped packed keys, float64 all-reduce(MAX / SUM), broadcast -- i.e. dtype / op support and the
init path of dart_planner_amd.distributed, not the exchange itself (that needs >= 2 GPUs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from dart_planner_amd import distributed as D
import numpy as np
keys = torch.from_numpy(np.array([(0x80000000 | 77) << 32 | 5, (0x7F000000) << 32 | 9], dtype=np.uint64).view(np.int64).copy()).to(dev)
ref = keys.clone()
_orig = dist.get_world_size
dist.get_world_size = lambda *a, **k: 2          # force the collective branch although the world has one rank
D.allreduce_min_keys(keys)
s = torch.arange(4, dtype=torch.float64, device=dev) + 1
m = D.allreduce_population_mean(s.clone())
dist.get_world_size = _orig
x = torch.ones(1, dtype=torch.float64, device=dev); dist.all_reduce(x, op=dist.ReduceOp.MAX)
b = torch.arange(9, dtype=torch.float32, device=dev); dist.broadcast(b, src=0)
torch.cuda.synchronize()
assert torch.equal(keys, ref) and torch.allclose(m, s[:-1] / s[-1]) and float(x) == 1.0
print("RCCL single-rank probe ok:", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()
