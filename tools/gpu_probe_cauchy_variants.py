"""Developer probe (GPU box): HIP-event time of the batched solve with the closed-form Cauchy point of the first iteration (default) against the
published sequential search (se3mpc_set_solver_variant(1): SciPy's accumulation, every breakpoint of the first iteration walked one by one).
`python tools/gpu_probe_cauchy_variants.py`."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops
ops=Ops(); dev=ops.be.device
for N,B in ((30,8192),(6,8192),(30,65536),(50,8192)):
    prm=Params.reference_defaults(horizon=N)
    g=torch.Generator(device=dev); g.manual_seed(1)
    p0=torch.rand(B,3,device=dev,generator=g)*40-20; v0=torch.rand(B,3,device=dev,generator=g)*10-5; goal=torch.rand(B,3,device=dev,generator=g)*40-20
    for var in (0,1,0,1):
        ops.lib.set_solver_variant(var)
        o=None
        for _ in range(5): o=ops.solve(prm,p0,v0,goal,out=o)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): o=ops.solve(prm,p0,v0,goal,out=o)
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} B={B} variant {var}: {e0.elapsed_time(e1)*1e3/50:.1f} us per batch solve")
ops.lib.set_solver_variant(0)
