"""Developer probe (GPU box): median wall time of the one-launch Monte-Carlo (se3mpc_monte_carlo_*: 4096 drones x 33 planning cycles x 15 control +
simulator steps, bench.py's closed_loop leg), float32 and float64, with a checksum of the final positions -- for A/B runs of two builds of the
library (SE3MPC_LIBRARY=...): same checksum = same bits.  `python tools/gpu_probe_monte_carlo.py`."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dart_planner_amd.capi import Params
from dart_planner_amd.control.closed_loop import ClosedLoopMonteCarlo
from dart_planner_amd.ops import Ops
ops=Ops(); dev=ops.be.device
S,cycles,substeps,sim_dt=4096,33,15,0.01
prm=Params.reference_defaults(); cp,sp=ops.lib.controller_default_params(), ops.lib.simulator_default_params()
res={}
for name,dtype in (("f32",torch.float32),("f64",torch.float64)):
    g=torch.Generator(device=dev); g.manual_seed(5)
    p0=torch.tensor([0.0,0.0,2.0],dtype=dtype,device=dev).repeat(S,1)+0.2*torch.randn(S,3,dtype=dtype,device=dev,generator=g)
    v0=0.3*torch.randn(S,3,dtype=dtype,device=dev,generator=g)
    goal=torch.tensor([8.0,0.0,5.0],dtype=dtype,device=dev).repeat(S,1).contiguous()
    wind=torch.randn(S,3,dtype=dtype,device=dev,generator=g).contiguous()
    mc=ClosedLoopMonteCarlo(ops,prm,cp,sp)
    run=lambda: mc.run_fused(p0,v0,goal,cycles,substeps,sim_dt,wind=wind)["pos"]
    pos=run(); torch.cuda.synchronize()
    ts=[]
    for _ in range(7):
        torch.cuda.synchronize(); t0=time.perf_counter(); pos=run(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    print(name, "one-launch Monte-Carlo %.3f ms"%(np.median(ts)*1e3), "checksum", float(pos.double().sum()))
