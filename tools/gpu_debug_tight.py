import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import parity_checks as pc
from dart_planner_amd.ops import Ops, TorchBackend
from dart_planner_amd.capi import Params
from oracle import se3mpc_oracle as orc
ops = Ops(TorchBackend("cuda:0"))
N,B=20,48
rng=np.random.default_rng(5)
prm=Params.reference_defaults(horizon=N,pgtol=1e-9,ftol=1e-8,max_iterations=40)
cfg=pc.oracle_cfg(prm)
p0,v0,goal,_=pc.random_batch(rng,B,N)
t=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
out=ops.solve(prm,t(p0),t(v0),t(goal)); info=ops.info_to_host(out['info']); X=out['x'].cpu().numpy()
for i in range(B):
    xr,ir=orc.solve(p0[i],v0[i],goal[i],cfg)
    flag = (int(info['nit'][i]),int(info['nfev'][i]),int(info['status'][i]))==(ir['nit'],ir['nfev'],ir['status'])
    print(i, (int(info['nit'][i]),int(info['nfev'][i]),int(info['status'][i]),int(info['task'][i])), (ir['nit'],ir['nfev'],ir['status']), 'ok' if flag else 'MISMATCH', '%.2e'%np.max(np.abs(X[i]-xr)), repr(p0[i].tolist()), repr(v0[i].tolist()), repr(goal[i].tolist()) if not flag else '')
