import sys, os, json, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import parity_checks as pc
from dart_planner_amd.ops import Ops, TorchBackend
ops = Ops(TorchBackend("cuda:0"))
data=np.load(os.path.join(ROOT,'tests/golden/solve_cases.npz')); meta=json.load(open(os.path.join(ROOT,'tests/golden/solve_cases.json')))
for c in meta['cases']:
    k=c['key']
    if k not in sys.argv[1:]: continue
    prm=pc.solve_params(c)
    t=lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).to("cuda:0")
    out=ops.solve(prm,t(data[k+'p0']),t(data[k+'v0']),t(data[k+'goal']))
    x=out['x'].cpu().numpy()[0]; info=ops.info_to_host(out['info'])[0]
    err=np.abs(x-data[k+'x']); i=np.argsort(err)[-8:]
    print(k, info, 'max err', err.max())
    for j in i: print('  idx',j,'block',j//(3*c['N']),'gpu',repr(x[j]),'ref',repr(data[k+'x'][j]), 'x0', data[k+'x0'][j])
