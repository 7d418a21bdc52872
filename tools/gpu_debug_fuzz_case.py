"""Developer probe: the re-weighted option set of tools/gpu_fuzz_solver.py -- list every problem whose counts differ from SciPy's with the
kernel's and SciPy's (nit, nfev, status, fun), the position gap, and the same under the published Cauchy search (solver variant 1)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity_checks as pc
from dart_planner_amd.capi import Params
from dart_planner_amd.ops import Ops, TorchBackend
from oracle import se3mpc_oracle as orc
ops = Ops(TorchBackend("cuda:0"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
N, ov, seed = 20, dict(position_weight=3.0, velocity_weight=40.0, thrust_weight=2.5, acceleration_weight=0.2), 1000 + 17 * 14
for dt in (np.float64, np.float32):
    h = pc.Harness(ops, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0"), lambda a: a.detach().cpu().numpy(), dt)
    rng = np.random.default_rng(seed)
    prm = Params.reference_defaults(horizon=N, **ov)
    cfg = pc.oracle_cfg(prm)
    p0, v0, goal, _ = pc.random_batch(rng, B, N)
    res = {}
    for var in (0, 1):
        ops.lib.set_solver_variant(var)
        out = ops.solve(prm, h.prob(p0), h.prob(v0), h.prob(goal))
        res[var] = (ops.info_to_host(out["info"]).copy(), h.to_host(out["x"]).astype(float))
    ops.lib.set_solver_variant(0)
    for i in range(B):
        r = lambda a: a[i].astype(dt).astype(float)
        xr, ir = orc.solve(r(p0), r(v0), r(goal), cfg)
        row = {}
        for var in (0, 1):
            info, X = res[var]
            got = (int(info["nit"][i]), int(info["nfev"][i]), int(info["status"][i]))
            row[var] = dict(counts=got, fun=float(info["fun"][i]), gap=float(np.max(np.abs(X[i, :3 * N] - xr[:3 * N]))))
        if row[0]["counts"] != (ir["nit"], ir["nfev"], ir["status"]) and row[0]["gap"] > 1e-5:
            print(json.dumps(dict(dtype=np.dtype(dt).name, problem=i, scipy=(ir["nit"], ir["nfev"], ir["status"]), scipy_fun=float(orc.objective(xr[None], r(goal)[None], cfg)[0]),
                                  closed_form=row[0], published=row[1])), flush=True)
print("done")
