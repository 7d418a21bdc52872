import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, numpy as np
from dart_planner_amd.ops import Ops
from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper
ops = Ops(); dev = ops.be.device
rng = np.random.default_rng(5)
m = ExplicitGeometricMapper(resolution=0.2, max_range=50.0, ops=ops)
for _ in range(8):
    m.add_obstacle(rng.uniform(-8, 8, 3) + [0, 0, 2], float(rng.uniform(0.5, 1.5)))
n = 360
ang = 2 * np.pi * np.arange(n) / n
dirs = np.stack([np.cos(ang), np.sin(ang), np.zeros(n)], 1)
hit = rng.random(n) < 0.1
dist = np.where(hit, rng.uniform(2.0, 20.0, n), 50.0)
org = np.tile([0.3, 0.1, 2.0], (n, 1))
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = m.map.update_rays(org, dirs, dist, hit.astype(np.int32))
    torch.cuda.synchronize()
    print(i, "ms", round((time.perf_counter() - t0) * 1e3, 3), r, m.map.capacity, flush=True)
