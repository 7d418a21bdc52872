"""GPU box: many random voxel-map scenes against the oracle's dict walk, bit for bit (tests/voxel_checks.check_random_scenes).
usage: python tools/gpu_fuzz_voxel.py [scenes] [rays]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import voxel_checks as vc
from dart_planner_amd.ops import Ops, TorchBackend
ops = Ops(TorchBackend("cuda:0"))
scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0 = time.time()
for seed in range(5):
    vc.check_random_scenes(ops, n_scenes=scenes // 5, n_rays=rays, seed=100 + seed)
print(json.dumps(dict(scenes=scenes // 5 * 5, scans=2 * (scenes // 5 * 5), rays_per_scan=rays, result="every table (keys, counts, float64 probabilities), "
                      "500 point queries and 16 trajectory checks per scene identical to the oracle", seconds=round(time.time() - t0, 1))))
