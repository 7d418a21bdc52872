"""Developer probe (GPU box): the voxel-map kernels at the sizes of the reference's planning loop
(cloud/main_improved_threelayer.py: 360-ray scans, 20 m local grid at 0.2 m = 10^6 cells, 8192 candidate plans),
with the oracle's dict walk (= the reference's arithmetic) timed beside them on one host core."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from dart_planner_amd.ops import Ops
from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper, SensorObservation
from oracle import mapper_oracle as mo

ops = Ops(); dev = ops.be.device
rng = np.random.default_rng(5)
m = ExplicitGeometricMapper(resolution=0.2, max_range=50.0, ops=ops)
orc = mo.VoxelMap(0.2, 50.0)
for _ in range(8):
    c, r = rng.uniform(-8, 8, 3) + [0, 0, 2], float(rng.uniform(0.5, 1.5))
    m.add_obstacle(c, r); orc.add_obstacle(c, r)

def scan(origin, n=360):
    dirs = mo.lidar_directions(n)
    hits = np.where(rng.random(n) < 0.1, rng.uniform(2.0, 20.0, n), np.nan)
    return [SensorObservation(position=np.array(origin, float), direction=d, hit_distance=(None if np.isnan(h) else float(h)),
                              max_range=50.0, timestamp=0.0) for d, h in zip(dirs, hits)], dirs, hits

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def dev_timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

out = {}
# ---- update_map: one 360-ray scan (wall time of the host call incl. uploads; kernels alone from device events)
obs, dirs, hits = scan([0.0, 0.0, 2.0])
res = m.update_map(obs)
t0 = time.perf_counter(); n_o = orc.update_map([o.position for o in obs], dirs, [None if np.isnan(h) else float(h) for h in hits], [50.0] * len(obs)); t_cpu = time.perf_counter() - t0
obs2, dirs2, hits2 = scan([0.3, 0.1, 2.0])
t_gpu = timed(lambda: m.update_map(obs2), 5)
o_ = np.array([o.position for o in obs2]); d_ = dirs2; dist_ = np.array([min(h if h else 50.0, 50.0) if not np.isnan(h) else 50.0 for h in hits2]); hit_ = (~np.isnan(hits2)).astype(np.int32)
t_kern = dev_timed(lambda: m.map.update_rays(o_, d_, dist_, hit_), 5)
out["update_map_360_rays"] = dict(voxel_updates=res["updated_voxels"], gpu_call_ms=t_gpu * 1e3, gpu_update_rays_ms=t_kern * 1e3,
                                  cpu_oracle_ms=t_cpu * 1e3, voxels=len(m.map), capacity=m.map.capacity)
# ---- point queries: 10^6 positions resident on the device
P = torch.rand(1_000_000, 3, device=dev, dtype=torch.float64) * 40 - 20
t_q = dev_timed(lambda: m.map.query(P), 20)
Ph = P[:20000].cpu().numpy()
t0 = time.perf_counter(); orc.query(Ph); t_cq = (time.perf_counter() - t0) / len(Ph)
out["query_1M"] = dict(gpu_ms=t_q * 1e3, gpu_queries_per_s=1e6 / t_q, cpu_oracle_queries_per_s=1 / t_cq)
# ---- the planning loop's obstacle refresh: 20 m local grid at 0.2 m (10^6 cells) -> 20 spheres
centre = np.array([0.3, 0.1, 2.0])
t_f = dev_timed(lambda: m.map.local_spheres(centre, 20.0, 0.6, 20, 1.0), 20)
t_fw = timed(lambda: m.local_obstacle_spheres(centre, 20.0, 0.6, 20, 1.0), 20)
t_ref_shape = timed(lambda: m.get_local_occupancy_grid(centre, 20.0), 3)
out["local_spheres_100cubed"] = dict(gpu_kernels_ms=t_f * 1e3, gpu_call_with_readback_ms=t_fw * 1e3,
                                     gpu_reference_shaped_grid_call_ms=t_ref_shape * 1e3,
                                     cpu_oracle_est_ms=1e6 * t_cq * 1e3, cells=100 ** 3)
# ---- trajectory safety of a batch of plans (8192 x 30 positions x 7 probes)
T = torch.rand(8192, 30, 3, device=dev, dtype=torch.float32) * 20 - 10
t_s = dev_timed(lambda: m.map.trajectories_safe(T, margin=1.0, threshold=0.6), 20)
Th = T[:40].cpu().numpy().astype(np.float64)
t0 = time.perf_counter(); [orc.is_trajectory_safe(p, 1.0, 0.6) for p in Th]; t_cs = (time.perf_counter() - t0) / len(Th)
out["trajectories_safe_8192x30"] = dict(gpu_ms=t_s * 1e3, gpu_trajectories_per_s=8192 / t_s, cpu_oracle_trajectories_per_s=1 / t_cs)
print(json.dumps(out, indent=1))
