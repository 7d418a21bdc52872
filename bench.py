#!/usr/bin/env python3
"""bench.py -- SE(3) rollouts/s (horizon 30, batch 8192 per GPU) on N MI355X + roofline + CPU baseline.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU,
RCCL).  Rank 0 prints ONE JSON line.

A *step* is one pass of the hot path over one batch: the shooting-form rollout + running/terminal
cost + exact thrust gradient (SURVEY.md section 8d "canonical rollout", 4*(6N+10) B per rollout) of
``batch`` = 8192 synthetic trajectories resident in HBM, with the batch argmin folded into the same
kernel.  The batch is fixed per GPU (weak scaling); the only cross-GPU exchange is ONE bucketed
RCCL all-reduce(MIN) of the K packed (cost, index) keys at the end of the timed region
(SURVEY.md section 8e).  Steps cycle through a ring of distinct input/output batches larger than the
256 MiB Infinity Cache, so every step streams its operands from HBM.

Two legs time the same K steps (SURVEY.md section 7 "report both"):
  * primary (`value`, `roofline`): the steps are independent batches (a Monte-Carlo sweep, many
    planners), so `--steps-per-launch` S = 64 of them go into ONE multi-batch kernel launch (grid.y);
  * `single_launch`: one kernel launch per 8192-rollout step (sequentially dependent sampling
    iterations of one planner): 6 MB per launch = 1 us of HBM time, i.e. latency-bound.
Launches are captured once into a hipGraph and replayed inside the timed region.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8192, help="rollouts per GPU per step (BASELINE.json metric: 8192)")
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--ring", type=int, default=0, help="distinct batches cycled through (0 = enough to exceed 512 MiB)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the p95 solve-latency leg")
    ap.add_argument("--no-obstacle-source", action="store_true", help="skip the voxel-map leg (mapper -> planner obstacles)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--steps-per-launch", type=int, default=64,
                    help="independent 8192-rollout steps issued as ONE multi-batch kernel launch (1 = one launch per step)")
    ap.add_argument("--no-single", action="store_true", help="skip the one-launch-per-step leg")
    ap.add_argument("--variant", type=int, default=0, help="se3mpc_set_rollout_variant (0 = auto)")
    ap.add_argument("--sweep", action="store_true", help="also time saturating batch sizes (extra keys)")
    return ap.parse_args()


def make_ring(torch, dev, B, N, ring, seed):
    """cfg-2 distribution of SURVEY.md section 8d: p0 ~ U(-20,20)^3, v0 ~ U(-5,5)^3, goal ~ U(-20,20)^3,
    T = (0,0,14.715) + N(0, 2^2) clipped to the thrust box (planner.py:390-400)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    p0 = torch.rand(ring, 3, B, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(ring, 3, B, device=dev, generator=g) * 10 - 5
    goal = torch.rand(ring, 3, B, device=dev, generator=g) * 40 - 20
    T = torch.randn(ring, 3 * N, B, device=dev, generator=g) * 2
    T[:, 2::3] += 14.715
    txy = 25.0 * math.sin(math.pi / 4)
    T[:, 0::3].clamp_(-txy, txy); T[:, 1::3].clamp_(-txy, txy); T[:, 2::3].clamp_(2.0, 25.0)
    cost = torch.empty(ring, B, device=dev)
    grad = torch.empty(ring, 3 * N, B, device=dev)
    return p0, v0, goal, T, cost, grad


def _cpu_worker(args):
    """One host core: the oracle's batched NumPy rollout+cost+grad in a loop for `seconds`."""
    B, N, seconds, seed = args
    import time as _t
    import numpy as _np
    from oracle import se3mpc_oracle as orc
    cfg = orc.OracleConfig(prediction_horizon=N)
    rng = _np.random.default_rng(seed)
    p0, v0, goal = rng.uniform(-20, 20, (B, 3)), rng.uniform(-5, 5, (B, 3)), rng.uniform(-20, 20, (B, 3))
    T = rng.normal(0, 2, (B, N, 3)) + [0, 0, cfg.hover_thrust]
    orc.rollout_cost_grad(p0, v0, goal, T, cfg)
    n, t0 = 0, _t.perf_counter()
    while _t.perf_counter() - t0 < seconds:
        orc.rollout_cost_grad(p0, v0, goal, T, cfg)
        n += 1
    return n, _t.perf_counter() - t0


def cpu_baseline(B, N, seconds):
    """The oracle's batched NumPy restatement of the same rollout+cost+gradient (float64) on the host cores of
    this box: one worker process per core of the box's CPU share (16 for a one-GPU box), each looping over its
    own 8192-trajectory batch; plus the reference-shaped leg (one problem per call, Python loops as in
    planner.py:516-580) on one core.  MUST run before this process touches the GPU: the workers are spawned."""
    import concurrent.futures as cf
    import multiprocessing as mp
    from oracle import se3mpc_oracle as orc
    Bs = min(B, 8192)
    cores = max(1, min(16, os.cpu_count() or 1))
    single = _cpu_worker((Bs, N, min(3.0, seconds / 4), 1))
    try:
        with cf.ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_cpu_worker, [(Bs, N, seconds, 10 + i) for i in range(cores)]))
    except Exception as e:                       # a box that forbids worker processes: report the single-core figure
        res, cores = [single], 1
        print(f"[bench] cpu_baseline: worker pool unavailable ({e!r}); single core only", file=sys.stderr)
    total = sum(n * Bs / el for n, el in res)
    cfg = orc.OracleConfig(prediction_horizon=N)
    rng = np.random.default_rng(1)
    p0, v0, goal = rng.uniform(-20, 20, 3), rng.uniform(-5, 5, 3), rng.uniform(-20, 20, 3)
    x = orc.straight_line_init(p0, v0, goal, cfg)
    m, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < min(3.0, seconds / 3):
        orc.objective_loops(x, goal, cfg); orc.gradient_loops(x, goal, cfg)
        m += 1
    el1 = time.perf_counter() - t1
    # the reference-shaped SOLVE (plan_trajectory's arithmetic: per-step Python loops in f and g, SciPy L-BFGS-B), one
    # problem per call, one core: what the p95 solve latency of the `solve` leg stands beside (~5 ms at N=30)
    ts = []
    for i in range(40):
        gi = rng.uniform(-5, 5, 3); gi[2] = abs(gi[2]) + 0.5
        t2 = time.perf_counter()
        orc.plan_reference_shaped(np.array([0.0, 0.0, 1.0]), np.zeros(3), gi, cfg)
        ts.append((time.perf_counter() - t2) * 1e3)
    return dict(value=total, unit="rollouts/s", cores=cores, kind="port",
                sample=f"{cores} worker processes x {seconds:.0f} s of the oracle's batched NumPy rollout+cost+grad (float64), "
                       f"{Bs} trajectories per pass, horizon {N} ({sum(n for n, _ in res)} passes in all)",
                single_core_value=single[0] * Bs / single[1], reference_shaped_evals_per_s=m / el1,
                reference_shaped_solve_ms=dict(horizon=N, calls=len(ts), p50=float(np.percentile(ts, 50)), p95=float(np.percentile(ts, 95))),
                host_cpus=os.cpu_count())


def timed_region(torch, dist, world, dev, K, launch_all, graph, keys, allreduce_min_keys):
    """The contract's timed region: barrier + synchronize, K steps, [one bucketed all-reduce(MIN) of the
    K keys], synchronize + barrier; MAX over ranks.  Returns (elapsed_s, device_ms between the HIP events
    that bracket the K steps on the launch stream)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    if graph is not None:
        graph.replay()
    else:
        launch_all()
    e1.record()
    if world > 1:
        allreduce_min_keys(keys)     # the single exchange: ONE bucketed all-reduce(MIN) of the K packed keys
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), e0.elapsed_time(e1)


def capture(torch, dev, fn):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph.replay()                                              # one untimed replay (uploads the graph)
    torch.cuda.synchronize()
    return graph


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_stats = None
    if world == 1 and not a.no_cpu_baseline:
        cpu_stats = cpu_baseline(a.batch, a.horizon, a.cpu_seconds)      # before any HIP call: it spawns worker processes
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # rehearsal of N ranks on a 1-GPU box (SE3MPC_DIST_BACKEND=gloo) shares the card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from dart_planner_amd.distributed import allreduce_min_keys, init_distributed
    init_distributed(os.environ.get("SE3MPC_DIST_BACKEND", "nccl"), device=dev)
    if a.gpus != world and rank == 0:
        print(f"[bench] --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1", file=sys.stderr)

    from dart_planner_amd.capi import Params
    from dart_planner_amd.ops import Ops, TorchBackend
    ops = Ops(TorchBackend(dev))
    ops.lib.set_rollout_variant(a.variant)
    B, N, K, W, S = a.batch, a.horizon, a.steps, a.warmup, max(1, a.steps_per_launch)
    prm = Params.reference_defaults(horizon=N)
    bytes_per_rollout = 4 * (6 * N + 10)
    slot_bytes = 4 * B * ((9 + 3 * N) + (1 + 3 * N))
    ring = a.ring if a.ring > 0 else max(2, math.ceil(512 * 2 ** 20 / slot_bytes))
    ring = S * max(2, math.ceil(ring / S))                     # whole launches, consecutive launches on disjoint slots
    p0, v0, goal, T, cost, grad = make_ring(torch, dev, B, N, ring, seed=3 + rank)
    per = (B + 63) // 64                                        # wave-key slots per batch
    wave_keys = torch.zeros(max(K, W, S), per, dtype=torch.int64, device=dev)
    keys = torch.full((max(K, 1),), -1, dtype=torch.int64, device=dev)
    base = rank * B

    def launch(i0, n):
        """steps i0 .. i0+n-1 (n <= S consecutive ring slots) as ONE launch; step i's wavefronts write
        their partial argmin keys to wave_keys[i]"""
        s0 = i0 % ring
        if n == 1:
            ops.rollout_cost_grad(prm, p0[s0], v0[s0], goal[s0], T[s0], out=(cost[s0], grad[s0]), wave_keys=wave_keys[i0], index_base=base)
        else:
            ops.rollout_cost_grad_batched(prm, p0[s0:s0 + n], v0[s0:s0 + n], goal[s0:s0 + n], T[s0:s0 + n], cost[s0:s0 + n],
                                          grad[s0:s0 + n], wave_keys=wave_keys[i0:i0 + n], index_base=base)

    def run_steps(nsteps, per_launch, fold=True):
        i = 0
        while i < nsteps:
            n = min(per_launch, nsteps - i)
            n = min(n, ring - (i % ring))
            launch(i, n)
            i += n
        if fold:                                                # one bucketed fold of all steps' wave keys -> keys[step]
            ops.reduce_keys(wave_keys[:nsteps], keys[:nsteps])

    results = {}
    for mode, per_l in (("primary", S), ("single_launch", 1)):
        if mode == "single_launch" and (S == 1 or a.no_single):
            continue
        run_steps(W, per_l, fold=False)                         # untimed warm-up (eager)
        torch.cuda.synchronize()
        graph = None if a.no_graph else capture(torch, dev, lambda: run_steps(K, per_l))
        if world > 1:
            # untimed: the first collective of a given shape pays communicator / kernel set-up (ms); the timed
            # region must only see the steady-state exchange
            allreduce_min_keys(keys.clone())
            dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev), op=dist.ReduceOp.MAX)
        keys.fill_(-1)
        torch.cuda.synchronize()
        elapsed, dev_ms = timed_region(torch, dist, world, dev, K, lambda: run_steps(K, per_l), graph, keys, allreduce_min_keys)
        nlaunch = sum(1 for _ in _launch_sizes(K, per_l, ring))
        kh = keys[:K].cpu().numpy().view(np.uint64)
        results[mode] = dict(elapsed=elapsed, launch_ms=dev_ms / nlaunch, nlaunch=nlaunch, per=per_l, graph=graph is not None,
                             keys_valid=bool(np.all((kh & np.uint64(0xFFFFFFFF)) < np.uint64(world * B))))
        del graph

    solve_stats = None if a.no_solve else solve_leg(torch, ops, dev, B, N, rank, world)
    voxel_stats = None if (a.no_obstacle_source or rank != 0) else obstacle_source_leg(torch, ops, dev)

    if rank == 0:
        r = results["primary"]
        per_launch_rollouts = B * min(S, K)
        achieved = bytes_per_rollout * per_launch_rollouts / (r["launch_ms"] * 1e-3) / 1e9
        traffic = profiled_traffic(B, N, S)
        res = {
            "metric": "SE(3) rollouts/sec (N=30, batch=8192) + p95 solve ms, at 1/2/4/8 MI355X",
            "value": world * B * K / r["elapsed"], "unit": "rollouts/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": r["elapsed"] / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"horizon={N} SE(3) rollout + cost + thrust-gradient (+ fused batch argmin), "
                                   f"batch={B} per GPU per step, {S} independent steps per kernel launch "
                                   f"({'one multi-batch launch = grid.y' if S > 1 else 'one launch per step'}), "
                                   f"ring of {ring} distinct batches in HBM ({ring * slot_bytes / 2 ** 20:.0f} MiB), "
                                   f"{'hipGraph replay' if r['graph'] else 'eager launches'}",
                       "horizon": N, "batch_per_gpu": B, "global_batch": world * B, "steps_per_launch": S, "ring": ring,
                       "parallelism": f"batch-sharded x{world}, one bucketed all-reduce(MIN) of {K} keys"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None if traffic is None else traffic[0],
                         "traffic_source": None if traffic is None else f"profiles/{traffic[1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                         "kernel": "se3mpc::rollout_kernel<float, 30, REG, SPLIT, GRAD>", "kernel_us": r["launch_ms"] * 1e3,
                         "launches": r["nlaunch"], "rollouts_per_launch": per_launch_rollouts,
                         "bytes_per_launch": bytes_per_rollout * per_launch_rollouts,
                         "note": "achieved = algorithmic bytes (4*(6N+10) B/rollout x rollouts per launch) / average "
                                 "launch duration (HIP events around the launches of the timed region on the launch stream)"},
            "keys_valid": r["keys_valid"],
        }
        if "single_launch" in results:
            q = results["single_launch"]
            g1 = bytes_per_rollout * B / (q["launch_ms"] * 1e-3) / 1e9
            res["single_launch"] = {"what": "the same K steps, ONE kernel launch per 8192-rollout step (sequentially dependent iterations)",
                                    "value": world * B * K / q["elapsed"], "ms_per_step": q["elapsed"] / K * 1e3,
                                    "kernel_us": q["launch_ms"] * 1e3, "achieved_GB_per_s": g1, "frac": g1 / HBM_PEAK_GBPS,
                                    "keys_valid": q["keys_valid"]}
        if solve_stats is not None:
            res["solve"] = solve_stats
        if voxel_stats is not None:
            res["obstacle_source"] = voxel_stats
        if a.sweep:
            res["sweep"] = sweep(torch, ops, prm, dev, N)
        if cpu_stats is not None:
            res["cpu_baseline"] = cpu_stats
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch_sizes(nsteps, per_launch, ring):
    i = 0
    while i < nsteps:
        n = min(per_launch, nsteps - i, ring - (i % ring))
        yield n
        i += n


def solve_leg(torch, ops, dev, B, N, rank, world):
    """p95 of the per-call wall time (np.percentile convention of the reference's
    tests/test_real_time_latency.py:296-304; >= 200 timed calls after >= 20 warm-ups, device
    synchronised inside the timed region) of (a) ONE hover->waypoint solve through the planner mirror
    (`plan_trajectory`, what the contract test times) and (b) the batched solve of `B` problems in one
    launch.  Every rank runs its own shard; rank 0 reports."""
    import torch.distributed as dist
    from dart_planner_amd.capi import Params
    from dart_planner_amd.common.types import DroneState
    from dart_planner_amd.planning.se3_mpc_planner import SE3MPCConfig, SE3MPCPlanner
    out = {}
    for prec in ("f64", "f32"):
        pl = SE3MPCPlanner(SE3MPCConfig(prediction_horizon=N), precision=prec, device=dev)
        st = DroneState(timestamp=0.0, position=np.array([0.0, 0.0, 1.0]), velocity=np.zeros(3))
        rng = np.random.default_rng(0)
        goals = rng.uniform(-5, 5, (240, 3)); goals[:, 2] = np.abs(goals[:, 2]) + 0.5
        ts = []
        for i, g in enumerate(goals):
            t0 = time.perf_counter()
            pl.plan_trajectory(st, g)
            torch.cuda.synchronize()
            if i >= 20:
                ts.append((time.perf_counter() - t0) * 1e3)
        out[f"single_{prec}"] = {"horizon": N, "calls": len(ts), "p50_ms": float(np.percentile(ts, 50)),
                                 "p95_ms": float(np.percentile(ts, 95)), "max_ms": float(np.max(ts))}
    prm = Params.reference_defaults(horizon=N)
    g = torch.Generator(device=dev); g.manual_seed(100 + rank)
    p0 = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    v0 = torch.rand(B, 3, device=dev, generator=g) * 10 - 5
    goal = torch.rand(B, 3, device=dev, generator=g) * 40 - 20
    ts = []
    nfev = None
    for i in range(220):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = ops.solve(prm, p0, v0, goal)
        torch.cuda.synchronize()
        if i >= 20:
            ts.append((time.perf_counter() - t0) * 1e3)
    info = ops.info_to_host(o["info"])
    mean_ms = float(np.mean(ts))
    agg = torch.tensor([B / (mean_ms * 1e-3)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(agg)
    out["batch_f32"] = {"horizon": N, "batch_per_gpu": B, "calls": len(ts), "p50_ms": float(np.percentile(ts, 50)),
                        "p95_ms": float(np.percentile(ts, 95)), "solves_per_s": float(agg.item()),
                        "mean_nfev": float(info["nfev"].mean()), "mean_nit": float(info["nit"].mean()),
                        "rollouts_inside_solves_per_s": float(agg.item() * info["nfev"].mean())}
    return out


def obstacle_source_leg(torch, ops, dev):
    """The caller side of the path at the sizes of the reference's planning cycle
    (cloud/main_improved_threelayer.py:204-209, 381-398): one 360-ray scan into the device voxel map, then the
    20 m local grid at 0.2 m (10^6 cells) -> 20 obstacle spheres, then the safety check of 8192 30-step plans.
    Device-event times of the kernels with resident inputs; seeded synthetic scene."""
    from dart_planner_amd.perception.explicit_geometric_mapper import ExplicitGeometricMapper
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

    def dev_ms(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    m.map.update_rays(org, dirs, dist, hit.astype(np.int32))          # first scan: creates the voxels, sizes the table
    upd = dev_ms(lambda: m.map.update_rays(org, dirs, dist, hit.astype(np.int32)), 5)
    centre = np.array([0.3, 0.1, 2.0])
    grid = dev_ms(lambda: m.map.local_spheres(centre, 20.0, 0.6, 20, 1.0), 20)
    plans = torch.rand(8192, 30, 3, device=dev, dtype=torch.float32) * 20 - 10
    safe = dev_ms(lambda: m.map.trajectories_safe(plans, margin=1.0, threshold=0.6), 20)
    return {"what": "device voxel map (reference: ExplicitGeometricMapper's dict walk), float64 index arithmetic, bit-exact",
            "update_map_360_rays_ms": upd, "local_grid_1e6_cells_to_spheres_ms": grid, "trajectory_safe_8192x30_ms": safe,
            "voxels": len(m.map), "table_capacity": m.map.capacity}


def profiled_traffic(B, N, S):
    """HBM bytes per launch of the timed kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_traffic.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes, same command
    without the graph) -- PMC counters cannot be read from inside this process.  None if no profile
    of this exact workload is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("batch") == B and d.get("horizon") == N and d.get("steps_per_launch", 1) == S:
            best = (float(d["traffic_bytes_per_launch"]), os.path.basename(f))
    return best


def sweep(torch, ops, prm, dev, N):
    """Saturating batches (one launch each, eager): where the HBM roofline is actually reachable."""
    out = []
    for B in (65536, 1 << 20, 1 << 22):
        p0, v0, goal, T, cost, grad = make_ring(torch, dev, B, N, 2, seed=11)
        for _ in range(3):
            ops.rollout_cost_grad(prm, p0[0], v0[0], goal[0], T[0], out=(cost[0], grad[0]))
        torch.cuda.synchronize()
        reps = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            ops.rollout_cost_grad(prm, p0[i & 1], v0[i & 1], goal[i & 1], T[i & 1], out=(cost[i & 1], grad[i & 1]))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbps = 4 * (6 * N + 10) * B / (ms * 1e-3) / 1e9
        out.append({"batch": B, "kernel_us": ms * 1e3, "rollouts_per_s": B / (ms * 1e-3), "GB_per_s": gbps,
                    "frac_of_peak": gbps / HBM_PEAK_GBPS})
        del p0, v0, goal, T, cost, grad
    return out


if __name__ == "__main__":
    main()
