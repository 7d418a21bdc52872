"""world_size-2 `gloo` test of the multi-GPU path on CPU: restarts sharded over two ranks, packed-key
all-reduce(MIN), winner broadcast -- must equal the single-process answer over all restarts.
(Kernels: the product sources compiled for the host, tests/emu.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def _ops():
    import build_emu
    from numpy_backend import TorchCpuBackend
    from dart_planner_amd import capi
    from dart_planner_amd.ops import Ops
    return Ops(TorchCpuBackend(), capi.Library(build_emu.build()))


PROBLEM = dict(p0=[1.0, -2.0, 1.5], v0=[0.2, 0.0, -0.1], goal=[4.0, 1.0, 3.0])
R_TOTAL = 5


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dart_planner_amd.capi import Params
    from dart_planner_amd import distributed as D
    D.init_distributed("gloo")
    res = D.sharded_restart_solve(_ops(), Params.reference_defaults(horizon=6), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"],
                                  n_restarts=R_TOTAL, sigma=3.0, seed=4, precision="f64")
    # bucketed key exchange as bench.py does it: K keys, one collective
    raw = [((0x80000000 | (100 + 7 * rank)) << 32) | (10 * rank + 1), ((0x80000000 | (50 - rank)) << 32) | (10 * rank + 2)]
    keys = torch.from_numpy(np.array(raw, dtype=np.uint64).view(np.int64).copy())
    D.allreduce_min_keys(keys)
    # population mean (MPPI weights on the GLOBAL minimum: the all-reduced key) over a batch sharded across the ranks
    ops = _ops()
    Xall, call = _population()
    lo, hi = D.shard_bounds(Xall.shape[1], rank, world)
    Xs, cs = torch.from_numpy(np.ascontiguousarray(Xall[:, lo:hi])), torch.from_numpy(np.ascontiguousarray(call[lo:hi]))
    kmin = torch.zeros(1, dtype=torch.int64)
    ops.argmin(cs, index_base=lo, out=kmin)
    D.allreduce_min_keys(kmin)
    mean = D.allreduce_population_mean(ops.population_sums(Xs, cost=cs, temperature=POP_TEMPERATURE, ref_key=kmin))
    shoot = D.sharded_shooting_plan(_ops(), Params.reference_defaults(horizon=6), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"], n_samples=SHOOT_SAMPLES,
                                    iters=3, step=0.9, sigma=4.0, seed=8, precision="f64")
    q.put((rank, res["restart"], res["owner"], res["cost"], res["x"], keys.numpy().view(np.uint64).tolist(), mean.numpy(), shoot))
    torch.distributed.destroy_process_group()


POP_TEMPERATURE = 3.0
SHOOT_SAMPLES = 131


def _population():
    rng = np.random.default_rng(9)
    return rng.normal(size=(18, 301)).astype(np.float32), rng.uniform(100.0, 130.0, 301).astype(np.float32)


def test_shard_bounds():
    from dart_planner_amd.distributed import shard_bounds
    for total in (0, 1, 5, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_restart_argmin_matches_single_process():
    from dart_planner_amd.capi import Params
    from dart_planner_amd import distributed as D
    single = D.sharded_restart_solve(_ops(), Params.reference_defaults(horizon=6), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"],
                                     n_restarts=R_TOTAL, sigma=3.0, seed=4, precision="f64")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    Xall, call = _population()
    w = np.exp(-(call.astype(np.float64) - float(call.min())) / POP_TEMPERATURE)
    mean_ref = (Xall.astype(np.float64) * w).sum(1) / w.sum()
    shoot1 = D.sharded_shooting_plan(_ops(), Params.reference_defaults(horizon=6), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"], n_samples=SHOOT_SAMPLES,
                                     iters=3, step=0.9, sigma=4.0, seed=8, precision="f64")
    for rank, restart, owner, cost, x, keys, mean, shoot in got:
        # the sharded shooting plan (samples over two ranks, one key all-reduce, winner broadcast) == all samples in one process
        assert shoot["sample"] == shoot1["sample"] and shoot["cost"] == shoot1["cost"] and np.array_equal(shoot["T"], shoot1["T"])
        assert shoot["owner"] == (0 if shoot["sample"] < 66 else 1)
        assert np.allclose(mean, mean_ref, rtol=1e-12, atol=1e-14)
        assert restart == single["restart"] and cost == single["cost"]
        assert np.array_equal(x, single["x"])
        assert owner == (0 if restart < 3 else 1)                      # shards: rank 0 -> [0,3), rank 1 -> [3,5)
        # MIN of the unsigned keys: step 0 -> rank 0's (cost bits 100), step 1 -> rank 1's (cost bits 49)
        assert keys == [((0x80000000 | 100) << 32) | 1, ((0x80000000 | 49) << 32) | 12]


SHOOT3 = dict(n_samples=700, iters=3, step=2e-3, sigma=3.0, seed=5, precision="f64")        # 700 = 234 + 233 + 233: the remainder path; 3 sample blocks
SPHERES3 = [[1.5, -0.5, 2.0, 0.8], [3.0, 0.5, 2.5, 0.6]]


def _worker3(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dart_planner_amd.capi import Params
    from dart_planner_amd import distributed as D
    D.init_distributed("gloo")
    drawn = []
    real_randn = torch.randn

    def counting_randn(*a, **k):
        drawn.append(int(np.prod(a[:2])))
        return real_randn(*a, **k)

    torch.randn = counting_randn
    try:
        shoot = D.sharded_shooting_plan(_ops(), Params.reference_defaults(horizon=6, dt=0.1), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"],
                                        spheres=SPHERES3, obstacle_weight=30.0, **SHOOT3)
    finally:
        torch.randn = real_randn
    q.put((rank, shoot, sum(drawn)))
    torch.distributed.destroy_process_group()


def test_three_rank_shooting_plan_with_obstacles_and_a_remainder():
    """world_size 3, a sample count the world does not divide (shard_bounds' remainder path through sharded_shooting_plan), the
    obstacle-aware loop: every rank returns the single-process plan, and draws only the sample blocks its shard touches."""
    from dart_planner_amd.capi import Params
    from dart_planner_amd import distributed as D
    single = D.sharded_shooting_plan(_ops(), Params.reference_defaults(horizon=6, dt=0.1), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"],
                                     spheres=SPHERES3, obstacle_weight=30.0, **SHOOT3)
    blind = D.sharded_shooting_plan(_ops(), Params.reference_defaults(horizon=6, dt=0.1), PROBLEM["p0"], PROBLEM["v0"], PROBLEM["goal"], **SHOOT3)
    assert not np.array_equal(single["T"], blind["T"])                 # the spheres do change the plan
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in procs), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bounds = [D.shard_bounds(SHOOT3["n_samples"], r, 3) for r in range(3)]
    assert [hi - lo for lo, hi in bounds] == [234, 233, 233]
    for rank, shoot, drawn in got:
        assert shoot["sample"] == single["sample"] and shoot["cost"] == single["cost"] and np.array_equal(shoot["T"], single["T"])
        assert shoot["owner"] == next(r for r, (lo, hi) in enumerate(bounds) if lo <= single["sample"] < hi)
        lo, hi = bounds[rank]
        blocks = (hi + D.SAMPLE_BLOCK - 1) // D.SAMPLE_BLOCK - lo // D.SAMPLE_BLOCK
        assert drawn == blocks * D.SAMPLE_BLOCK * 18 and drawn < 18 * (SHOOT3["n_samples"] + D.SAMPLE_BLOCK)    # O(shard), not O(n_samples)
    assert got[0][2] + got[1][2] + got[2][2] <= 18 * (SHOOT3["n_samples"] + 3 * D.SAMPLE_BLOCK)


def test_shooting_samples_do_not_depend_on_the_shard():
    from dart_planner_amd.capi import Params
    from dart_planner_amd import distributed as D
    prm = Params.reference_defaults(horizon=6)
    full = D.shooting_samples(prm, 700, 2.0, 11, torch.device("cpu"), torch.float64)
    assert full.shape == (18, 700) and torch.all(full[:, 0] == torch.tensor([0.0, 0.0, prm.mass * prm.gravity] * 6, dtype=torch.float64))
    for lo, hi in ((0, 700), (0, 1), (255, 257), (256, 512), (300, 301), (511, 700), (700, 700)):
        assert torch.equal(D.shooting_samples(prm, 700, 2.0, 11, torch.device("cpu"), torch.float64, lo, hi), full[:, lo:hi])


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` without a torchrun environment must start 2 ranks as a child process and report
    them (dry run: rank plumbing only, gloo, no HIP) -- a driver that calls bench.py directly gets a real N-rank run
    or a non-zero exit, never a 1-rank line."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "20", "--warmup", "5"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_joined"] == 2 and d["steps"] == 20 and d["keys_valid"] is True
    # --gpus N under a torchrun environment of another size is refused, not silently relabelled
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env2,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_bench_timed_plan_shape_does_not_depend_on_steps():
    """The timed region is built from FULL launches repeated to a minimum duration, whatever --steps says."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for K in (1, 20, 64, 65, 20000):
        lpp = -(-K // 64)
        p = bench.timed_plan(K, 64, pass_ms=0.065 * lpp, min_ms=20.0)
        assert p["launches_per_pass"] == lpp and p["steps_per_pass"] == 64 * lpp
        assert p["repeats"] * 0.065 * lpp >= 20.0 and p["passes_per_graph"] * lpp <= bench.MAX_GRAPH_NODES
    p = bench.timed_plan(20, 1, pass_ms=20 * 0.0045, min_ms=20.0)
    assert p["launches_per_pass"] == 20 and p["repeats"] * 20 * 0.0045 >= 20.0
