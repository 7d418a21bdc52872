"""Developer probe (GPU box): parity of the lane-layout kernels vs the oracle and timing of the
three rollout variants over batch sizes.  Writes gpurun_out/probe_eval.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dart_planner_amd.capi as capi
if not hasattr(__import__("ctypes").CDLL(capi.DEFAULT_LIBRARY), "se3mpc_solve_f32"):
    capi._TYPED_API.pop("solve")
from dart_planner_amd.ops import Ops
from oracle import se3mpc_oracle as orc

ops = Ops()
lib = ops.lib
dev = ops.be.device
out = {"device": torch.cuda.get_device_name(0), "devices": lib.device_count()}
rng = np.random.default_rng(0)

def lane(a, B, dt):
    return torch.from_numpy(np.ascontiguousarray(a.reshape(B, -1).T)).to(dev, dt).contiguous()

par = {}
for dt, tname in ((torch.float64, "f64"), (torch.float32, "f32")):
    for N in (1, 6, 30, 50, 13):
        B = 1000
        cfg = orc.OracleConfig(prediction_horizon=N)
        prm = capi.Params.reference_defaults(horizon=N)
        p0, v0, goal = rng.uniform(-20, 20, (B, 3)), rng.uniform(-5, 5, (B, 3)), rng.uniform(-20, 20, (B, 3))
        T = rng.normal(0, 2, (B, N, 3)) + [0, 0, cfg.hover_thrust]
        c_ref, g_ref = orc.rollout_cost_grad(p0, v0, goal, T, cfg)
        for var in (1, 2, 3, 4, 5):
            lib.set_rollout_variant(var)
            cost, gT, P, V = ops.rollout_cost_grad(prm, lane(p0, B, dt), lane(v0, B, dt), lane(goal, B, dt), lane(T, B, dt), want_states=True)
            torch.cuda.synchronize()
            e1 = float(np.max(np.abs(cost.cpu().numpy() - c_ref) / np.abs(c_ref)))
            e2 = float(np.max(np.abs(gT.cpu().numpy().T.reshape(B, N, 3) - g_ref)) / np.max(np.abs(g_ref)))
            par[f"{tname}_N{N}_var{var}"] = (e1, e2)
        lib.set_rollout_variant(0)
        X = np.concatenate([rng.uniform(-120, 120, (B, 3 * N)), rng.uniform(-15, 15, (B, 3 * N)), T.reshape(B, -1)], axis=1)
        f, g = ops.cost_grad(prm, lane(X, B, dt), lane(goal, B, dt))
        fr = orc.objective(X, goal, cfg)
        par[f"{tname}_N{N}_costgrad"] = (float(np.max(np.abs(f.cpu().numpy() - fr) / fr)), float(np.max(np.abs(g.cpu().numpy().T - orc.gradient(X, goal, cfg)))))
        acc, att, rates, thr = ops.extract(prm, lane(T, B, dt))
        ex = orc.extract_solution_batch(X, cfg)
        par[f"{tname}_N{N}_extract"] = (float(np.max(np.abs(att.cpu().numpy().T.reshape(B, N, 3) - ex["attitudes"]))), float(np.max(np.abs(rates.cpu().numpy().T.reshape(B, N, 3) - ex["body_rates"]))))
        key = ops.argmin(f)
        par[f"{tname}_N{N}_argmin"] = (ops.decode_key(key)[0], int(np.argmin(f.cpu().numpy())))
out["parity"] = par
print(json.dumps(par, indent=0)[:3000])

# timing
tim = {}
N = 30
prm = capi.Params.reference_defaults(horizon=N)
for B in (8192, 65536, 1 << 20, 1 << 22):
    g = torch.Generator(device=dev); g.manual_seed(1)
    p0 = (torch.rand(3, B, device=dev, generator=g) * 40 - 20)
    v0 = (torch.rand(3, B, device=dev, generator=g) * 10 - 5)
    goal = (torch.rand(3, B, device=dev, generator=g) * 40 - 20)
    T = torch.randn(3 * N, B, device=dev, generator=g) * 2
    T[2::3] += 14.715
    cost = torch.empty(B, device=dev); gT = torch.empty(3 * N, B, device=dev)
    for var in (1, 3, 4, 5, 2):
        lib.set_rollout_variant(var)
        for _ in range(5):
            ops.rollout_cost_grad(prm, p0, v0, goal, T, out=(cost, gT))
        torch.cuda.synchronize()
        K = 200 if B <= 65536 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            ops.rollout_cost_grad(prm, p0, v0, goal, T, out=(cost, gT))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        bytes_ = 4 * (6 * N + 10) * B
        tim[f"B{B}_var{var}"] = dict(us=ms * 1e3, rollouts_per_s=B / (ms * 1e-3), GBps=bytes_ / (ms * 1e-3) / 1e9)
        print(B, var, tim[f"B{B}_var{var}"], flush=True)
    lib.set_rollout_variant(0)
    # argmin alone
    key = torch.empty(1, dtype=torch.int64, device=dev)
    for _ in range(5): ops.argmin(cost, out=key)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ops.argmin(cost, out=key)
    e1.record(); torch.cuda.synchronize()
    tim[f"B{B}_argmin_us"] = e0.elapsed_time(e1) / 100 * 1e3
    print(B, "argmin us", tim[f"B{B}_argmin_us"], flush=True)
    # parity-form cost_grad streaming
    X = torch.randn(9 * N, B, device=dev)
    f = None
    for _ in range(3): ops.cost_grad(prm, X, goal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.cost_grad(prm, X, goal)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tim[f"B{B}_costgrad"] = dict(us=ms * 1e3, GBps=4 * (18 * N + 1) * B / (ms * 1e-3) / 1e9)
    print(B, "costgrad", tim[f"B{B}_costgrad"], flush=True)
out["timing"] = tim
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_eval.json"), "w"), indent=1)
