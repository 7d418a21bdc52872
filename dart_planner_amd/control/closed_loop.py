"""Batched closed loop on the device: planner -> (plan sample -> geometric controller -> simulator) for many drones at once.

What the reference's closed-loop tests do one drone and one Python call at a time
(tests/test_planner_controller_contract.py:115-162, :255-316; tests/test_monte_carlo_sim.py:24-72) runs here as two launches
per planning cycle for B drones: ``se3mpc_solve_*`` (every drone re-plans from its own state) and ``se3mpc_closed_loop_*``
(``substeps`` control + simulator steps against the fresh plan, which is read in place from the solver's outputs).  No host
arithmetic, no copies between the two; the only host work per cycle is the N plan stamps (planner.py:661: start + arange(N)*dt).
"""
from typing import Optional

from ..capi import ControllerParams, Params, SimulatorParams


class ClosedLoopMonteCarlo:
    """Receding-horizon Monte-Carlo over B drones (BASELINE.json config 5's named test shape).

    ops: dart_planner_amd.ops.Ops;  params: se3mpc_params of the planner (horizon N, dt);  controller / simulator: the
    C-ABI parameter structs (defaults: the reference's "sitl_optimized" controller and DroneSimulator())."""

    def __init__(self, ops, params: Params, controller: Optional[ControllerParams] = None, simulator: Optional[SimulatorParams] = None):
        self.ops, self.params = ops, params
        self.controller = controller if controller is not None else ops.lib.controller_default_params()
        self.simulator = simulator if simulator is not None else ops.lib.simulator_default_params()

    def run(self, p0, v0, goal, cycles: int, substeps: int, sim_dt: float, wind=None, log: bool = False):
        """p0, v0, goal: (B, 3) device tensors (float32 or float64: the precision of the whole loop); wind: None, (3,) or (B, 3) newtons.
        -> dict(pos, vel, att, omega (B, 3), time (B,), controller_state (B, 12), logs = [(solve outputs, closed-loop outputs)] if log)."""
        import torch
        ops, prm = self.ops, self.params
        dev = ops.be.device
        B, N = p0.shape[0], prm.horizon
        pos, vel = p0.clone(), v0.clone()
        att, om = torch.zeros_like(p0), torch.zeros_like(p0)
        time = torch.zeros(B, dtype=torch.float64, device=dev)
        st = ops.controller_state(self.controller, B)
        k = torch.arange(N, dtype=torch.float64, device=dev)
        logs = []
        sol = None
        for c in range(cycles):
            # without logs every cycle writes the same plan tensors again (the closed-loop launch of a cycle is ordered before the next solve)
            sol = ops.solve(prm, pos, vel, goal, want_trajectory=True if log else "accelerations", out=None if log else sol)
            stamps = (c * substeps * sim_dt) + k * prm.dt
            X = sol["x"]
            out = ops.closed_loop(self.controller, self.simulator, st, time, pos, vel, att, om, stamps, X, X[:, 3 * N:], sol["accelerations"],
                                  nsteps=substeps, sim_dt=sim_dt, strides=(9 * N, 9 * N, 3 * N), wind=wind, stop_at_plan_end=False, log=log)
            if log:
                logs.append((sol, out))
        return dict(pos=pos, vel=vel, att=att, omega=om, time=time, controller_state=st, logs=logs)

    def run_fused(self, p0, v0, goal, cycles: int, substeps: int, sim_dt: float, wind=None, want_last_plan: bool = False):
        """The same Monte-Carlo in ONE launch (``se3mpc_monte_carlo_*``: every cycle's solve and control / simulator steps inside one kernel,
        each drone paying only for its own slow solves instead of waiting, 2 x `cycles` times, at a kernel boundary for the slowest drone of
        the batch).  Same code, same bits as :meth:`run`.  One host synchronise at the end reads the overflow counter; if a solve needed more
        L-BFGS memory than the launch's LDS image holds (never with the reference's options) the run is repeated by :meth:`run`."""
        import torch
        ops = self.ops
        dev = ops.be.device
        B = p0.shape[0]
        pos, vel = p0.clone(), v0.clone()
        att, om = torch.zeros_like(p0), torch.zeros_like(p0)
        time = torch.zeros(B, dtype=torch.float64, device=dev)
        st = ops.controller_state(self.controller, B)
        out = ops.monte_carlo(self.params, self.controller, self.simulator, st, time, pos, vel, att, om, goal, cycles, substeps, sim_dt, wind=wind,
                              want_last_plan=want_last_plan)
        if int(ops.be.to_host(out["overflowed"])[0]) != 0:
            return self.run(p0, v0, goal, cycles, substeps, sim_dt, wind=wind)
        return dict(pos=pos, vel=vel, att=att, omega=om, time=time, controller_state=st, logs=[], last_plan=out if want_last_plan else None)

    def capture(self, B: int, dtype, cycles: int, substeps: int, sim_dt: float, with_wind: bool = True):
        """The whole Monte-Carlo (2 x `cycles` kernel launches + the plan stamps) captured ONCE into a hipGraph; each call of the
        returned function copies new initial conditions into the graph's static inputs, replays it and returns the static outputs
        (overwritten by the next call).  Removes the per-launch host cost (~30 us of Python + launch per call, 66 calls per run)."""
        import torch
        dev = self.ops.be.device
        static = dict(p0=torch.zeros(B, 3, dtype=dtype, device=dev), v0=torch.zeros(B, 3, dtype=dtype, device=dev),
                      goal=torch.zeros(B, 3, dtype=dtype, device=dev), wind=torch.zeros(B, 3, dtype=dtype, device=dev) if with_wind else None)
        run = lambda: self.run(static["p0"], static["v0"], static["goal"], cycles, substeps, sim_dt, wind=static["wind"])
        run()                                                    # warm-up outside the capture (library load, allocator pools)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                out = run()
        torch.cuda.current_stream().wait_stream(side)

        def replay(p0, v0, goal, wind=None):
            static["p0"].copy_(p0); static["v0"].copy_(v0); static["goal"].copy_(goal)
            if with_wind:
                static["wind"].copy_(wind) if wind is not None else static["wind"].zero_()
            graph.replay()
            return out
        replay.graph = graph
        return replay
