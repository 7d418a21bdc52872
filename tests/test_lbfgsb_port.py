"""oracle/lbfgsb_port.py (the restatement of L-BFGS-B the HIP solver is modelled on) pinned
against the installed SciPy -- the library the reference itself calls
(src/dart_planner/planning/se3_mpc_planner.py:256-268) -- on the reference's objective/gradient
(golden solve cases, incl. the ABNORMAL line-search terminations its inconsistent gradient
causes) and on generic box-constrained problems that exercise memory wrap-around, restarts and
the projected subspace step.  CPU only."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import lbfgsb_port as lb
from oracle import se3mpc_oracle as orc


def test_port_reproduces_reference_solves(golden_solve):
    data, meta = golden_solve
    for c in meta["cases"]:
        k = c["key"]
        cfg = orc.OracleConfig(prediction_horizon=c["N"], dt=c["dt"], max_iterations=c["maxiter"],
                               convergence_tolerance=c["tol"])
        goal = data[k + "goal"]
        x0 = orc.straight_line_init(data[k + "p0"], data[k + "v0"], goal, cfg)
        b = orc.bounds(cfg)
        r = lb.minimize(lambda x: (float(orc.objective(x, goal, cfg)), orc.gradient(x, goal, cfg)), x0, b[:, 0], b[:, 1],
                        m=10, ftol=10 * c["tol"], pgtol=c["tol"], maxiter=c["maxiter"])
        assert (r.nit, r.nfev, r.status) == (c["nit"], c["nfev"], c["status"]), (k, r.task)
        assert np.allclose(r.x, data[k + "x"], rtol=0, atol=1e-10), k
        assert np.isclose(r.fun, float(data[k + "fun"]), rtol=1e-12), k


@pytest.mark.parametrize("seed", range(12))
def test_port_matches_scipy_on_generic_box_problems(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 50))
    A = rng.normal(size=(n, n)); Q = A @ A.T / n + 0.05 * np.eye(n); b = rng.normal(size=n) * 3
    kind = seed % 3

    def fg(x):
        if kind == 0:
            return 0.5 * x @ Q @ x - b @ x, Q @ x - b
        if kind == 1:
            f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
            g = np.zeros(n)
            g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1]); g[1:] += 200 * (x[1:] - x[:-1] ** 2)
            return f, g
        return np.sum(np.cos(x) * x) + 0.5 * x @ Q @ x, -np.sin(x) * x + np.cos(x) + Q @ x

    lo, hi = -rng.uniform(0.2, 2, n), rng.uniform(0.2, 2, n)
    x0 = rng.uniform(-3, 3, n)
    for m in (10, 3):
        mi = int(rng.integers(5, 80))
        tr = []
        r = lb.minimize(fg, x0, lo, hi, m=m, ftol=1e-12, pgtol=1e-9, maxiter=mi, trace=True)
        res = minimize(lambda x: fg(x)[0], x0, jac=lambda x: fg(x)[1], method="L-BFGS-B", bounds=list(zip(lo, hi)),
                       callback=lambda xk: tr.append(np.array(xk)),
                       options=dict(maxiter=mi, gtol=1e-9, ftol=1e-12, maxcor=m))
        assert (r.nit, r.nfev, r.status) == (res.nit, res.nfev, res.status), (seed, m, r.task, res.message)
        assert np.allclose(r.x, res.x, rtol=0, atol=1e-9)
        assert len(tr) == len(r.trace) and all(np.allclose(a, b_, atol=1e-6) for a, b_ in zip(r.trace, tr))  # rounding-order noise is amplified mid-run on Rosenbrock
