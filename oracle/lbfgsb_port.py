"""Restatement of L-BFGS-B 3.0 (Byrd, Lu, Nocedal, Zhu; Morales-Nocedal 2011 subspace step) --
TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's solve is one call of ``scipy.optimize.minimize(method="L-BFGS-B")``
(src/dart_planner/planning/se3_mpc_planner.py:256-268).  SciPy's implementation (pinned
scipy==1.16.0 by the reference, 1.15.3 in this image: both the C translation of the Fortran
L-BFGS-B 3.0) is a third-party dependency that is not under /root/reference, so this module
restates the published algorithm -- routine by routine, with the routine names of lbfgsb.f --
and ``tests/test_lbfgsb_port.py`` pins it against the installed SciPy: identical iterates
(callback trace) and identical nit / nfev / status on the reference's objective, including the
runs that end in ABNORMAL_TERMINATION_IN_LNSRCH because the reference's gradient is not the
gradient of its objective.  The HIP solver (dart_planner_amd/csrc/solve_kernel.hip) follows
this restatement step for step; it is NOT derived from it mechanically and shares no code.

All variables here are bounded on both sides (nbd = 2), as in the reference's box
(planner.py:378-402); the general nbd cases of the Fortran are kept where they are cheap.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import numpy as np

EPSMCH = np.finfo(float).eps
BIG = 1.0e10
FTOL, GTOL, XTOL = 1.0e-3, 0.9, 0.1      # line-search constants of lnsrlb


@dataclass
class Result:
    x: np.ndarray
    fun: float
    nit: int
    nfev: int
    status: int          # scipy: 0 converged, 1 limit reached, 2 abnormal
    task: str
    trace: List[np.ndarray] = field(default_factory=list)


# ------------------------------------------------------------------------------- dcstep
def dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    """MINPACK-2 dcstep: safeguarded cubic/quadratic step and interval update."""
    sgnd = dp * (dx / abs(dx))
    if fp > fx:
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp < stx:
            gamma = -gamma
        p = (gamma - dx) + theta
        q = ((gamma - dx) + gamma) + dp
        r = p / q
        stpc = stx + r * (stp - stx)
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx)
        if abs(stpc - stx) < abs(stpq - stx):
            stpf = stpc
        else:
            stpf = stpc + (stpq - stpc) / 2.0
        brackt = True
    elif sgnd < 0.0:
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = ((gamma - dp) + gamma) + dx
        r = p / q
        stpc = stp + r * (stx - stp)
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if abs(stpc - stp) > abs(stpq - stp):
            stpf = stpc
        else:
            stpf = stpq
        brackt = True
    elif abs(dp) < abs(dx):
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt(max(0.0, (theta / s) ** 2 - (dx / s) * (dp / s)))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = (gamma + (dx - dp)) + gamma
        r = p / q
        if r < 0.0 and gamma != 0.0:
            stpc = stp + r * (stx - stp)
        elif stp > stx:
            stpc = stpmax
        else:
            stpc = stpmin
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if brackt:
            if abs(stpc - stp) < abs(stpq - stp):
                stpf = stpc
            else:
                stpf = stpq
            if stp > stx:
                stpf = min(stp + 0.66 * (sty - stp), stpf)
            else:
                stpf = max(stp + 0.66 * (sty - stp), stpf)
        else:
            if abs(stpc - stp) > abs(stpq - stp):
                stpf = stpc
            else:
                stpf = stpq
            stpf = min(stpmax, stpf)
            stpf = max(stpmin, stpf)
    else:
        if brackt:
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
            s = max(abs(theta), abs(dy), abs(dp))
            gamma = s * math.sqrt((theta / s) ** 2 - (dy / s) * (dp / s))
            if stp > sty:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dy
            r = p / q
            stpc = stp + r * (sty - stp)
            stpf = stpc
        elif stp > stx:
            stpf = stpmax
        else:
            stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0.0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    stp = stpf
    return stx, fx, dx, sty, fy, dy, stp, brackt


# ------------------------------------------------------------------------------- dcsrch
class DcsrchState:
    """Persistent state of one Moré-Thuente search (the isave/dsave of dcsrch)."""
    __slots__ = ("brackt", "stage", "ginit", "gtest", "gx", "gy", "finit", "fx", "fy", "stx", "sty", "stmin",
                 "stmax", "width", "width1")


def dcsrch(f, g, stp, ftol, gtol, xtol, stpmin, stpmax, task, st: DcsrchState):
    """Returns (stp, task).  task in {'START','FG','CONVERGENCE','WARNING...','ERROR...'}."""
    xtrapl, xtrapu, p5, p66 = 1.1, 4.0, 0.5, 0.66
    if task == "START":
        if stp < stpmin:
            return stp, "ERROR: STP .LT. STPMIN"
        if stp > stpmax:
            return stp, "ERROR: STP .GT. STPMAX"
        if g >= 0.0:
            return stp, "ERROR: INITIAL G .GE. ZERO"
        if stpmax < stpmin:
            return stp, "ERROR: STPMAX .LT. STPMIN"
        st.brackt = False
        st.stage = 1
        st.finit = f
        st.ginit = g
        st.gtest = ftol * st.ginit
        st.width = stpmax - stpmin
        st.width1 = st.width / p5
        st.stx, st.fx, st.gx = 0.0, st.finit, st.ginit
        st.sty, st.fy, st.gy = 0.0, st.finit, st.ginit
        st.stmin = 0.0
        st.stmax = stp + xtrapu * stp
        return stp, "FG"
    ftest = st.finit + stp * st.gtest
    if st.stage == 1 and f <= ftest and g >= 0.0:
        st.stage = 2
    task = "FG"
    if st.brackt and (stp <= st.stmin or stp >= st.stmax):
        task = "WARNING: ROUNDING ERRORS PREVENT PROGRESS"
    if st.brackt and st.stmax - st.stmin <= xtol * st.stmax:
        task = "WARNING: XTOL TEST SATISFIED"
    if stp == stpmax and f <= ftest and g <= st.gtest:
        task = "WARNING: STP = STPMAX"
    if stp == stpmin and (f > ftest or g >= st.gtest):
        task = "WARNING: STP = STPMIN"
    if f <= ftest and abs(g) <= gtol * (-st.ginit):
        task = "CONVERGENCE"
    if task.startswith("WARN") or task.startswith("CONV"):
        return stp, task
    if st.stage == 1 and f <= st.fx and f > ftest:
        fm = f - stp * st.gtest
        fxm = st.fx - st.stx * st.gtest
        fym = st.fy - st.sty * st.gtest
        gm = g - st.gtest
        gxm = st.gx - st.gtest
        gym = st.gy - st.gtest
        st.stx, fxm, gxm, st.sty, fym, gym, stp, st.brackt = dcstep(st.stx, fxm, gxm, st.sty, fym, gym, stp, fm, gm,
                                                                    st.brackt, st.stmin, st.stmax)
        st.fx = fxm + st.stx * st.gtest
        st.fy = fym + st.sty * st.gtest
        st.gx = gxm + st.gtest
        st.gy = gym + st.gtest
    else:
        st.stx, st.fx, st.gx, st.sty, st.fy, st.gy, stp, st.brackt = dcstep(st.stx, st.fx, st.gx, st.sty, st.fy, st.gy,
                                                                            stp, f, g, st.brackt, st.stmin, st.stmax)
    if st.brackt:
        if abs(st.sty - st.stx) >= p66 * st.width1:
            stp = st.stx + p5 * (st.sty - st.stx)
        st.width1 = st.width
        st.width = abs(st.sty - st.stx)
    if st.brackt:
        st.stmin = min(st.stx, st.sty)
        st.stmax = max(st.stx, st.sty)
    else:
        st.stmin = stp + xtrapl * (stp - st.stx)
        st.stmax = stp + xtrapu * (stp - st.stx)
    stp = max(stp, stpmin)
    stp = min(stp, stpmax)
    if (st.brackt and (stp <= st.stmin or stp >= st.stmax)) or (st.brackt and st.stmax - st.stmin <= xtol * st.stmax):
        stp = st.stx
    return stp, "FG"


# ------------------------------------------------------------------------------- small dense helpers
def dpofa(a: np.ndarray, n: int) -> int:
    """LINPACK dpofa on the leading n x n block: upper factor R (R'R = A) in the upper triangle.
    Returns info (0 ok, k = leading minor k not positive definite)."""
    for j in range(n):
        s = 0.0
        for k in range(j):
            t = a[k, j] - float(np.dot(a[:k, k], a[:k, j]))
            t = t / a[k, k]
            a[k, j] = t
            s += t * t
        s = a[j, j] - s
        if s <= 0.0:
            return j + 1
        a[j, j] = math.sqrt(s)
    return 0


def dtrsl_upper(t: np.ndarray, n: int, b: np.ndarray, transposed: bool) -> int:
    """LINPACK dtrsl with an UPPER triangular t: job 01 (t x = b) or job 11 (t' x = b), in place."""
    for j in range(n):
        if t[j, j] == 0.0:
            return j + 1
    if not transposed:          # job 01: back substitution
        b[n - 1] = b[n - 1] / t[n - 1, n - 1]
        for jj in range(1, n):
            j = n - 1 - jj
            b[:j + 1] += -b[j + 1] * t[:j + 1, j + 1]
            b[j] = b[j] / t[j, j]
    else:                       # job 11: forward substitution with t'
        b[0] = b[0] / t[0, 0]
        for j in range(1, n):
            b[j] = b[j] - float(np.dot(t[:j, j], b[:j]))
            b[j] = b[j] / t[j, j]
    return 0


class Lbfgsb:
    """mainlb of lbfgsb.f for an all-boxed problem (every nbd == 2)."""

    def __init__(self, n: int, m: int, lo: np.ndarray, hi: np.ndarray, factr_eps: float, pgtol: float,
                 maxls: int = 20):
        self.n, self.m = n, m
        self.l, self.u = np.asarray(lo, float), np.asarray(hi, float)
        self.tol = factr_eps            # factr * epsmch  (== scipy's ftol)
        self.pgtol = pgtol
        self.maxls = maxls
        self.ws = np.zeros((n, m)); self.wy = np.zeros((n, m))          # columns oldest -> newest
        self.sy = np.zeros((m, m)); self.ss = np.zeros((m, m)); self.wt = np.zeros((m, m))
        self.wn = np.zeros((2 * m, 2 * m))
        self.col = 0; self.theta = 1.0; self.iupdat = 0; self.updatd = False
        self.iwhere = np.zeros(n, dtype=int)

    # ---- projgr
    def projgr(self, x, g) -> float:
        gi = np.where(g < 0.0, np.maximum(x - self.u, g), np.minimum(x - self.l, g))
        return float(np.max(np.abs(gi))) if self.n else 0.0

    # ---- bmv: product of the 2col x 2col middle matrix with v
    def bmv(self, v: np.ndarray) -> Tuple[np.ndarray, int]:
        col, sy, wt = self.col, self.sy, self.wt
        p = np.zeros(2 * col)
        if col == 0:
            return p, 0
        p[col] = v[col]
        for i in range(1, col):
            s = 0.0
            for k in range(i):
                s += sy[i, k] * v[k] / sy[k, k]
            p[col + i] = v[col + i] + s
        b = p[col:2 * col].copy()
        info = dtrsl_upper(wt, col, b, transposed=True)
        if info:
            return p, info
        p[col:2 * col] = b
        for i in range(col):
            p[i] = v[i] / math.sqrt(sy[i, i])
        b = p[col:2 * col].copy()
        info = dtrsl_upper(wt, col, b, transposed=False)
        if info:
            return p, info
        p[col:2 * col] = b
        for i in range(col):
            p[i] = -p[i] / math.sqrt(sy[i, i])
        for i in range(col):
            s = 0.0
            for k in range(i + 1, col):
                s += sy[k, i] * p[col + k] / sy[i, i]
            p[i] += s
        return p, 0

    # ---- cauchy: generalized Cauchy point
    def cauchy(self, x, g, sbgnrm):
        n, col, theta = self.n, self.col, self.theta
        l, u = self.l, self.u
        xcp = x.copy()
        c = np.zeros(2 * col)
        self.nseg = 0
        if sbgnrm <= 0.0:
            return xcp, c, 0
        d = np.zeros(n)
        p = np.zeros(2 * col)
        f1 = 0.0
        tbp = np.full(n, np.inf)          # breakpoint of each variable (inf = none)
        bnded = True
        nbreak = 0
        nfree_cnt = 0
        for i in range(n):
            neggi = -g[i]
            if self.iwhere[i] != 3 and self.iwhere[i] != -1:
                tl = x[i] - l[i]
                tu = u[i] - x[i]
                xlower = tl <= 0.0
                xupper = tu <= 0.0
                self.iwhere[i] = 0
                if xlower:
                    if neggi <= 0.0:
                        self.iwhere[i] = 1
                elif xupper:
                    if neggi >= 0.0:
                        self.iwhere[i] = 2
                else:
                    if abs(neggi) <= 0.0:
                        self.iwhere[i] = -3
            if self.iwhere[i] != 0 and self.iwhere[i] != -1:
                d[i] = 0.0
            else:
                d[i] = neggi
                f1 -= neggi * neggi
                for j in range(col):
                    p[j] += self.wy[i, j] * neggi
                    p[col + j] += self.ws[i, j] * neggi
                if neggi < 0.0:
                    nbreak += 1
                    tbp[i] = tl / (-neggi)
                elif neggi > 0.0:
                    nbreak += 1
                    tbp[i] = tu / neggi
                else:
                    nfree_cnt += 1
        if theta != 1.0:
            p[col:] *= theta
        if nbreak == 0 and nfree_cnt == 0:
            return xcp, c, 0
        f2 = -theta * f1
        f2_org = f2
        if col > 0:
            v, info = self.bmv(p)
            if info:
                return xcp, c, info
            f2 -= float(np.dot(v, p))
        dtm = -f1 / f2
        tsum = 0.0
        self.nseg = 1
        nleft = nbreak
        tj = 0.0
        all_fixed = False
        while nleft > 0:
            tj0 = tj
            ibp = int(np.argmin(tbp))          # next smallest breakpoint (hpsolb)
            tj = float(tbp[ibp])
            dt = tj - tj0
            if dtm < dt:
                break
            tsum += dt
            nleft -= 1
            tbp[ibp] = np.inf
            dibp = d[ibp]
            d[ibp] = 0.0
            if dibp > 0.0:
                zibp = u[ibp] - x[ibp]; xcp[ibp] = u[ibp]; self.iwhere[ibp] = 2
            else:
                zibp = l[ibp] - x[ibp]; xcp[ibp] = l[ibp]; self.iwhere[ibp] = 1
            if nleft == 0 and nbreak == n:
                dtm = dt
                all_fixed = True
                break
            self.nseg += 1
            dibp2 = dibp * dibp
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp
            f2 = f2 - theta * dibp2
            if col > 0:
                c += dt * p
                wbp = np.concatenate([self.wy[ibp, :col], theta * self.ws[ibp, :col]])
                v, info = self.bmv(wbp)
                if info:
                    return xcp, c, info
                wmc = float(np.dot(c, v)); wmp = float(np.dot(p, v)); wmw = float(np.dot(wbp, v))
                p -= dibp * wbp
                f1 += dibp * wmc
                f2 += 2.0 * dibp * wmp - dibp2 * wmw
            f2 = max(EPSMCH * f2_org, f2)
            if nleft > 0:
                dtm = -f1 / f2
            elif bnded and nfree_cnt == 0:
                f1 = 0.0; f2 = 0.0; dtm = 0.0
            else:
                dtm = -f1 / f2
        if not all_fixed:
            if dtm <= 0.0:
                dtm = 0.0
            tsum += dtm
            xcp += tsum * d
        if col > 0:
            c += dtm * p
        return xcp, c, 0

    # ---- formk: LEL^T factorization of the indefinite K matrix of the subspace problem
    def formk(self, free: np.ndarray) -> int:
        col, theta, sy = self.col, self.theta, self.sy
        m2 = 2 * col
        act = ~free
        WYf, WSf = self.wy[free, :col], self.ws[free, :col]
        WSa, WYa = self.ws[act, :col], self.wy[act, :col]
        yzzy = WYf.T @ WYf                       # Y'ZZ'Y
        saas = WSa.T @ WSa                       # S'AA'S
        la = WSa.T @ WYa                         # S'AA'Y  (strict lower part used)
        rz = WSf.T @ WYf                         # S'ZZ'Y  (upper part incl. diagonal used)
        wn = np.zeros((m2, m2))
        for iy in range(col):
            is_ = col + iy
            for jy in range(iy + 1):
                js = col + jy
                wn[jy, iy] = yzzy[iy, jy] / theta
                wn[js, is_] = saas[iy, jy] * theta
            for jy in range(iy):
                wn[jy, is_] = -la[iy, jy]
            for jy in range(iy, col):
                wn[jy, is_] = rz[iy, jy]
            wn[iy, iy] += sy[iy, iy]
        info = dpofa(wn, col)
        if info:
            return -1
        for js in range(col, m2):
            b = wn[:col, js].copy()
            dtrsl_upper(wn, col, b, transposed=True)
            wn[:col, js] = b
        for is_ in range(col, m2):
            for js in range(is_, m2):
                wn[is_, js] += float(np.dot(wn[:col, is_], wn[:col, js]))
        sub = wn[col:, col:].copy()
        info = dpofa(sub, col)
        wn[col:, col:] = sub
        if info:
            return -2
        self.wn = wn
        return 0

    # ---- cmprlb: reduced gradient at the Cauchy point
    def cmprlb(self, x, g, z, c, free):
        col, theta = self.col, self.theta
        r = -theta * (z[free] - x[free]) - g[free]
        mc, info = self.bmv(c)
        if info:
            return r, -8
        for j in range(col):
            a1 = mc[j]; a2 = theta * mc[col + j]
            r += self.wy[free, j] * a1 + self.ws[free, j] * a2
        return r, 0

    # ---- subsm: subspace minimization (v3.0 with the projected Newton step)
    def subsm(self, x, g, z, r, free):
        col, theta = self.col, self.theta
        l, u = self.l, self.u
        idx = np.nonzero(free)[0]
        nsub = len(idx)
        if nsub == 0:
            return z, 0
        d = r.copy()
        wv = np.zeros(2 * col)
        for i in range(col):
            wv[i] = float(np.dot(self.wy[idx, i], d))
            wv[col + i] = theta * float(np.dot(self.ws[idx, i], d))
        m2 = 2 * col
        if dtrsl_upper(self.wn, m2, wv, transposed=True):
            return z, 1
        wv[:col] = -wv[:col]
        if dtrsl_upper(self.wn, m2, wv, transposed=False):
            return z, 1
        for jy in range(col):
            js = col + jy
            d += self.wy[idx, jy] * wv[jy] / theta + self.ws[idx, jy] * wv[js]
        d *= 1.0 / theta
        # projected Newton step
        xp = z.copy()
        znew = z.copy()
        xk = z[idx] + d
        znew[idx] = np.minimum(u[idx], np.maximum(l[idx], xk))
        iword = int(np.any((znew[idx] == l[idx]) | (znew[idx] == u[idx])))
        self.iword = iword
        if iword == 0:
            return znew, 0
        dd_p = float(np.dot(znew - x, g))
        if dd_p > 0.0:
            znew = xp.copy()
            alpha = 1.0
            temp1 = alpha
            ibd = -1
            for i in range(nsub):
                k = idx[i]; dk = d[i]
                if dk < 0.0:
                    temp2 = l[k] - znew[k]
                    if temp2 >= 0.0:
                        temp1 = 0.0
                    elif dk * alpha < temp2:
                        temp1 = temp2 / dk
                elif dk > 0.0:
                    temp2 = u[k] - znew[k]
                    if temp2 <= 0.0:
                        temp1 = 0.0
                    elif dk * alpha > temp2:
                        temp1 = temp2 / dk
                if temp1 < alpha:
                    alpha = temp1
                    ibd = i
            if alpha < 1.0:
                dk = d[ibd]; k = idx[ibd]
                if dk > 0.0:
                    znew[k] = u[k]; d[ibd] = 0.0
                elif dk < 0.0:
                    znew[k] = l[k]; d[ibd] = 0.0
            znew[idx] += alpha * d
        return znew, 0

    # ---- matupd + formt
    def matupd(self, s_vec, y_vec, rr, dr, stp, dtd):
        m = self.m
        if self.iupdat <= m:
            self.col = self.iupdat
        else:                                   # drop the oldest pair
            self.ws[:, :m - 1] = self.ws[:, 1:m].copy(); self.wy[:, :m - 1] = self.wy[:, 1:m].copy()
            self.ss[:m - 1, :m - 1] = self.ss[1:m, 1:m].copy(); self.sy[:m - 1, :m - 1] = self.sy[1:m, 1:m].copy()
        col = self.col
        self.ws[:, col - 1] = s_vec
        self.wy[:, col - 1] = y_vec
        self.theta = rr / dr
        for j in range(col - 1):
            self.sy[col - 1, j] = float(np.dot(s_vec, self.wy[:, j]))
            self.ss[j, col - 1] = float(np.dot(self.ws[:, j], s_vec))
        self.ss[col - 1, col - 1] = dtd if stp == 1.0 else stp * stp * dtd
        self.sy[col - 1, col - 1] = dr

    def formt(self) -> int:
        col, theta, sy, ss = self.col, self.theta, self.sy, self.ss
        wt = np.zeros((self.m, self.m))
        for j in range(col):
            wt[0, j] = theta * ss[0, j]
        for i in range(1, col):
            for j in range(i, col):
                k1 = min(i, j)
                ddum = 0.0
                for k in range(k1):
                    ddum += sy[i, k] * sy[j, k] / sy[k, k]
                wt[i, j] = ddum + theta * ss[i, j]
        info = dpofa(wt, col)
        self.wt = wt
        return -3 if info else 0

    def reset_memory(self):
        self.col = 0; self.theta = 1.0; self.iupdat = 0; self.updatd = False


def minimize(fun_and_grad: Callable[[np.ndarray], Tuple[float, np.ndarray]], x0, lo, hi, *, m=10, ftol=0.5,
             pgtol=0.05, maxiter=15, maxfun=15000, maxls=20, trace=False) -> Result:
    """scipy.optimize._lbfgsb_py._minimize_lbfgsb + mainlb, for an all-boxed problem."""
    n = len(x0)
    # SciPy evaluates through ScalarFunction, which returns its cached (f, g) -- and does not count an
    # evaluation -- when asked for the x it evaluated last (a line search whose steps have shrunk
    # below rounding asks for the same point again).  result.nfev is that count.
    raw_fg = fun_and_grad
    cache = {"x": None, "f": None, "g": None, "n": 0}

    def fun_and_grad(xq):
        if cache["x"] is None or not np.array_equal(xq, cache["x"]):
            cache["x"] = np.array(xq, float)
            cache["f"], cache["g"] = raw_fg(cache["x"])
            cache["n"] += 1
        return cache["f"], cache["g"]

    S = Lbfgsb(n, m, lo, hi, ftol, pgtol, maxls)
    x = np.clip(np.asarray(x0, float), lo, hi)          # active(): project x0 into the box
    S.iwhere[:] = np.where(S.u - S.l <= 0.0, 3, 0)
    f, g = fun_and_grad(x); g = np.asarray(g, float).copy()
    nfev, nit = cache["n"], 0
    tr: List[np.ndarray] = []
    sbgnrm = S.projgr(x, g)
    if sbgnrm <= pgtol:
        return Result(x, f, 0, nfev, 0, "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL", tr)
    it = 0            # mainlb's iter
    free_prev = None
    while True:
        # ---------------- Cauchy point (label 222)
        z, c, info = S.cauchy(x, g, sbgnrm)
        if info:
            S.reset_memory(); continue
        free = S.iwhere <= 0
        nfree = int(np.count_nonzero(free))
        # ---------------- subspace minimization (label 333)
        if nfree != 0 and S.col != 0:
            info = S.formk(free)            # (the Fortran re-forms K only when the sets or the pairs changed)
            if info == 0:
                r, info = S.cmprlb(x, g, z, c, free)
            if info == 0:
                z, info = S.subsm(x, g, z, r, free)
            if info:
                S.reset_memory(); continue
        d = z - x
        # ---------------- line search (label 666, lnsrlb)
        dtd = float(np.dot(d, d)); dnorm = math.sqrt(dtd)
        if it == 0:
            stpmx = 1.0
        else:
            stpmx = BIG
            for i in range(n):
                a1 = d[i]
                if a1 < 0.0:
                    a2 = S.l[i] - x[i]
                    if a2 >= 0.0:
                        stpmx = 0.0
                    elif a1 * stpmx < a2:
                        stpmx = a2 / a1
                elif a1 > 0.0:
                    a2 = S.u[i] - x[i]
                    if a2 <= 0.0:
                        stpmx = 0.0
                    elif a1 * stpmx > a2:
                        stpmx = a2 / a1
        stp = 1.0                                   # boxed problem: stp = 1 also at iter 0
        t_x, r_g, fold = x.copy(), g.copy(), f
        ifun, iback = 0, 0
        st = DcsrchState()
        ls_task = "START"
        gd = gdold = 0.0
        ls_info = 0
        while True:
            gd = float(np.dot(g, d))
            if ifun == 0:
                gdold = gd
                if gd >= 0.0:
                    ls_info = -4
                    break
            stp, ls_task = dcsrch(f, gd, stp, FTOL, GTOL, XTOL, 0.0, stpmx, ls_task, st)
            if ls_task.startswith("CONV") or ls_task.startswith("WARN"):
                break
            if ls_task.startswith("ERROR"):
                # dcsrch rejected its inputs: lnsrlb keeps going as if FG were requested (task != CONV/WARN)
                pass
            ifun += 1; iback = ifun - 1
            if iback >= maxls:
                break
            x = z.copy() if stp == 1.0 else stp * d + t_x
            f, g = fun_and_grad(x); g = np.asarray(g, float).copy()
            nfev = cache["n"]
        if ls_info != 0 or iback >= maxls:
            # restore the previous iterate
            x, g, f = t_x, r_g, fold
            if S.col == 0:
                return Result(x, f, nit, nfev, 2, "ABNORMAL_TERMINATION_IN_LNSRCH", tr)
            S.reset_memory()
            continue
        # ---------------- new iterate
        it += 1
        sbgnrm = S.projgr(x, g)
        nit += 1
        if trace:
            tr.append(x.copy())
        # scipy driver checks after NEW_X
        stop_driver = None
        if nit >= maxiter:
            stop_driver = "STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT"
        elif nfev > maxfun:
            stop_driver = "STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT"
        if stop_driver:
            return Result(x, f, nit, nfev, 1, stop_driver, tr)
        # ---------------- termination tests (label 777)
        if sbgnrm <= pgtol:
            return Result(x, f, nit, nfev, 0, "CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL", tr)
        ddum = max(abs(fold), abs(f), 1.0)
        if (fold - f) <= S.tol * ddum:
            return Result(x, f, nit, nfev, 0, "CONVERGENCE: RELATIVE REDUCTION OF F <= FACTR*EPSMCH", tr)
        # ---------------- BFGS update
        r_y = g - r_g
        rr = float(np.dot(r_y, r_y))
        if stp == 1.0:
            dr = gd - gdold; ddum = -gdold
            s_vec = d
        else:
            dr = (gd - gdold) * stp; ddum = -gdold * stp
            s_vec = stp * d
        if dr <= EPSMCH * ddum:
            S.updatd = False
            continue
        S.updatd = True
        S.iupdat += 1
        S.matupd(s_vec, r_y, rr, dr, stp, dtd)
        if S.formt():
            S.reset_memory()
