#pragma once
// solve_device.hpp -- device code of the batched SE(3) MPC solve (included by solve_kernel.hip and monte_carlo.hip)
//
// batched SE(3) MPC solve: G lanes per problem (G = 8, 16, 32 or 64), 64 / G problems per wavefront.
//
// Replaces SE3MPCPlanner._solve_se3_mpc (reference planner.py:230-280): cold start (:329-359),
// box (:378-402), scipy.optimize.minimize(method="L-BFGS-B", jac=_objective_gradient, bounds,
// maxiter, gtol, ftol) with the reference's objective/gradient pair (:516-580), and
// _extract_solution_from_result (:582-654).  The L-BFGS-B is a from-scratch wavefront-parallel
// implementation of the published algorithm (Byrd-Lu-Nocedal-Zhu 1995; Morales-Nocedal 2011;
// More-Thuente line search), structured like oracle/lbfgsb_port.py which is pinned to SciPy.
//
// Mapping.  n = 9N decision variables, N <= 64.  Lane k of a problem's group owns horizon step k: its nine register
// slots are (block, axis) = P_k, V_k, T_k -- block, axis, objective term and box of every slot are compile-time facts,
// the only per-lane facts are "k >= N" (padding lane: variables fixed at 0) and "k == N - 1" (terminal position row).
// The group size G is the smallest of 8 / 16 / 32 / 64 that holds the horizon (the host widens it while that still
// fills the chip), so the reference's default horizon 6 packs eight problems into one wavefront, horizon 30 two.
// Everything that was wave-uniform in a one-problem-per-wavefront kernel (the L-BFGS scalars, the line-search
// state, the 2col x 2col middle matrices) is group-uniform here: groups of a wavefront are independent problems in
// (possibly) different branches.  Dot products / norms / argmins are per-lane partials + a group-local all-reduce
// (in-row DPP butterfly, + v_permlane16_swap for G = 32; G = 64: the whole-wavefront DPP reduction whose result
// is an SGPR).  The L-BFGS pairs S, Y live in LDS, each lane touching only its own elements (bank = lane:
// conflict free).  The m x m / 2m x 2m middle matrices and their Cholesky / triangular solves run in registers
// of every lane for col <= 2 (all a solve with the reference's options ever needs) and as "scalar sections" of the
// group's first lane on LDS beyond.  Every quantity that feeds a branch of the algorithm is computed in double; the
// _f32 entry point only stores S, Y and the results in float.
// No global memory is touched between reading (p0, v0, goal[, x0]) and writing the results.
#include <hip/hip_runtime.h>
#include <time.h>

#include "se3mpc_common.hpp"
#include <se3mpc_wave_ops.hpp>

// -DSE3MPC_SOLVE_PROFILE (tools/build_solve_profile.sh + tools/gpu_profile_solve_sections.py, never the shipped build): per-section cycle sums of every wavefront, written over
// the first 128 bytes of its `attitudes` output row.  Sections: 0 start-up + first evaluation, 1 later evaluations, 2 Cauchy point (rest), 3 subspace
// minimisation (rest), 4 line search without its evaluations, 5 convergence tests + BFGS update, 6 results, 7 total; 8 Cauchy pass 1, 9 closed-form
// pass, 10 p = W'd + first bmv, 11 breakpoint loop, 12 subspace formk + factor, 13 line search set-up (d, dtd, stpmx), 15 = number of crossings.
#ifdef SE3MPC_SOLVE_PROFILE
#define SE3MPC_TICK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) tsec[i] += now_ - tlast; tlast = now_; }
#define SE3MPC_COUNT(i) { if (lane == 0) tsec[i] += 1; }
#else
#define SE3MPC_TICK(i)
#define SE3MPC_COUNT(i)
#endif

#ifndef SE3MPC_SOLVE_WAVES
#define SE3MPC_SOLVE_WAVES 2      // resident wavefronts per SIMD the register allocation leaves room for
#endif

namespace se3mpc {

constexpr double kEps = 2.220446049250313e-16;   // DBL_EPSILON (epsmch)
constexpr double kBig = 1.0e10;
constexpr double kInf = __builtin_huge_val();

struct SolveDev {
  int N, n, has_goal, m, mlds, only_overflow, maxiter, maxls, maxfun, seq_cauchy;
  double dt, mass, grav, hover, wp, wv, wa, wT, term;
  double pos_b, v_max, txy, tz_lo, tz_hi;
  double pgtol, ftol;
  // products of the above formed once on the host (IEEE double, the very operations the kernel would do): as kernel arguments they are
  // scalar registers a vector instruction reads directly; computed in the kernel they would be vector registers -- in the packed
  // kernels every lane's own copy -- live from their first use to the end of the solve
  double two_wp, two_wv, two_wT, term_wp;
};

static SolveDev make_solve_dev(const se3mpc_params& p) {
  SolveDev d;
  d.N = p.horizon; d.n = 9 * p.horizon; d.has_goal = p.has_goal; d.m = p.max_corrections;
  d.mlds = p.max_corrections; d.only_overflow = 0; d.seq_cauchy = 0;
  d.maxiter = p.max_iterations; d.maxls = p.max_linesearch; d.maxfun = p.max_fun;
  d.dt = p.dt; d.mass = p.mass; d.grav = p.gravity; d.hover = p.mass * p.gravity;
  d.wp = p.position_weight; d.wv = p.velocity_weight; d.wa = p.acceleration_weight; d.wT = p.thrust_weight;
  d.term = p.terminal_factor;
  d.pos_b = p.position_bound; d.v_max = p.max_velocity; d.txy = p.max_thrust * sin(p.max_tilt_angle);
  d.tz_lo = p.min_thrust; d.tz_hi = p.max_thrust;
  d.pgtol = p.pgtol; d.ftol = p.ftol;
  d.two_wp = 2.0 * d.wp; d.two_wv = 2.0 * d.wv; d.two_wT = 2.0 * d.wT; d.term_wp = d.term * d.wp;
  return d;
}

// Moré-Thuente safeguarded step (MINPACK-2 dcstep).
__device__ __forceinline__ void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                       double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / fabs(dx));
  double stpf;
  if (fp > fx) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta, qq = ((gamma - dx) + gamma) + dp, r = p / qq;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    stpf = (fabs(stpc - stx) < fabs(stpq - stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, qq = ((gamma - dp) + gamma) + dx, r = p / qq;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
    brackt = true;
  } else if (fabs(dp) < fabs(dx)) {
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, qq = (gamma + (dx - dp)) + gamma, r = p / qq;
    double stpc;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      stpf = (fabs(stpc - stp) < fabs(stpq - stp)) ? stpc : stpq;
      if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = fmax(fabs(theta), fmax(fabs(dy), fabs(dp)));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta, qq = ((gamma - dp) + gamma) + dy, r = p / qq;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) stpf = stpmax;
    else stpf = stpmin;
  }
  // interval update, written as value selects (as `if (..) {sty = ..} else {stx = ..}` the compiler stores through a selected pointer,
  // which puts fx, dx, fy, dy into scratch memory)
  const bool up = fp > fx, swap = !up && sgnd < 0.0;
  const double nsty = up ? stp : (swap ? stx : sty), nfy = up ? fp : (swap ? fx : fy), ndy = up ? dp : (swap ? dx : dy);
  const double nstx = up ? stx : stp, nfx = up ? fx : fp, ndx = up ? dx : dp;
  sty = nsty; fy = nfy; dy = ndy;
  stx = nstx; fx = nfx; dx = ndx;
  stp = stpf;
}

// State of one line search (dcsrch's isave/dsave); every lane holds an identical copy.
struct LineSearch {
  bool brackt; int stage;
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};
enum { LS_FG = 0, LS_CONV = 1, LS_WARN = 2, LS_ERROR = 3 };

__device__ __forceinline__ int dcsrch(double f, double g, double& stp, double stpmin, double stpmax, bool start, LineSearch& s) {
  const double ftol = 1.0e-3, gtol = 0.9, xtol = 0.1, xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  if (start) {
    if (stp < stpmin || stp > stpmax || g >= 0.0 || stpmax < stpmin) return LS_ERROR;
    s.brackt = false; s.stage = 1; s.finit = f; s.ginit = g; s.gtest = ftol * g;
    s.width = stpmax - stpmin; s.width1 = s.width / p5;
    s.stx = 0.0; s.fx = f; s.gx = g; s.sty = 0.0; s.fy = f; s.gy = g;
    s.stmin = 0.0; s.stmax = stp + xtrapu * stp;
    return LS_FG;
  }
  const double ftest = s.finit + stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  int task = LS_FG;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) task = LS_WARN;
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = LS_WARN;
  if (stp == stpmax && f <= ftest && g <= s.gtest) task = LS_WARN;
  if (stp == stpmin && (f > ftest || g >= s.gtest)) task = LS_WARN;
  if (f <= ftest && fabs(g) <= gtol * (-s.ginit)) task = LS_CONV;
  if (task != LS_FG) return task;
  // ONE dcstep call on local copies (the modified function of stage 1 or the function itself): handing dcstep either locals or the
  // members by reference made the compiler keep fx, gx, fy, gy in scratch memory behind a selected pointer.
  const bool modified = s.stage == 1 && f <= s.fx && f > ftest;
  double fxv = s.fx, gxv = s.gx, fyv = s.fy, gyv = s.gy, fv = f, gv = g;
  if (modified) {
    fv = f - stp * s.gtest;
    fxv = s.fx - s.stx * s.gtest; fyv = s.fy - s.sty * s.gtest;
    gv = g - s.gtest;
    gxv = s.gx - s.gtest; gyv = s.gy - s.gtest;
  }
  dcstep(s.stx, fxv, gxv, s.sty, fyv, gyv, stp, fv, gv, s.brackt, s.stmin, s.stmax);
  if (modified) {
    fxv = fxv + s.stx * s.gtest; fyv = fyv + s.sty * s.gtest;
    gxv = gxv + s.gtest; gyv = gyv + s.gtest;
  }
  s.fx = fxv; s.gx = gxv; s.fy = fyv; s.gy = gyv;
  if (s.brackt) {
    if (fabs(s.sty - s.stx) >= p66 * s.width1) stp = s.stx + p5 * (s.sty - s.stx);
    s.width1 = s.width; s.width = fabs(s.sty - s.stx);
  }
  if (s.brackt) { s.stmin = fmin(s.stx, s.sty); s.stmax = fmax(s.stx, s.sty); }
  else { s.stmin = stp + xtrapl * (stp - s.stx); s.stmax = stp + xtrapu * (stp - s.stx); }
  stp = fmax(stp, stpmin);
  stp = fmin(stp, stpmax);
  if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) || (s.brackt && s.stmax - s.stmin <= xtol * s.stmax)) stp = s.stx;
  return LS_FG;
}

// ---- scalar sections (lane 0 only, operands in LDS) ------------------------------------------
// LINPACK dpofa on the leading n x n block of a (row stride ld): upper factor in the upper triangle.
__device__ inline int dpofa(double* a, int ld, int n) {
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int k = 0; k < j; ++k) {
      double t = a[k * ld + j];
      for (int i = 0; i < k; ++i) t -= a[i * ld + k] * a[i * ld + j];
      t = t / a[k * ld + k];
      a[k * ld + j] = t;
      s += t * t;
    }
    s = a[j * ld + j] - s;
    if (s <= 0.0) return j + 1;
    a[j * ld + j] = sqrt(s);
  }
  return 0;
}
// LINPACK dtrsl, t upper triangular (row stride ld): transposed ? t' x = b : t x = b, in place.
__device__ inline int dtrsl_upper(const double* t, int ld, int n, double* b, int bstride, bool transposed) {
  for (int j = 0; j < n; ++j) if (t[j * ld + j] == 0.0) return j + 1;
  if (!transposed) {
    b[(n - 1) * bstride] = b[(n - 1) * bstride] / t[(n - 1) * ld + (n - 1)];
    for (int j = n - 2; j >= 0; --j) {
      const double temp = -b[(j + 1) * bstride];
      for (int i = 0; i <= j; ++i) b[i * bstride] += temp * t[i * ld + (j + 1)];
      b[j * bstride] = b[j * bstride] / t[j * ld + j];
    }
  } else {
    b[0] = b[0] / t[0];
    for (int j = 1; j < n; ++j) {
      double s = b[j * bstride];
      for (int i = 0; i < j; ++i) s -= t[i * ld + j] * b[i * bstride];
      b[j * bstride] = s / t[j * ld + j];
    }
  }
  return 0;
}
// bmv: p = M v for the 2col x 2col middle matrix (sy, wt with row stride m).
__device__ inline int bmv(const double* sy, const double* wt, int m, int col, const double* v, double* p) {
  if (col == 0) return 0;
  p[col] = v[col];
  for (int i = 1; i < col; ++i) {
    double s = 0.0;
    for (int k = 0; k < i; ++k) s += sy[i * m + k] * v[k] / sy[k * m + k];
    p[col + i] = v[col + i] + s;
  }
  int info = dtrsl_upper(wt, m, col, p + col, 1, true);
  if (info) return info;
  for (int i = 0; i < col; ++i) p[i] = v[i] / sqrt(sy[i * m + i]);
  info = dtrsl_upper(wt, m, col, p + col, 1, false);
  if (info) return info;
  for (int i = 0; i < col; ++i) p[i] = -p[i] / sqrt(sy[i * m + i]);
  for (int i = 0; i < col; ++i) {
    double s = 0.0;
    for (int k = i + 1; k < col; ++k) s += sy[k * m + i] * p[col + k] / sy[i * m + i];
    p[i] += s;
  }
  return 0;
}


// ---- register-resident small-matrix routines -------------------------------------------------
// With the reference's options a solve stops after 1-3 iterations, i.e. the L-BFGS memory holds col = 1 or 2 pairs whenever the
// middle matrices are used at all.  For those sizes the 2col x 2col algebra (bmv, the LEL' factorisation of formk, the two
// triangular solves of subsm, formt's Cholesky) is a handful of flops whose cost on lane 0 was pure LDS latency: every operand a
// dependent LDS round trip, every result a write + barrier + broadcast read (38 % of the kernel's wave-cycles sat in s_waitcnt).
// Here every lane runs the same algebra on wave-uniform values held in REGISTERS (compile-time indices, fully unrolled): operands
// are fetched once per section with independent broadcast reads, results are already in every lane, and only the state that must
// survive the iteration (sy, ss, wt; pv, cv inside a Cauchy search) is written back, by lane 0.  The operation order is that of
// the LDS routines above (LINPACK dpofa / dtrsl, bmv), so both paths produce the same bits.  col > kFastCol keeps the LDS path.
constexpr int kFastCol = 2;

template <int C>
struct MidRegs {
  double sy[C][C];   // S'Y, lower triangle + diagonal
  double wt[C][C];   // Cholesky factor of theta*S'S + L D^-1 L', upper triangle
};

template <int C>
__device__ __forceinline__ void load_mid(const double* sy, const double* wt, int m, MidRegs<C>& M) {
#pragma unroll
  for (int i = 0; i < C; ++i) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
      M.sy[i][k] = (k <= i) ? sy[i * m + k] : 0.0;
      M.wt[i][k] = (k >= i) ? wt[i * m + k] : 0.0;
    }
  }
}

// bmv on registers: p = M v for the 2C x 2C middle matrix
template <int C>
__device__ __forceinline__ int bmv_regs(const MidRegs<C>& M, const double (&v)[2 * C], double (&p)[2 * C]) {
  p[C] = v[C];
#pragma unroll
  for (int i = 1; i < C; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) s += M.sy[i][k] * v[k] / M.sy[k][k];
    p[C + i] = v[C + i] + s;
  }
#pragma unroll
  for (int j = 0; j < C; ++j) if (M.wt[j][j] == 0.0) return j + 1;
  p[C] = p[C] / M.wt[0][0];                                   // dtrsl, transposed
#pragma unroll
  for (int j = 1; j < C; ++j) {
    double s = p[C + j];
#pragma unroll
    for (int i = 0; i < j; ++i) s -= M.wt[i][j] * p[C + i];
    p[C + j] = s / M.wt[j][j];
  }
#pragma unroll
  for (int i = 0; i < C; ++i) p[i] = v[i] / sqrt(M.sy[i][i]);
  p[C + C - 1] = p[C + C - 1] / M.wt[C - 1][C - 1];            // dtrsl, not transposed
#pragma unroll
  for (int j = C - 2; j >= 0; --j) {
    const double temp = -p[C + j + 1];
#pragma unroll
    for (int i = 0; i <= j; ++i) p[C + i] += temp * M.wt[i][j + 1];
    p[C + j] = p[C + j] / M.wt[j][j];
  }
#pragma unroll
  for (int i = 0; i < C; ++i) p[i] = -p[i] / sqrt(M.sy[i][i]);
#pragma unroll
  for (int i = 0; i < C; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = i + 1; k < C; ++k) s += M.sy[k][i] * p[C + k] / M.sy[i][i];
    p[i] += s;
  }
  return 0;
}

// LINPACK dpofa on the block a[OFF .. OFF+C)[OFF .. OFF+C) of an N2 x N2 register matrix (upper factor in the upper triangle)
template <int N2, int OFF, int C>
__device__ __forceinline__ int dpofa_regs(double (&a)[N2][N2]) {
#pragma unroll
  for (int j = 0; j < C; ++j) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < j; ++k) {
      double t = a[OFF + k][OFF + j];
#pragma unroll
      for (int i = 0; i < k; ++i) t -= a[OFF + i][OFF + k] * a[OFF + i][OFF + j];
      t = t / a[OFF + k][OFF + k];
      a[OFF + k][OFF + j] = t;
      s += t * t;
    }
    s = a[OFF + j][OFF + j] - s;
    if (s <= 0.0) return j + 1;
    a[OFF + j][OFF + j] = sqrt(s);
  }
  return 0;
}

// dtrsl on a full N x N upper-triangular register matrix, right-hand side b
template <int N>
__device__ __forceinline__ int dtrsl_regs(const double (&t)[N][N], double (&b)[N], bool transposed) {
#pragma unroll
  for (int j = 0; j < N; ++j) if (t[j][j] == 0.0) return j + 1;
  if (!transposed) {
    b[N - 1] = b[N - 1] / t[N - 1][N - 1];
#pragma unroll
    for (int jr = 0; jr < N - 1; ++jr) {                      // j = N-2 .. 0 (counted upwards: the descending form is left rolled, and b[] in scratch)
      const int j = N - 2 - jr;
      const double temp = -b[j + 1];
#pragma unroll
      for (int i = 0; i < N; ++i) if (i <= j) b[i] += temp * t[i][j + 1];
      b[j] = b[j] / t[j][j];
    }
  } else {
    b[0] = b[0] / t[0][0];
#pragma unroll
    for (int j = 1; j < N; ++j) {
      double s = b[j];
#pragma unroll
      for (int i = 0; i < j; ++i) s -= t[i][j] * b[i];
      b[j] = s / t[j][j];
    }
  }
  return 0;
}

// The factorisation half of formk on registers: wn = [ K11  K12 ; .  K22 ] (upper triangle) -> LEL' factor, as the LDS code:
// dpofa(K11); K12 <- R11^-T K12 column by column; K22 += K12' K12; dpofa(K22).  0, -1 or -2.
template <int C>
__device__ __forceinline__ int formk_factor_regs(double (&wn)[2 * C][2 * C]) {
  if (dpofa_regs<2 * C, 0, C>(wn)) return -1;
#pragma unroll
  for (int js = C; js < 2 * C; ++js) {                        // dtrsl_upper(wn, ld, col, wn + js, ld, transposed)
    wn[0][js] = wn[0][js] / wn[0][0];
#pragma unroll
    for (int j = 1; j < C; ++j) {
      double sacc = wn[j][js];
#pragma unroll
      for (int i = 0; i < j; ++i) sacc -= wn[i][j] * wn[i][js];
      wn[j][js] = sacc / wn[j][j];
    }
  }
#pragma unroll
  for (int is = C; is < 2 * C; ++is) {
#pragma unroll
    for (int js = is; js < 2 * C; ++js) {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k) sacc += wn[k][is] * wn[k][js];
      wn[is][js] += sacc;
    }
  }
  if (dpofa_regs<2 * C, C, C>(wn)) return -2;
  return 0;
}


template <int C>
struct ColTag { static constexpr int value = C; };

// A lane's row of one L-BFGS pair in LDS: (s_0, y_0, s_1, y_1, ... s_8, y_8) [+ padding], 16-byte aligned.  Read / written whole with
// 16-byte DS accesses (ds_read_b128 / ds_write_b128): all of a pass's reads are in flight before the first value is used.
template <typename IO>
__device__ __forceinline__ void load_row(const IO* row, IO (&w)[2 * 9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef IO vec4 __attribute__((ext_vector_type(4)));
  typedef IO vec2 __attribute__((ext_vector_type(2)));
  const vec4* r4 = reinterpret_cast<const vec4*>(row);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const vec4 v = r4[i]; w[4 * i] = v[0]; w[4 * i + 1] = v[1]; w[4 * i + 2] = v[2]; w[4 * i + 3] = v[3]; }
  const vec2 t = *reinterpret_cast<const vec2*>(row + 16);
  w[16] = t[0]; w[17] = t[1];
#else
  for (int i = 0; i < 18; ++i) w[i] = row[i];
#endif
}
template <typename IO>
__device__ __forceinline__ void store_row(IO* row, const IO (&w)[2 * 9]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef IO vec4 __attribute__((ext_vector_type(4)));
  typedef IO vec2 __attribute__((ext_vector_type(2)));
  vec4* r4 = reinterpret_cast<vec4*>(row);
#pragma unroll
  for (int i = 0; i < 4; ++i) { vec4 v; v[0] = w[4 * i]; v[1] = w[4 * i + 1]; v[2] = w[4 * i + 2]; v[3] = w[4 * i + 3]; r4[i] = v; }
  vec2 t; t[0] = w[16]; t[1] = w[17];
  *reinterpret_cast<vec2*>(row + 16) = t;
#else
  for (int i = 0; i < 18; ++i) row[i] = w[i];
#endif
}

// Compiler-only memory barrier (no instruction).  The S, Y pairs are read from LDS in four phases of the subspace step; without this the
// compiler merges the four reads of every element and keeps all 2 * col * 9 values in registers across the whole section (36 VGPRs in the
// float kernel, 72 in the double one), which is what pushed the packed kernels over 256 registers.  LDS reads are cheap; spills are not.
__device__ __forceinline__ void reload_lds() { asm volatile("" ::: "memory"); }

// Identity the compiler cannot see through (no instruction).  The line search saves x_old = x; everything it could recompute from x_old
// (the step z - x_old, the old gradient) the compiler would otherwise recognise as values it already holds and KEEP them in registers
// across the search -- 36 VGPRs at the kernel's pressure peak.  With x_old opaque those values are dead during the search and are
// formed again (same expressions, same bits) where they are used.
__device__ __forceinline__ double opaque(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// ---- the solver ------------------------------------------------------------------------------
constexpr int kSlots = 9;            // register slots of a lane: slot j = (block j / 3, axis j % 3) of the lane's horizon step

// box of slot j (planner.py:378-402); j is a compile-time constant wherever this is called from an unrolled loop
__device__ __forceinline__ double box_lo(const SolveDev& q, int j) { return j < 3 ? -q.pos_b : (j < 6 ? -q.v_max : (j < 8 ? -q.txy : q.tz_lo)); }
__device__ __forceinline__ double box_hi(const SolveDev& q, int j) { return j < 3 ? q.pos_b : (j < 6 ? q.v_max : (j < 8 ? q.txy : q.tz_hi)); }

// doubles of LDS per problem for the small matrices with storage for m pairs: sy, ss, wt [m][m], wn [2m][2m], pv, cv, vv, wbp, wv [2m],
// sc [8]; made odd so that the broadcast reads of the (up to eight) problems of a wavefront fall into different banks
__host__ __device__ constexpr int small_doubles(int m) { return (7 * m * m + 10 * m + 8) | 1; }
// values per lane and pair in the S, Y image (see the kernel's LDS carve-up), and where the image starts (16-byte aligned)
template <typename IO> __host__ __device__ constexpr int pair_row_values() { return sizeof(IO) == 4 ? 20 : 18; }
__host__ __device__ constexpr size_t pairs_offset_bytes(int P, int m) { return ((size_t)P * small_doubles(m) * sizeof(double) + 15) / 16 * 16; }


// bytes of LDS one wavefront's solver image needs with storage for m L-BFGS pairs at G lanes per problem
static inline size_t solve_lds_bytes(int m, int G, size_t io_size) {
  const size_t pairs = (size_t)m * kWave * (io_size == 4 ? pair_row_values<float>() : pair_row_values<double>()) * io_size;
  return pairs_offset_bytes(kWave / G, m) + pairs;
}

}  // namespace se3mpc
