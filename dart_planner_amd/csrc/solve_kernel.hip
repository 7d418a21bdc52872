// solve_kernel.hip -- batched SE(3) MPC solve: G lanes per problem (G = 8, 16, 32 or 64), 64 / G problems per wavefront.
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
__device__ int dpofa(double* a, int ld, int n) {
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
__device__ int dtrsl_upper(const double* t, int ld, int n, double* b, int bstride, bool transposed) {
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
__device__ int bmv(const double* sy, const double* wt, int m, int col, const double* v, double* p) {
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

// 2nd launch-bounds argument = wavefronts per SIMD the register allocation must leave room for: two
// resident wavefronts per SIMD (<= 256 VGPR+AGPR each) overlap each other's DPP/LDS latencies.
template <typename IO, int G>
__global__ void __launch_bounds__(64, SE3MPC_SOLVE_WAVES)
solve_kernel(SolveDev q, int B, const IO* __restrict__ p0g, const IO* __restrict__ v0g, const IO* __restrict__ goalg,
             const IO* __restrict__ x0g, IO* __restrict__ Xg, se3mpc_solve_info* __restrict__ infog,
             IO* __restrict__ accg, IO* __restrict__ attg, IO* __restrict__ ratesg, IO* __restrict__ thrustg,
             unsigned long long* doneg, unsigned long long ticket) {
  HIP_DYNAMIC_SHARED(unsigned char, lds_raw)
  constexpr int P = kWave / G, J = kSlots;
  const int lane = lane_id();
  const int k = lane & (G - 1);            // horizon step owned by this lane
  const int grp = lane / G;                // problem slot inside the wavefront
  const int pb = blockIdx.x * P + grp;     // problem index
  if (pb >= B) return;                     // (a whole group leaves together)
  const int n = q.n, N = q.N, n3 = 3 * q.N;
  // Two-tier memory: the first launch gives every problem LDS for `mlds` L-BFGS pairs (2-4: all a solve with the reference's options
  // ever stores -- it stops after 1-3 iterations, i.e. at most two updates) so that two wavefronts per SIMD fit a CU's LDS; a
  // problem that needs a pair more leaves with task = SE3MPC_TASK_OVERFLOW and is re-solved from scratch by the second launch
  // (mlds = maxcor, only_overflow = 1), in which every other group exits at once.  `m` below is the STORAGE bound; the
  // algorithm's memory is still q.m.
  if (q.only_overflow && infog[pb].task != SE3MPC_TASK_OVERFLOW) return;
  const int m = q.mlds;
  // LDS carve-up: per-problem small matrices (doubles) first, then the S / Y pairs of the whole wavefront in the IO type
  double* sy = reinterpret_cast<double*>(lds_raw) + grp * small_doubles(m);   // [m][m]  S'Y (lower triangle used)
  double* ss = sy + m * m;                             // [m][m]  S'S (upper triangle used)
  double* wt = ss + m * m;                             // [m][m]  Cholesky factor of theta*S'S + L D^-1 L'
  double* wn = wt + m * m;                             // [2m][2m] LEL' factor of the subspace K matrix
  double* pv = wn + 4 * m * m;                         // [2m]  p = W'd            (Cauchy)
  double* cv = pv + 2 * m;                             // [2m]  c = W'(xcp - x)    (Cauchy)
  double* vv = cv + 2 * m;                             // [2m]  scratch M*...
  double* wbp = vv + 2 * m;                            // [2m]  row of W at a breakpoint
  double* wv = wbp + 2 * m;                            // [2m]  subspace rhs
  double* sc = wv + 2 * m;                             // [8]   scalars handed out of scalar sections
  // pairs: [m][64 lanes][LS] -- a lane's row of one pair holds its nine (s_j, y_j) couples contiguously (18 values + padding to a
  // multiple of 16 bytes whose dword stride, 20 for float / 36 for double, keeps 16-byte reads of the 16 lanes of a group in
  // disjoint banks): one pass over a pair is five (float) or nine (double) ds_read_b128 per lane instead of eighteen ds_read_b32,
  // each of which the register-starved schedule waited for on its own (SQ_WAIT_ANY was 45 % of a lone wavefront's cycles).
  constexpr int LS = pair_row_values<IO>();
  IO* pairs = reinterpret_cast<IO*>(lds_raw + pairs_offset_bytes(P, m));

  // ---- per-lane facts and problem data
  const bool live = k < N;                 // k >= N: padding lane, its nine variables are fixed at 0
  const bool last = k == N - 1;            // terminal position row
  double goal[3] = {0.0, 0.0, 0.0};
  if (q.has_goal) {
#pragma unroll
    for (int a = 0; a < 3; ++a) goal[a] = (double)goalg[pb * 3 + a];
  }
  const bool cold = x0g == nullptr;
  double ps[3] = {0.0, 0.0, 0.0}, vs[3] = {0.0, 0.0, 0.0};
  if (cold) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { ps[a] = (double)p0g[pb * 3 + a]; vs[a] = (double)v0g[pb * 3 + a]; }
  }
  // the goal coordinate the objective sees: 0 on a padding lane, whose position slots then have gradient 0 like its other slots
  double gl[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) gl[a] = live ? goal[a] : 0.0;

#ifdef SE3MPC_SOLVE_PROFILE
  __shared__ unsigned long long tsec[16];                     // in LDS (lane 0 adds): eight SGPR pairs of counters would change the register allocation measured
  if (lane < 16) tsec[lane] = 0;
  __syncthreads();
  unsigned long long tlast = __builtin_readcyclecounter();
  const unsigned long long tstart = tlast;
#endif
  // ---- cold start (planner.py:329-359) or caller x0, projected into the box (L-BFGS-B `active`)
  double x[J], g[J], z[J], d[J], xo[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int blk = j / 3, a = j % 3;
    double xv = 0.0;
    if (!cold) {
      xv = live ? (double)x0g[(size_t)pb * n + blk * n3 + 3 * k + a] : 0.0;
    } else {
      const double denom = (double)(N - 1 > 1 ? N - 1 : 1);
      if (blk == 0) {
        const double alpha = (double)k / denom;
        xv = q.has_goal ? (1.0 - alpha) * ps[a] + alpha * goal[a] : ps[a];
      } else if (blk == 1) {
        if (q.has_goal) {
          const double a1 = (double)k / denom, a0 = (double)(k - 1) / denom;
          xv = (((1.0 - a1) * ps[a] + a1 * goal[a]) - ((1.0 - a0) * ps[a] + a0 * goal[a])) / q.dt;
        }
        xv = (k == 0) ? vs[a] : xv;
      } else {
        xv = (a == 2) ? q.hover : 0.0;
      }
    }
    x[j] = live ? fmin(fmax(xv, box_lo(q, j)), box_hi(q, j)) : 0.0;
    g[j] = 0.0; z[j] = x[j]; d[j] = 0.0; xo[j] = x[j];
  }

  // the reference's gradient (planner.py:552-580) of one slot.  The same expression wherever a gradient value is needed again (the
  // previous iterate's gradient in the BFGS update and after a failed line search is RECOMPUTED from the saved x, not kept in nine more
  // register pairs); contraction off so that both places round the product the same way.
  auto grad_of = [&](int j, double xv) -> double {
#pragma clang fp contract(off)
    if (j < 3) return q.has_goal ? 2.0 * q.wp * (xv - gl[j]) : 0.0;
    if (j < 6) return 2.0 * q.wv * xv;
    return 2.0 * q.wT * xv;
  };
  // objective (planner.py:516-550) and gradient at x; with_gd: also g(x)'d, its reduction interleaved with the objective's (the
  // line search wants both)
  // A line-search evaluation (with_gd) does not store the gradient: it only needs g(x)'d, and the nine gradient values are formed again
  // from x when the search has ended -- so that no gradient registers are live across the line search, the kernel's pressure peak.
  double gd_fused = 0.0;
  auto eval_fg = [&](bool with_gd = false) -> double {
    double part = 0.0, gdp = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (q.has_goal) {
        const double e = x[a] - gl[a];
        double fj = q.wp * (e * e);
        const double ft = q.term * q.wp * (e * e);
        fj = last ? fj + ft : fj;
        part += fj;
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double xv = x[3 + a];
      part += q.wv * (xv * xv);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double xv = x[6 + a];
      const double ac = xv / q.mass - (a == 2 ? q.grav : 0.0);
      const double dv = xv - (a == 2 ? q.hover : 0.0);
      part += q.wa * (ac * ac) + q.wT * (dv * dv);
    }
    part = live ? part : 0.0;
    if (with_gd) {
#pragma unroll
      for (int j = 0; j < J; ++j) gdp += grad_of(j, x[j]) * (z[j] - xo[j]);      // d = z - x_old, formed here (see opaque())
      double r2[2] = {part, gdp};
      group_sum_n<G, 2>(r2);
      gd_fused = r2[1];
      return r2[0];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = grad_of(j, x[j]);
    return group_sum<G>(part);
  };
  // projected gradient norm (projgr)
  auto projgr = [&]() -> double {
    double mx = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const double gj = g[j];
      const double up = fmax(x[j] - box_hi(q, j), gj), dn = fmin(x[j] - box_lo(q, j), gj);      // both sides, then one select: no divergent branches
      const double gi = gj < 0.0 ? up : dn;
      mx = fmax(mx, fabs(gi));
    }
    mx = live ? mx : 0.0;
    return group_max<G>(mx);
  };
  auto ROW = [&](int c) -> IO* { return static_cast<IO*>(__builtin_assume_aligned(pairs + (c * kWave + lane) * LS, 16)); };
  auto WS = [&](int c, int j) -> IO& { return ROW(c)[2 * j]; };
  auto WY = [&](int c, int j) -> IO& { return ROW(c)[2 * j + 1]; };

  // ---- L-BFGS-B state (identical in every lane of the group)
  int col = 0, iupdat = 0, iter = 0, nit = 0;
  double theta = 1.0;
  int iwhere[J];
#pragma unroll
  for (int j = 0; j < J; ++j) iwhere[j] = live ? 0 : 3;   // padding = fixed variables
  // nfev counts like scipy's ScalarFunction: asking again for the x evaluated last (a line search whose steps shrank below
  // rounding) returns the same (f, g) and is not counted.  x_is_last: the registers x hold the x of the last evaluation.
  int nfev = 1;
  bool x_is_last = true;
  double f = eval_fg();
  double sbgnrm = projgr();
  int status = 0, task = 0;
  double fold = f;

  if (sbgnrm <= q.pgtol) { task = SE3MPC_TASK_CONV_PGTOL; status = 0; }

  SE3MPC_TICK(0)
  int guard = 0;                           // every pass either ends an iteration or drops the memory; bounded anyway
  while (task == 0) {
    if (++guard > 4 * (q.maxiter + 8)) { task = SE3MPC_TASK_ABNORMAL; status = 2; break; }
    // ===================================================================== Cauchy point
    int info = 0;
    {
      // pass 1: search direction d = -g on the free variables, breakpoints, p = W'd, f1 = -d'd
      double tbp[J];
      double f1p = 0.0;
      int nbr = 0;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        // written as value selects throughout: as nested ifs this loop became ~50 exec-mask instructions per slot
        const double lo = box_lo(q, j), hi = box_hi(q, j);
        const double neggi = -g[j];
        const double tl = x[j] - lo, tu = hi - x[j];
        const bool xlower = tl <= 0.0, xupper = tu <= 0.0;
        const int atlo = neggi <= 0.0 ? 1 : 0, athi = neggi >= 0.0 ? 2 : 0, flat = fabs(neggi) <= 0.0 ? -3 : 0;
        const int iwb = xlower ? atlo : (xupper ? athi : flat);
        const int iw = live ? iwb : 3;                       // (nothing of the previous iterate's iwhere survives but "padding lane" = fixed)
        iwhere[j] = iw;
        const bool moving = iw == 0;
        const double dj = moving ? neggi : 0.0;
        d[j] = dj;
        f1p -= dj * dj;
        // breakpoint t_j = num / |g_j| (tl / (-neggi) for a descending, tu / neggi for an ascending variable).  Only the numerator is kept here:
        // the quotient is formed where a breakpoint's VALUE is needed -- never on the first iterate (theta = 1: t_j <= 1 <=> num <= |g_j|, exactly)
        // and not on an iterate whose Cauchy point lies before every breakpoint, which a product decides (below): the common case
        const bool brk = moving & (neggi != 0.0);
        tbp[j] = brk ? (neggi < 0.0 ? tl : tu) : kInf;
        nbr += brk ? 1 : 0;
        z[j] = x[j];
      }
      // tbp[] holds numerators until this turns them into breakpoints (nine f64 divisions per lane)
      auto breakpoints = [&]() {
#pragma unroll
        for (int j = 0; j < J; ++j) tbp[j] = tbp[j] / fabs(g[j]);            // +inf / |g| stays +inf (|g| is finite); brk implies g != 0
      };
      SE3MPC_TICK(8)
      if (col == 0 && !q.seq_cauchy) {
        // No L-BFGS pairs yet: B = theta*I and the piecewise quadratic along the projected path is
        //   m(t) = sum_i g_i^2 (theta*tau_i^2/2 - tau_i),  tau_i = min(t, t_i),
        // whose derivative sum_{t_i > t} g_i^2 (theta*t - 1) is negative on [0, 1/theta): the sequential
        // search of the published algorithm (f1_k = -(D - S_k)(1 - theta*t_k), f2_k = theta*(D - S_k), hence
        // dtm_k = 1/theta - t_k at every breakpoint) crosses exactly the breakpoints t_i <= 1/theta and stops
        // at t = 1/theta.  This is where ~170 of a typical solve's ~172 crossings happen (all of them at
        // the first iterate), so they are taken in one parallel pass instead of 170 group reductions.
        if (sbgnrm > 0.0) {
          const double tstar = 1.0 / theta;
          const bool unit = theta == 1.0;                                  // the first iterate of every solve
          if (!unit) breakpoints();
#pragma unroll
          for (int j = 0; j < J; ++j) {
            // fl(num / |g|) <= 1 <=> num <= |g| (rounding is monotone and fl(1) = 1): no quotient while theta is 1
            const bool reached = unit ? tbp[j] <= fabs(g[j]) : tbp[j] <= tstar;
            const bool hit = (iwhere[j] == 0) & reached;                   // d is 0 wherever iwhere != 0: z + tstar*d leaves those alone
            const bool upw = d[j] > 0.0;
            z[j] = hit ? (upw ? box_hi(q, j) : box_lo(q, j)) : z[j] + tstar * d[j];
            iwhere[j] = hit ? (upw ? 2 : 1) : iwhere[j];
            d[j] = hit ? 0.0 : d[j];
          }
        }
        SE3MPC_TICK(9)
      } else {
      double f1 = group_sum<G>(f1p);
      const int nbreak = group_sum_i32<G>(nbr);
      // p = W'd (2col group reductions), second half scaled by theta
      for (int c = 0; c < col; ++c) {
        double a1 = 0.0, a2 = 0.0;
        IO w[2 * J];
        load_row(ROW(c), w);
#pragma unroll
        for (int j = 0; j < J; ++j) { a1 += (double)w[2 * j + 1] * d[j]; a2 += (double)w[2 * j] * d[j]; }
        double r2[2] = {a1, a2};
        group_sum_n<G, 2>(r2);
        a1 = r2[0]; a2 = r2[1];
        if (k == 0) { pv[c] = a1; pv[col + c] = theta * a2; cv[c] = 0.0; cv[col + c] = 0.0; }
      }
      group_sync<G>();
      if (sbgnrm > 0.0 && nbreak > 0) {
        double f2 = -theta * f1;
        const double f2_org = f2;
        if (col > 0 && col <= kFastCol) {
          auto init_fast = [&](auto tag) {
            constexpr int C = decltype(tag)::value;
            MidRegs<C> M;
            load_mid<C>(sy, wt, m, M);
            double v[2 * C], pr[2 * C];
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) v[i] = pv[i];
            info = bmv_regs<C>(M, v, pr);
            double dot = 0.0;
#pragma unroll
            for (int i = 0; i < 2 * C; ++i) dot += pr[i] * v[i];
            f2 -= dot;
          };
          if (col == 1) init_fast(ColTag<1>{}); else init_fast(ColTag<2>{});
        } else if (col > 0) {
          if (k == 0) {
            const int inf = bmv(sy, wt, m, col, pv, vv);
            double dot = 0.0;
            for (int i = 0; i < 2 * col; ++i) dot += vv[i] * pv[i];
            sc[0] = dot; sc[1] = (double)inf;
          }
          group_sync<G>();
          info = (int)sc[1];
          f2 -= sc[0];
          group_sync<G>();
        }
        SE3MPC_TICK(10)
        if (info == 0) {
          double dtm = -f1 / f2, tsum = 0.0, tj = 0.0;
          int nleft = nbreak;
          bool all_fixed = false;
          // The search below stops before its first crossing when dtm < min_j t_j.  num * (1 - 2^-50) > dtm * |g| (both products rounded) implies
          // num / |g| > dtm * (1 + 2^-51), hence fl(num / |g|) > dtm: if that holds in every lane the loop would do nothing but form the nine
          // quotients per lane and their group minimum -- skipped.  Otherwise (a crossing, or too close to call) the published search runs.
          // (accumulated with `&`, not a short-circuit `&&` chain: straight-line compares instead of nine nested divergent regions)
          bool clear = dtm > 0.0;
#pragma unroll
          for (int j = 0; j < J; ++j) clear = clear & (tbp[j] * 0.99999999999999911182158029987 > dtm * fabs(g[j]));
          const bool skip_search = group_ballot<G>(!clear) == 0ull;
          if (!skip_search) breakpoints();
          while (!skip_search && nleft > 0) {
            // next smallest breakpoint: per-lane min, group min, owner = first lane holding it
            double tmin = tbp[0];
            int jm = 0;
#pragma unroll
            for (int j = 1; j < J; ++j) if (tbp[j] < tmin) { tmin = tbp[j]; jm = j; }
            const double tj0 = tj;
            tj = group_min<G>(tmin);
            const int src = first_lane(group_ballot<G>(tmin == tj));
            if (src < 0) { info = 1; break; }                 // NaN breakpoints: give up on this memory
            const int jsel = group_bcast<G>(jm, src);
            const double dt = tj - tj0;
            if (dtm < dt) break;
            tsum += dt;
            --nleft;
            SE3MPC_COUNT(15)
            // owner fixes its variable at the bound it hits
            double dib = 0.0, zib = 0.0;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              if (j == jsel && k == src) {
                const double lo = box_lo(q, j), hi = box_hi(q, j);
                dib = d[j]; d[j] = 0.0; tbp[j] = kInf;
                if (dib > 0.0) { zib = hi - x[j]; z[j] = hi; iwhere[j] = 2; }
                else { zib = lo - x[j]; z[j] = lo; iwhere[j] = 1; }
              }
            }
            const double dibp = group_bcast<G>(dib, src), zibp = group_bcast<G>(zib, src);
            if (nleft == 0 && nbreak == n) { dtm = dt; all_fixed = true; break; }
            const double dibp2 = dibp * dibp;
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
            f2 = f2 - theta * dibp2;
            const int ibp = (lane - k + src) * LS + 2 * jsel;      // the owner's (s, y) couple inside a pair's [64][LS] image
            // (the middle-matrix product of a crossing runs as a scalar section of the group's first lane for every col: a crossing with
            // pairs in memory is rare -- none in a typical solve -- and its register form was one of the kernel's two pressure peaks)
            if (col > 0) {
              if (k == 0) {
                for (int i = 0; i < 2 * col; ++i) cv[i] += dt * pv[i];
                for (int c = 0; c < col; ++c) { wbp[c] = (double)pairs[c * kWave * LS + ibp + 1]; wbp[col + c] = theta * (double)pairs[c * kWave * LS + ibp]; }
                const int inf = bmv(sy, wt, m, col, wbp, vv);
                double wmc = 0.0, wmp = 0.0, wmw = 0.0;
                for (int i = 0; i < 2 * col; ++i) { wmc += cv[i] * vv[i]; wmp += pv[i] * vv[i]; wmw += wbp[i] * vv[i]; }
                for (int i = 0; i < 2 * col; ++i) pv[i] -= dibp * wbp[i];
                sc[0] = wmc; sc[1] = wmp; sc[2] = wmw; sc[3] = (double)inf;
              }
              group_sync<G>();
              info = (int)sc[3];
              f1 += dibp * sc[0];
              f2 += 2.0 * dibp * sc[1] - dibp2 * sc[2];
              group_sync<G>();
              if (info != 0) break;
            }
            f2 = fmax(kEps * f2_org, f2);
            if (nleft > 0) dtm = -f1 / f2;
            else { f1 = 0.0; f2 = 0.0; dtm = 0.0; }       // every variable with d != 0 has hit a bound
          }
          SE3MPC_TICK(11)
          if (info == 0) {
            if (!all_fixed) {
              if (dtm <= 0.0) dtm = 0.0;
              tsum += dtm;
#pragma unroll
              for (int j = 0; j < J; ++j) z[j] += tsum * d[j];
            }
            if (col > 0) {
              if (k == 0) for (int i = 0; i < 2 * col; ++i) cv[i] += dtm * pv[i];
              group_sync<G>();
            }
          }
        }
      }
      }   // col > 0: sequential search
    }
    if (info != 0) { col = 0; theta = 1.0; iupdat = 0; continue; }   // singular middle matrix: refresh memory

    SE3MPC_TICK(2)
    // ===================================================================== subspace minimization
    int nfree_p = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) nfree_p += (iwhere[j] <= 0) ? 1 : 0;
    const int nfree = group_sum_i32<G>(nfree_p);
    if (nfree != 0 && col != 0) {
      // wvr: the subspace solution (K^-1 W'Z r) of the fast path, in registers of every lane; the LDS path leaves it in wv[]
      double wvr[2 * kFastCol] = {0.0, 0.0, 0.0, 0.0};
      if (col <= kFastCol) {
        auto subspace_fast = [&](auto tag) {
          constexpr int C = decltype(tag)::value;
          // ---- formk: the same inner products, accumulated into a register matrix (upper triangle of the 2C x 2C K matrix)
          double wnr[2 * C][2 * C];
#pragma unroll
          for (int i = 0; i < 2 * C; ++i)
#pragma unroll
            for (int kk = 0; kk < 2 * C; ++kk) wnr[i][kk] = 0.0;
          // all C*C cells' partial sums first, then ONE interleaved reduction of the 3 sums each cell needs (yzzy, saas for the
          // lower-left half, and the cross term: sa_y below the diagonal, sz_y on and above it)
          double sums[3 * C * C];
#pragma unroll
          for (int i = 0; i < 3 * C * C; ++i) sums[i] = 0.0;
          // (the rows of all C pairs are needed at once here: 18 C registers in the float kernel; the double kernel with two pairs would
          // need 72 and reads couple by couple instead -- 16 bytes per read there too)
          constexpr bool kRows = sizeof(IO) == 4 || C == 1;
          IO wr_[kRows ? C : 1][2 * J];
          if constexpr (kRows) {
#pragma unroll
            for (int c = 0; c < C; ++c) load_row(ROW(c), wr_[c]);
          }
#pragma unroll
          for (int j = 0; j < J; ++j) {                        // slot-outer: a slot's 2C values of S, Y are folded into every cell
            const bool fr = iwhere[j] <= 0;
            double wyv[C], wsv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
              if constexpr (kRows) { wyv[c] = (double)wr_[c][2 * j + 1]; wsv[c] = (double)wr_[c][2 * j]; }
              else { wyv[c] = (double)WY(c, j); wsv[c] = (double)WS(c, j); }
            }
#pragma unroll
            for (int iy = 0; iy < C; ++iy) {
#pragma unroll
              for (int jy = 0; jy < C; ++jy) {
                const double yy = wyv[iy] * wyv[jy], sS = wsv[iy] * wsv[jy], sY = wsv[iy] * wyv[jy];
                sums[3 * (iy * C + jy) + 0] += fr ? yy : 0.0;                                  // yzzy: free set
                sums[3 * (iy * C + jy) + 1] += fr ? 0.0 : sS;                                  // saas: active set
                sums[3 * (iy * C + jy) + 2] += ((jy < iy) ? !fr : fr) ? sY : 0.0;              // sa_y below the diagonal, sz_y on and above it
              }
            }
          }
          group_sum_n<G, 3 * C * C>(sums);
#pragma unroll
          for (int iy = 0; iy < C; ++iy) {
#pragma unroll
            for (int jy = 0; jy < C; ++jy) {
              const double yzzy = sums[3 * (iy * C + jy) + 0], saas = sums[3 * (iy * C + jy) + 1], cross = sums[3 * (iy * C + jy) + 2];
              if (jy <= iy) {
                wnr[jy][iy] = yzzy / theta + (jy == iy ? sy[iy * m + iy] : 0.0);
                wnr[C + jy][C + iy] = saas * theta;
              }
              wnr[jy][C + iy] = (jy < iy) ? -cross : cross;
            }
          }
          reload_lds();
          int inf = formk_factor_regs<C>(wnr);
          SE3MPC_TICK(12)
          // ---- cmprlb, scalar part: mc = M c
          MidRegs<C> M;
          load_mid<C>(sy, wt, m, M);
          double cr[2 * C], mc[2 * C];
#pragma unroll
          for (int i = 0; i < 2 * C; ++i) cr[i] = cv[i];
          if (inf == 0 && bmv_regs<C>(M, cr, mc)) inf = -8;
          info = inf;
          if (info != 0) return;
          // ---- cmprlb: r = -Z'(B(xcp - x) + g)   (held in d[] on the free variables)
#pragma unroll
          for (int j = 0; j < J; ++j) d[j] = -theta * (z[j] - x[j]) - g[j];     // every variable's value first, ONE select at the end (d stays 0 off the free set)
#pragma unroll
          for (int c = 0; c < C; ++c) {                        // pair-outer: one row in registers at a time; per element the additions keep their order
            IO w[2 * J];
            load_row(ROW(c), w);
#pragma unroll
            for (int j = 0; j < J; ++j) d[j] += (double)w[2 * j + 1] * mc[c] + (double)w[2 * j] * (theta * mc[C + c]);
          }
#pragma unroll
          for (int j = 0; j < J; ++j) d[j] = (iwhere[j] <= 0) ? d[j] : 0.0;
          reload_lds();
          // ---- subsm: wv = W'Z d ; wv = K^-1 wv   (d is 0 off the free set: no condition inside the sums)
          double wr[2 * C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            double a1 = 0.0, a2 = 0.0;
            IO w[2 * J];
            load_row(ROW(c), w);
#pragma unroll
            for (int j = 0; j < J; ++j) { a1 += (double)w[2 * j + 1] * d[j]; a2 += (double)w[2 * j] * d[j]; }
            wr[c] = a1; wr[C + c] = a2;
          }
          group_sum_n<G, 2 * C>(wr);
#pragma unroll
          for (int c = 0; c < C; ++c) wr[C + c] = theta * wr[C + c];
          reload_lds();
          int inf2 = dtrsl_regs<2 * C>(wnr, wr, true);
          if (!inf2) {
#pragma unroll
            for (int i = 0; i < C; ++i) wr[i] = -wr[i];
            inf2 = dtrsl_regs<2 * C>(wnr, wr, false);
          }
          info = inf2;
#pragma unroll
          for (int i = 0; i < C; ++i) { wvr[i] = wr[i]; wvr[kFastCol + i] = wr[C + i]; }
        };
        if (col == 1) subspace_fast(ColTag<1>{}); else subspace_fast(ColTag<2>{});
      } else {
        // ---- formk: K blocks from inner products over the free (Z) and active (A) sets
        for (int iy = 0; iy < col; ++iy) {
          for (int jy = 0; jy < col; ++jy) {
            double yzzy = 0.0, saas = 0.0, sa_y = 0.0, sz_y = 0.0;
  #pragma unroll
            for (int j = 0; j < J; ++j) {
              const bool fr = iwhere[j] <= 0;
              const double wyi = (double)WY(iy, j), wyj = (double)WY(jy, j), wsi = (double)WS(iy, j), wsj = (double)WS(jy, j);
              if (fr) { yzzy += wyi * wyj; sz_y += wsi * wyj; }
              else { saas += wsi * wsj; sa_y += wsi * wyj; }     // active set (padding rows of S, Y are zero)
            }
            // wn (upper triangle, row stride 2m):  [ D + Y'ZZ'Y/theta   -L_a' + R_z' ;  .   theta S'AA'S ]
            if (jy <= iy) {
              yzzy = group_sum<G>(yzzy); saas = group_sum<G>(saas);
              if (k == 0) {
                wn[jy * 2 * m + iy] = yzzy / theta + (jy == iy ? sy[iy * m + iy] : 0.0);
                wn[(col + jy) * 2 * m + (col + iy)] = saas * theta;
              }
            }
            if (jy < iy) {
              sa_y = group_sum<G>(sa_y);
              if (k == 0) wn[jy * 2 * m + (col + iy)] = -sa_y;
            } else {
              sz_y = group_sum<G>(sz_y);
              if (k == 0) wn[jy * 2 * m + (col + iy)] = sz_y;
            }
          }
        }
        group_sync<G>();
        if (k == 0) {
          const int ld = 2 * m;
          int inf = dpofa(wn, ld, col);
          if (inf) inf = -1;
          else {
            for (int js = col; js < 2 * col; ++js) dtrsl_upper(wn, ld, col, wn + js, ld, true);
            for (int is = col; is < 2 * col; ++is)
              for (int js = is; js < 2 * col; ++js) {
                double s = 0.0;
                for (int kk = 0; kk < col; ++kk) s += wn[kk * ld + is] * wn[kk * ld + js];
                wn[is * ld + js] += s;
              }
            if (dpofa(wn + col * ld + col, ld, col)) inf = -2;
          }
          // ---- cmprlb, scalar part: mc = M c
          if (inf == 0 && bmv(sy, wt, m, col, cv, vv)) inf = -8;
          sc[0] = (double)inf;
        }
        group_sync<G>();
        info = (int)sc[0];
        if (info == 0) {
          // ---- cmprlb: r = -Z'(B(xcp - x) + g)   (held in d[] on the free variables)
  #pragma unroll
          for (int j = 0; j < J; ++j) d[j] = (iwhere[j] <= 0) ? (-theta * (z[j] - x[j]) - g[j]) : 0.0;
          for (int c = 0; c < col; ++c) {
            const double a1 = vv[c], a2 = theta * vv[col + c];
  #pragma unroll
            for (int j = 0; j < J; ++j) if (iwhere[j] <= 0) d[j] += (double)WY(c, j) * a1 + (double)WS(c, j) * a2;
          }
          // ---- subsm: wv = W'Z d ; wv = K^-1 wv ; d = (d + Z'W wv-ish)/theta
          for (int c = 0; c < col; ++c) {
            double a1 = 0.0, a2 = 0.0;
  #pragma unroll
            for (int j = 0; j < J; ++j) if (iwhere[j] <= 0) { a1 += (double)WY(c, j) * d[j]; a2 += (double)WS(c, j) * d[j]; }
            double r2[2] = {a1, a2};
            group_sum_n<G, 2>(r2);
            a1 = r2[0]; a2 = r2[1];
            if (k == 0) { wv[c] = a1; wv[col + c] = theta * a2; }
          }
          group_sync<G>();
          if (k == 0) {
            int inf = dtrsl_upper(wn, 2 * m, 2 * col, wv, 1, true);
            if (!inf) {
              for (int i = 0; i < col; ++i) wv[i] = -wv[i];
              inf = dtrsl_upper(wn, 2 * m, 2 * col, wv, 1, false);
            }
            sc[0] = (double)inf;
          }
          group_sync<G>();
          info = (int)sc[0];
        }
      }   // col > kFastCol: LDS path
      {
        if (info == 0) {
          if (col <= kFastCol) {
#pragma unroll
            for (int c = 0; c < kFastCol; ++c) {
              if (c < col) {
                const double b1 = wvr[c] / theta, b2 = wvr[kFastCol + c];
                IO w[2 * J];
                load_row(ROW(c), w);
#pragma unroll
                for (int j = 0; j < J; ++j) {
                  const double t = d[j] + ((double)w[2 * j + 1] * b1 + (double)w[2 * j] * b2);
                  d[j] = (iwhere[j] <= 0) ? t : d[j];
                }
              }
            }
          } else {
            for (int c = 0; c < col; ++c) {
              const double b1 = wv[c] / theta, b2 = wv[col + c];
#pragma unroll
              for (int j = 0; j < J; ++j) if (iwhere[j] <= 0) d[j] += (double)WY(c, j) * b1 + (double)WS(c, j) * b2;
            }
          }
          const double rth = 1.0 / theta;
          // projected Newton point; xo is free here (the line search re-saves it): xo keeps xcp
          bool hitp = false;
          double ddp = 0.0;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            xo[j] = z[j];
            const bool fr = iwhere[j] <= 0;
            const double lo = box_lo(q, j), hi = box_hi(q, j);
            const double dn = d[j] * rth;
            const double xk = fmin(hi, fmax(lo, z[j] + dn));
            d[j] = fr ? dn : d[j];
            z[j] = fr ? xk : z[j];
            hitp = hitp | (fr & ((xk == lo) | (xk == hi)));
            ddp += (z[j] - x[j]) * g[j];
          }
          const bool iword = group_ballot<G>(hitp) != 0ull;
          if (iword) {
            const double dd_p = group_sum<G>(ddp);
            if (dd_p > 0.0) {
              // not a descent direction: back to xcp and truncate the Newton step at the first bound
              double amin = 1.0;
              unsigned imin = 0xFFFFFFFFu;
#pragma unroll
              for (int j = 0; j < J; ++j) {
                z[j] = xo[j];
                if (iwhere[j] <= 0) {
                  const double lo = box_lo(q, j), hi = box_hi(q, j);
                  const double dk = d[j];
                  double cand = 1.0;
                  if (dk < 0.0) { const double t2 = lo - z[j]; cand = (t2 >= 0.0) ? 0.0 : ((dk * 1.0 < t2) ? t2 / dk : 1.0); }
                  else if (dk > 0.0) { const double t2 = hi - z[j]; cand = (t2 <= 0.0) ? 0.0 : ((dk * 1.0 > t2) ? t2 / dk : 1.0); }
                  // ties go to the variable that comes first in the reference's packing [P | V | T], row = 3k + axis (subsm's strict `<` in variable order):
                  // inside a lane the slots are visited in that order, across lanes the smallest index wins below
                  const unsigned idx = (unsigned)((j / 3) * n3 + 3 * k + (j % 3));
                  if (cand < amin) { amin = cand; imin = idx; }
                }
              }
              const double alpha = group_min<G>(amin);
              const unsigned ibd = group_min_u32<G>(amin == alpha && alpha < 1.0 ? imin : 0xFFFFFFFFu);
#pragma unroll
              for (int j = 0; j < J; ++j) {
                if (iwhere[j] <= 0) {
                  const unsigned idx = (unsigned)((j / 3) * n3 + 3 * k + (j % 3));
                  if (alpha < 1.0 && idx == ibd) {
                    if (d[j] > 0.0) { z[j] = box_hi(q, j); d[j] = 0.0; }
                    else if (d[j] < 0.0) { z[j] = box_lo(q, j); d[j] = 0.0; }
                  }
                  z[j] += alpha * d[j];
                }
              }
            }
          }
        }
      }
      if (info != 0) { col = 0; theta = 1.0; iupdat = 0; continue; }
    }

    SE3MPC_TICK(3)
    // ===================================================================== line search (lnsrlb)
    double dtdp = 0.0, gd0p = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) { d[j] = z[j] - x[j]; dtdp += d[j] * d[j]; gd0p += g[j] * d[j]; }
    double r2[2] = {dtdp, gd0p};
    group_sum_n<G, 2>(r2);
    const double dtd = r2[0];
    gd_fused = r2[1];                                         // g'd at the start of the search
    double stpmx;
    if (iter == 0) stpmx = 1.0;
    else {
      double smx = kBig;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double lo = box_lo(q, j), hi = box_hi(q, j);
        const double a1 = d[j];
        const bool neg = a1 < 0.0, pos = a1 > 0.0;
        const double a2 = neg ? lo - x[j] : hi - x[j];
        const bool blocked = neg ? a2 >= 0.0 : (pos & (a2 <= 0.0));
        const bool tighter = neg ? a1 * smx < a2 : (pos & (a1 * smx > a2));
        const double quot = a2 / a1;
        smx = blocked ? 0.0 : (tighter ? quot : smx);
      }
      stpmx = group_min<G>(smx);
    }
    double stp = 1.0;
#pragma unroll
    for (int j = 0; j < J; ++j) xo[j] = opaque(x[j]);
    SE3MPC_TICK(13)
    fold = f;
    int ifun = 0, iback = 0, ls_info = 0;
    double gd = 0.0, gdold = 0.0;
    LineSearch ls;
    bool start = true;
    while (true) {
      gd = gd_fused;                                          // reduced together with dtd (first pass) or with f (after an evaluation)
      if (ifun == 0) {
        gdold = gd;
        if (gd >= 0.0) { ls_info = -4; break; }
      }
      const int lt = dcsrch(f, gd, stp, 0.0, stpmx, start, ls);
      start = false;
      if (lt == LS_CONV || lt == LS_WARN) break;
      if (lt == LS_ERROR) { ls_info = -4; break; }          // dcsrch rejected its inputs (never with a feasible d)
      ++ifun; iback = ifun - 1;
      if (iback >= q.maxls) break;
      // the trial point; `moved`: it differs from the x of the last evaluation (which the registers still hold)
      bool moved = false;
      if (stp == 1.0) {
#pragma unroll
        for (int j = 0; j < J; ++j) { moved = moved | !(z[j] == x[j]); x[j] = z[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < J; ++j) { const double xn = stp * (z[j] - xo[j]) + xo[j]; moved = moved | !(xn == x[j]); x[j] = xn; }
      }
      if (!x_is_last || group_ballot<G>(moved) != 0ull) ++nfev;
      x_is_last = true;
      SE3MPC_TICK(4)
      f = eval_fg(true);
      SE3MPC_TICK(1)
    }
    SE3MPC_TICK(4)
    if (ls_info != 0 || iback >= q.maxls) {
      // restore the previous iterate (its gradient recomputed).  x held the x of the last evaluation, or the iterate itself when
      // this search evaluated nothing; after the restore it still does only if the two are the same point.
      bool differs = false;
#pragma unroll
      for (int j = 0; j < J; ++j) { differs = differs | !(x[j] == xo[j]); x[j] = xo[j]; g[j] = grad_of(j, xo[j]); }     // (g was not touched by the search: this is what it holds already, bit for bit)
      if (group_ballot<G>(differs) != 0ull) x_is_last = false;
      f = fold;
      if (col == 0) { task = SE3MPC_TASK_ABNORMAL; status = 2; break; }
      col = 0; theta = 1.0; iupdat = 0;
      continue;
    }

    // ===================================================================== new iterate
    ++iter; ++nit;
#pragma unroll
    for (int j = 0; j < J; ++j) g[j] = grad_of(j, x[j]);      // the gradient at the accepted point (the search's evaluations did not store it)
    sbgnrm = projgr();
    if (nit >= q.maxiter) { task = SE3MPC_TASK_STOP_MAXITER; status = 1; break; }
    if (nfev > q.maxfun) { task = SE3MPC_TASK_STOP_MAXFUN; status = 1; break; }
    if (sbgnrm <= q.pgtol) { task = SE3MPC_TASK_CONV_PGTOL; status = 0; break; }
    {
      const double ddum = fmax(fabs(fold), fmax(fabs(f), 1.0));
      if ((fold - f) <= q.ftol * ddum) { task = SE3MPC_TASK_CONV_FTOL; status = 0; break; }
    }
    // ---- BFGS update (matupd + formt); d is formed again from x_old, then xo becomes y = g - g(x_old)
    double rrp = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) { d[j] = z[j] - xo[j]; xo[j] = g[j] - grad_of(j, xo[j]); rrp += xo[j] * xo[j]; }
    const double rr = group_sum<G>(rrp);
    double dr, ddum;
    if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
    else {
      dr = (gd - gdold) * stp; ddum = -gdold * stp;
#pragma unroll
      for (int j = 0; j < J; ++j) d[j] *= stp;
    }
    if (dr <= kEps * ddum) continue;                    // skip the update
    ++iupdat;
    if (iupdat > m && m < q.m) { task = SE3MPC_TASK_OVERFLOW; status = 1; break; }   // second tier re-solves this problem
    if (iupdat <= m) col = iupdat;
    else {
      // memory full: drop the oldest pair (each lane shifts its own elements; the first lane the small matrices)
      for (int c = 0; c + 1 < m; ++c) {
#pragma unroll
        for (int j = 0; j < J; ++j) { WS(c, j) = WS(c + 1, j); WY(c, j) = WY(c + 1, j); }
      }
      if (k == 0)
        for (int i = 0; i + 1 < m; ++i)
          for (int kk = 0; kk + 1 < m; ++kk) { ss[i * m + kk] = ss[(i + 1) * m + kk + 1]; sy[i * m + kk] = sy[(i + 1) * m + kk + 1]; }
      group_sync<G>();
    }
    { IO w[2 * J];
#pragma unroll
      for (int j = 0; j < J; ++j) { w[2 * j] = (IO)d[j]; w[2 * j + 1] = (IO)xo[j]; }
      store_row(ROW(col - 1), w); }
    theta = rr / dr;
    bool formt_failed;
    if (col <= kFastCol) {
      // matupd's new row / column and formt on registers: every lane forms theta*S'S + L D^-1 L' and its Cholesky factor from
      // group-uniform values; the first lane writes the state (sy, ss, wt) back for the next iteration
      auto update_fast = [&](auto tag) {
        constexpr int C = decltype(tag)::value;
        double syr[C][C], ssr[C][C], wtr[C][C];
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
          for (int kk = 0; kk < C; ++kk) { syr[i][kk] = (kk <= i && i < C - 1) ? sy[i * m + kk] : 0.0; ssr[i][kk] = (kk >= i && kk < C - 1) ? ss[i * m + kk] : 0.0; wtr[i][kk] = 0.0; }
#pragma unroll
        for (int c = 0; c + 1 < C; ++c) {
          double a1 = 0.0, a2 = 0.0;
          IO w[2 * J];
          load_row(ROW(c), w);
#pragma unroll
          for (int j = 0; j < J; ++j) { a1 += d[j] * (double)w[2 * j + 1]; a2 += (double)w[2 * j] * d[j]; }
          double r2[2] = {a1, a2};
          group_sum_n<G, 2>(r2);
          syr[C - 1][c] = r2[0]; ssr[c][C - 1] = r2[1];
        }
        ssr[C - 1][C - 1] = (stp == 1.0) ? dtd : stp * stp * dtd;
        syr[C - 1][C - 1] = dr;
        // formt: T = theta*S'S + L D^-1 L', Cholesky factor in wt
#pragma unroll
        for (int jj = 0; jj < C; ++jj) wtr[0][jj] = theta * ssr[0][jj];
#pragma unroll
        for (int i = 1; i < C; ++i)
#pragma unroll
          for (int jj = i; jj < C; ++jj) {
            const int k1 = i < jj ? i : jj;
            double dd = 0.0;
#pragma unroll
            for (int kk = 0; kk < C; ++kk) if (kk < k1) dd += syr[i][kk] * syr[jj][kk] / syr[kk][kk];
            wtr[i][jj] = dd + theta * ssr[i][jj];
          }
        formt_failed = dpofa_regs<C, 0, C>(wtr) != 0;
        group_sync<G>();
        if (k == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) { sy[(C - 1) * m + c] = syr[C - 1][c]; ss[c * m + (C - 1)] = ssr[c][C - 1]; }
#pragma unroll
          for (int i = 0; i < C; ++i)
#pragma unroll
            for (int jj = 0; jj < C; ++jj) if (jj >= i) wt[i * m + jj] = wtr[i][jj];
        }
        group_sync<G>();
      };
      if (col == 1) update_fast(ColTag<1>{}); else update_fast(ColTag<2>{});
    } else {
      for (int c = 0; c + 1 < col; ++c) {
        double a1 = 0.0, a2 = 0.0;
  #pragma unroll
        for (int j = 0; j < J; ++j) { a1 += d[j] * (double)WY(c, j); a2 += (double)WS(c, j) * d[j]; }
        a1 = group_sum<G>(a1); a2 = group_sum<G>(a2);
        if (k == 0) { sy[(col - 1) * m + c] = a1; ss[c * m + (col - 1)] = a2; }
      }
      if (k == 0) {
        ss[(col - 1) * m + (col - 1)] = (stp == 1.0) ? dtd : stp * stp * dtd;
        sy[(col - 1) * m + (col - 1)] = dr;
        // formt: T = theta*S'S + L D^-1 L', Cholesky factor in wt
        for (int jj = 0; jj < col; ++jj) wt[jj] = theta * ss[jj];
        for (int i = 1; i < col; ++i)
          for (int jj = i; jj < col; ++jj) {
            const int k1 = i < jj ? i : jj;
            double dd = 0.0;
            for (int kk = 0; kk < k1; ++kk) dd += sy[i * m + kk] * sy[jj * m + kk] / sy[kk * m + kk];
            wt[i * m + jj] = dd + theta * ss[i * m + jj];
          }
        sc[0] = (double)dpofa(wt, m, col);
      }
      group_sync<G>();
      formt_failed = sc[0] != 0.0;
      group_sync<G>();
    }
    if (formt_failed) { col = 0; theta = 1.0; iupdat = 0; }

  }

  SE3MPC_TICK(5)
  // ===================================================================== results
  if (live) {
#pragma unroll
    for (int j = 0; j < J; ++j) Xg[(size_t)pb * n + (j / 3) * n3 + 3 * k + (j % 3)] = (IO)x[j];
  }
  if (k == 0 && infog != nullptr) {
    se3mpc_solve_info r;
    r.fun = f; r.nit = nit; r.nfev = nfev; r.status = status; r.task = task;
    infog[pb] = r;
  }
  // a restart solve asks for x and info only (every trajectory output null): nothing to extract; a closed loop that reads the plan in
  // place wants the accelerations but no attitudes / rates / thrust magnitudes: no frames to build
  const bool want_frames = attg != nullptr || ratesg != nullptr || thrustg != nullptr;
  // ---- _extract_solution_from_result (planner.py:582-654): lane k = step k holds T_k in its own registers
  const double t0 = x[6], t1 = x[7], t2 = x[8];
  if (live && accg != nullptr) {
    const size_t o = (size_t)pb * n3 + 3 * k;
    accg[o] = (IO)(t0 / q.mass); accg[o + 1] = (IO)(t1 / q.mass); accg[o + 2] = (IO)(t2 / q.mass - q.grav);
  }
  if (want_frames) {
  const double mag = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
  const bool valid = live && mag > 1e-6;
  double b1[3] = {0, 0, 0}, b2[3] = {0, 0, 0}, b3[3] = {0, 0, 0};
  double roll = 0.0, pitch = 0.0, yaw = 0.0;
  if (valid) {
    b3[0] = t0 / mag; b3[1] = t1 / mag; b3[2] = t2 / mag;
    b1[0] = 0.0; b1[1] = -b3[2]; b1[2] = b3[1];
    const double n1 = sqrt(b1[1] * b1[1] + b1[2] * b1[2]);
    if (n1 > 1e-6) { b1[1] /= n1; b1[2] /= n1; } else { b1[0] = 1.0; b1[1] = 0.0; b1[2] = 0.0; }
    b2[0] = b3[1] * b1[2] - b3[2] * b1[1];
    b2[1] = b3[2] * b1[0] - b3[0] * b1[2];
    b2[2] = b3[0] * b1[1] - b3[1] * b1[0];
    roll = atan2(b2[2], b3[2]);
    pitch = asin(fmin(fmax(-b1[2], -1.0), 1.0));
    yaw = n1 > 1e-6 ? (b1[1] == 0.0 ? b1[1] : copysign(1.5707963267948966, b1[1])) : 0.0;   // atan2(b1y, b1x) with b1x exactly 0, or b1 = (1,0,0)
  }
  // prev_R of step k = R of the nearest earlier step with |T| > 1e-6 (planner.py:641-650): fetched from that step's lane
  const uint64_t vmask = group_ballot<G>(valid);
  const uint64_t below = vmask & ((k == 0) ? 0ull : (~0ull >> (64 - k)));
  const bool has_prev = valid && below != 0ull;
  const int pk = has_prev ? 63 - __builtin_clzll(below) : k;          // lane of the group that holds the previous frame (itself: unused)
  double q1[3], q2[3], q3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { q1[c] = group_gather<G>(b1[c], pk); q2[c] = group_gather<G>(b2[c], pk); q3[c] = group_gather<G>(b3[c], pk); }
  double w0 = 0.0, w1 = 0.0, w2 = 0.0;
  if (has_prev) {
    double d1[3], d2[3], d3[3];
    for (int c = 0; c < 3; ++c) {
      d1[c] = (b1[c] - q1[c]) / q.dt;
      d2[c] = (b2[c] - q2[c]) / q.dt;
      d3[c] = (b3[c] - q3[c]) / q.dt;
    }
    w0 = b3[0] * d2[0] + b3[1] * d2[1] + b3[2] * d2[2];
    w1 = b1[0] * d3[0] + b1[1] * d3[1] + b1[2] * d3[2];
    w2 = b2[0] * d1[0] + b2[1] * d1[1] + b2[2] * d1[2];
  }
  if (live) {
    const size_t o = (size_t)pb * n3 + 3 * k;
    if (attg != nullptr) { attg[o] = (IO)roll; attg[o + 1] = (IO)pitch; attg[o + 2] = (IO)yaw; }
    if (ratesg != nullptr) { ratesg[o] = (IO)w0; ratesg[o + 1] = (IO)w1; ratesg[o + 2] = (IO)w2; }
    if (thrustg != nullptr) thrustg[(size_t)pb * N + k] = (IO)mag;
  }
  }   // want_frames
  // se3mpc_plan_host_*: a ONE-wavefront launch tells the waiting host it is done -- every store above made visible to the system, then
  // the ticket (the first lane of the wavefront belongs to problem 0, which exists; groups of a wavefront reach this point together)
  if (doneg != nullptr) {
    __threadfence_system();
    if (lane == 0) *reinterpret_cast<volatile unsigned long long*>(doneg) = ticket;
  }
#ifdef SE3MPC_SOLVE_PROFILE
  __syncthreads();
  SE3MPC_TICK(6)
  if (lane == 0) tsec[7] = tlast - tstart;
  __syncthreads();
  if (lane == 0 && attg != nullptr && (size_t)3 * N * sizeof(IO) >= 128) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(attg + (size_t)pb * 3 * N);
    for (int i = 0; i < 16; ++i) dst[i] = tsec[i];
  }
#endif
}

static int g_solver_variant = 0;   // bit 0: published sequential Cauchy search also while the memory is empty; bits 8-15: forced group size (0 = automatic)

static size_t solve_lds_bytes(int m, int G, size_t io_size) {
  const size_t pairs = (size_t)m * kWave * (io_size == 4 ? pair_row_values<float>() : pair_row_values<double>()) * io_size;
  return pairs_offset_bytes(kWave / G, m) + pairs;
}

// Group size for problems of horizon N: the smallest group that holds the horizon packs the most problems into a wavefront -- the
// group-uniform algebra (line search, middle matrices, reductions) is then shared by 64 / G problems.  Measured on MI355X
// (profiles/r03_solve_group_probe.txt): the smallest group wins at every batch size from 1 K problems up, also where a wider group
// would put more wavefronts on the chip (N = 6: 1 K / 4 K / 8 K / 64 K problems take 32 / 37 / 48 / 180 us at G = 8 against
// 33 / 48 / 51 / 335 us at G = 16); the cost of co-resident problems diverging (a problem that stops after one iteration waits for
// its neighbours' third) is about an eighth of a wavefront's time.
static int solve_group_size(int N) {
  const int forced = (g_solver_variant >> 8) & 0xFF;
  const int G = N <= 8 ? 8 : (N <= 16 ? 16 : (N <= 32 ? 32 : 64));
  if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced >= G ? forced : G;
  return G;
}

// done != nullptr: the host-latency form (se3mpc_plan_host_*): the whole batch must fit ONE wavefront, the kernel stores `ticket`
// into *done (host-visible memory) as its last act.
template <typename IO>
int solve_impl(const se3mpc_params* p, int B, const IO* p0, const IO* v0, const IO* goal, const IO* x0, IO* X,
               se3mpc_solve_info* info, IO* acc, IO* att, IO* rates, IO* thrust, void* stream,
               unsigned long long* done = nullptr, unsigned long long ticket = 0) {
  if (p == nullptr) return SE3MPC_ERR_NULL;
  int rc = check_params_impl(p);
  if (rc) return rc;
  if (B < 0) return SE3MPC_ERR_SHAPE;
  if (B == 0) return SE3MPC_OK;
  if (!p0 || !v0 || !X || (p->has_goal && !goal)) return SE3MPC_ERR_NULL;
  SolveDev q = make_solve_dev(*p);
  q.seq_cauchy = g_solver_variant & 1;
  if (q.N > kWave) return SE3MPC_ERR_SHAPE;               // horizon <= 64 (check_params_impl says the same)
  hipStream_t s = (hipStream_t)stream;
  // (one problem alone in its wavefront runs the whole-wavefront form: its group-uniform values are SGPRs, its branches scalar)
  const int G = (B == 1 && ((g_solver_variant >> 8) & 0xFF) == 0) ? kWave : solve_group_size(q.N);
  const int waves = (int)(((long)B * G + kWave - 1) / kWave);
  if (done != nullptr && waves != 1) return SE3MPC_ERR_SHAPE;
#define SE3MPC_SOLVE_CASE(GG)                                                                                               \
  if (solve_lds_bytes(q.mlds, GG, sizeof(IO)) > 64 * 1024)                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&solve_kernel<IO, GG>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)solve_lds_bytes(q.mlds, GG, sizeof(IO)));                                                 \
  hipLaunchKernelGGL((solve_kernel<IO, GG>), dim3(waves), dim3(kWave), solve_lds_bytes(q.mlds, GG, sizeof(IO)), s, q, B, p0, \
                     v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket)
#define SE3MPC_SOLVE_LAUNCH()               \
  if (G == 8) { SE3MPC_SOLVE_CASE(8); }         \
  else if (G == 16) { SE3MPC_SOLVE_CASE(16); }  \
  else if (G == 32) { SE3MPC_SOLVE_CASE(32); }  \
  else { SE3MPC_SOLVE_CASE(64); }
  // Two tiers buy occupancy (two resident wavefronts per SIMD): the first gets the most pairs (<= 4, >= 2) whose LDS still lets
  // eight wavefronts share a CU's 160 KiB.  While every wavefront of the launch is resident at once even with the full memory's LDS
  // footprint (256 CUs x the wavefronts whose LDS fits a CU, at most one per SIMD) there is nothing to buy, and a single launch
  // with the full memory is quicker.
  constexpr size_t kLdsPerCu = 160 * 1024;
  int fast = 4;
  while (fast > 2 && solve_lds_bytes(fast, G, sizeof(IO)) > kLdsPerCu / 8) --fast;
  size_t full_per_cu = kLdsPerCu / solve_lds_bytes(q.m, G, sizeof(IO));
  if (full_per_cu > 4) full_per_cu = 4;
  const bool two_tier = info != nullptr && q.m > fast && (size_t)waves > 256 * full_per_cu;    // the tiers talk through info[].task
  q.mlds = two_tier ? fast : q.m;
  SE3MPC_SOLVE_LAUNCH();
  rc = launch_status("se3mpc_solve");
  if (rc != SE3MPC_OK || !two_tier) return rc;
  q.mlds = q.m;
  q.only_overflow = 1;
  SE3MPC_SOLVE_LAUNCH();
#undef SE3MPC_SOLVE_LAUNCH
#undef SE3MPC_SOLVE_CASE
  return launch_status("se3mpc_solve(second tier)");
}

}  // namespace se3mpc

using namespace se3mpc;

extern "C" int se3mpc_set_solver_variant(int variant) {
  const int forced = (variant >> 8) & 0xFF;
  if (variant < 0 || (variant & ~0xFF01) != 0 || !(forced == 0 || forced == 8 || forced == 16 || forced == 32 || forced == 64)) return SE3MPC_ERR_SHAPE;
  se3mpc::g_solver_variant = variant;
  return SE3MPC_OK;
}

extern "C" int se3mpc_solve_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                                const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att,
                                float* rates, float* thrust, void* stream) {
  return solve_impl<float>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream);
}
extern "C" int se3mpc_solve_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                                const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att,
                                double* rates, double* thrust, void* stream) {
  return solve_impl<double>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream);
}

// Host-latency form of the solve for SE3MPCPlanner.plan_trajectory (planner.py:215-228: one problem, the caller blocks until the plan
// exists): ONE call launches the solve on the caller's host-pinned, device-mapped buffers and returns when the result is there.
// Instead of hipStreamSynchronize (an interrupt / signal wait of 10-20 us on top of the kernel) the kernel's last act is a
// system-scope store of `ticket` into *done, on which this function spins.
template <typename IO>
static int plan_host_impl(const se3mpc_params* p, int B, const IO* p0, const IO* v0, const IO* goal, const IO* x0, IO* X,
                          se3mpc_solve_info* info, IO* acc, IO* att, IO* rates, IO* thrust, unsigned long long* done,
                          unsigned long long ticket, double timeout_us, void* stream) {
  if (done == nullptr) return SE3MPC_ERR_NULL;
  volatile unsigned long long* flag = done;
  if (*flag == ticket) return SE3MPC_ERR_SHAPE;            // the ticket must differ from what the flag holds
  const int rc = solve_impl<IO>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, stream, done, ticket);
  if (rc != SE3MPC_OK) return rc;
  if (B == 0) return SE3MPC_OK;
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 1; *flag != ticket; ++spins) {
    if ((spins & 255u) == 0) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3 > timeout_us) {
        // (a device that does not make the store visible while the host polls, or a very long solve: the ordinary wait)
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return launch_status("se3mpc_plan_host(synchronize)");
        return *flag == ticket ? SE3MPC_OK : SE3MPC_ERR_LAUNCH;
      }
    }
  }
  return SE3MPC_OK;
}

extern "C" int se3mpc_plan_host_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                                    const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att, float* rates,
                                    float* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us,
                                    void* stream) {
  return plan_host_impl<float>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket, timeout_us, stream);
}
extern "C" int se3mpc_plan_host_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                                    const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att, double* rates,
                                    double* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us,
                                    void* stream) {
  return plan_host_impl<double>(p, B, p0, v0, goal, x0, X, info, acc, att, rates, thrust, done, ticket, timeout_us, stream);
}
