/*
 * se3mpc.h -- C ABI of libse3mpc.so, the MI355X (gfx950) SE(3) MPC inner solver.
 *
 * Drop-in boundary for the hot path of DART-Planner's SE3MPCPlanner
 * (reference: src/dart_planner/planning/se3_mpc_planner.py, "planner.py" below).  The
 * reference has no FFI of its own (it is pure Python on NumPy/SciPy); these entry points are
 * what a ctypes binding inside that file would call instead of its NumPy/SciPy arithmetic.
 * INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns an
 *     int status (SE3MPC_OK == 0, negative = error); nothing throws across the ABI.
 *   - All data pointers are DEVICE pointers (HBM) owned by the caller; the library never
 *     allocates or frees device memory and never synchronises the stream (graph-capturable).
 *     `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Two arithmetic types per entry point: `_f32` (production) and `_f64` (the tolerance
 *     check of BASELINE.json config 5).  Parameters are always doubles on the host side.
 *   - Two data layouts:
 *       "lane layout"  (evaluation kernels): row-major [row][b], b fastest, leading dimension
 *         `ld` >= B elements -- one trajectory per lane, every (row) access of a wavefront is
 *         one coalesced 256-B line.  Rows of a decision vector follow the reference packing
 *         (planner.py:361-376): P block rows 3k+a, V block rows 3N+3k+a, T block rows
 *         6N+3k+a (k = step, a = axis).
 *       "problem layout" (solver): [b][row], row fastest -- the reference's own packing of one
 *         decision vector per problem; one 64-lane wavefront owns one problem.
 *     se3mpc_transpose_* converts between the two.
 *   - B == 0 is legal everywhere and is a no-op.  1 <= horizon <= SE3MPC_MAX_HORIZON.
 */
#ifndef SE3MPC_H
#define SE3MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3MPC_ABI_VERSION 1
#define SE3MPC_MAX_HORIZON 64      /* 9*N <= 576 decision variables per problem            */
#define SE3MPC_MAX_CORRECTIONS 10  /* L-BFGS memory m (SciPy default maxcor = 10)          */
#define SE3MPC_MAX_SPHERES 256     /* obstacle table staged in LDS (4 KB)                  */

enum se3mpc_status {
  SE3MPC_OK = 0,
  SE3MPC_ERR_NULL = -1,        /* a required pointer is NULL                               */
  SE3MPC_ERR_HORIZON = -2,     /* horizon outside [1, SE3MPC_MAX_HORIZON]                  */
  SE3MPC_ERR_SHAPE = -3,       /* B < 0, ld < B, K out of range, ...                       */
  SE3MPC_ERR_PARAM = -4,       /* non-finite / non-positive mass, dt, ...                  */
  SE3MPC_ERR_WORKSPACE = -5,   /* workspace too small                                      */
  SE3MPC_ERR_LAUNCH = -6,      /* HIP reported a launch error (see se3mpc_last_error)      */
  SE3MPC_ERR_NO_DEVICE = -7    /* no gfx950 device visible                                 */
};

/* SE3MPCConfig (planner.py:36-79) + constructor constants (planner.py:149-151), unit-stripped
 * to SI magnitudes.  se3mpc_default_params() fills the reference defaults. */
typedef struct se3mpc_params {
  int32_t horizon;              /* prediction_horizon N            (planner.py:41, default 6) */
  int32_t has_goal;             /* 0: goal_position is None -> goal terms skipped (:524,:546,:567) */
  double dt;                    /* planner.py:99-105 (timing manager: 1/400 s)                */
  double mass;                  /* 1.5 kg  (planner.py:149)                                   */
  double gravity;               /* 9.81    (planner.py:150)                                   */
  double position_weight;       /* 100     (planner.py:56)                                    */
  double velocity_weight;       /* 10      (planner.py:57)                                    */
  double acceleration_weight;   /* 1       (planner.py:58)                                    */
  double thrust_weight;         /* 0.1     (planner.py:59)                                    */
  double terminal_factor;       /* 10      (planner.py:548)                                   */
  double position_bound;        /* 100 m   (planner.py:384)                                   */
  double max_velocity;          /* 10 m/s  (planner.py:45, :388)                              */
  double max_acceleration;      /* 15      (planner.py:46, :489)                              */
  double max_thrust;            /* 25 N    (planner.py:48)                                    */
  double min_thrust;            /* 2 N     (planner.py:49)                                    */
  double max_tilt_angle;        /* pi/4    (planner.py:52, :393)                              */
  double safety_margin;         /* 1.5 m   (planner.py:64, :509)                              */
  /* scipy.optimize.minimize(method="L-BFGS-B") options as the reference sets them (:256-268) */
  int32_t max_iterations;       /* maxiter = 15 (planner.py:67)                               */
  int32_t max_corrections;      /* maxcor  = 10 (SciPy default)                               */
  int32_t max_linesearch;       /* maxls   = 20 (SciPy default)                               */
  int32_t max_fun;              /* maxfun  = 15000 (SciPy default)                            */
  double pgtol;                 /* gtol = convergence_tolerance = 0.05 (planner.py:264)       */
  double ftol;                  /* ftol = 10*convergence_tolerance = 0.5 (planner.py:265)     */
} se3mpc_params;

/* Per-problem result of the solver, mirroring scipy's OptimizeResult fields the reference
 * reads (planner.py:271-278): status 0 = converged, 1 = iteration/evaluation limit,
 * 2 = abnormal termination in line search.  `task` is the L-BFGS-B stop reason. */
typedef struct se3mpc_solve_info {
  double fun;        /* f at the returned x                                              */
  int32_t nit;       /* iterations                                                       */
  int32_t nfev;      /* objective/gradient evaluations                                   */
  int32_t status;    /* scipy status: 0, 1 or 2                                          */
  int32_t task;      /* enum se3mpc_task                                                 */
} se3mpc_solve_info;

enum se3mpc_task {
  SE3MPC_TASK_CONV_PGTOL = 1,    /* CONVERGENCE: NORM OF PROJECTED GRADIENT <= PGTOL      */
  SE3MPC_TASK_CONV_FTOL = 2,     /* CONVERGENCE: REL_REDUCTION_OF_F <= FACTR*EPSMCH       */
  SE3MPC_TASK_STOP_MAXITER = 3,  /* STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT           */
  SE3MPC_TASK_STOP_MAXFUN = 4,   /* STOP: TOTAL NO. OF F,G EVALUATIONS EXCEEDS LIMIT      */
  SE3MPC_TASK_ABNORMAL = 5,      /* ABNORMAL TERMINATION IN LNSRCH                        */
  SE3MPC_TASK_OVERFLOW = 6       /* internal: first-tier LDS too small, re-solved by the second launch; never returned */
};

/* ------------------------------------------------------------------ library / params */
int se3mpc_abi_version(void);
/* 0-terminated message of the last failing HIP call on this thread ("" if none). */
const char* se3mpc_last_error(void);
/* Number of visible gfx950 devices (0 if none / HIP not usable); never fails. */
int se3mpc_device_count(void);
/* Reference defaults (planner.py:36-79, :149-151); horizon 6, dt 1/400, has_goal 1. */
int se3mpc_default_params(se3mpc_params* out);
/* SE3MPC_OK or the error a kernel entry point would return for these params. */
int se3mpc_check_params(const se3mpc_params* p);

/* ------------------------------------------------------------------ lane layout: [row][b]
 * Common arguments: B trajectories, leading dimension ld (elements), device pointers.
 * p0, v0, goal: [3][ld].   X: [9N][ld].   T: [3N][ld].
 */

/* Cold start, replaces _create_straight_line_initialization (planner.py:329-359); with
 * project != 0 also clips into the box of _setup_optimization_bounds (planner.py:378-402),
 * which is what L-BFGS-B does to x0 before its first evaluation.  X0: [9N][ld]. */
int se3mpc_init_f32(const se3mpc_params* p, int B, int ld, const float* p0, const float* v0,
                    const float* goal, int project, float* X0, void* stream);
int se3mpc_init_f64(const se3mpc_params* p, int B, int ld, const double* p0, const double* v0,
                    const double* goal, int project, double* X0, void* stream);

/* Objective and the reference's gradient on the full decision vector, replaces
 * _objective_function (planner.py:516-550) and _objective_gradient (planner.py:552-580,
 * reproduced as is -- it is not the derivative of the objective).  f: [B]; g: [9N][ld] or NULL. */
int se3mpc_cost_grad_f32(const se3mpc_params* p, int B, int ld, const float* X, const float* goal,
                         float* f, float* g, void* stream);
int se3mpc_cost_grad_f64(const se3mpc_params* p, int B, int ld, const double* X, const double* goal,
                         double* f, double* g, void* stream);

/* Equality residuals of the translational dynamics, replaces _dynamics_constraints
 * (planner.py:426-462).  R: [6N][ld], row order as the reference: P0-p0 (3), V0-v0 (3), then
 * per k = 0..N-2 the position residual (3) and the velocity residual (3). */
int se3mpc_dynamics_residual_f32(const se3mpc_params* p, int B, int ld, const float* X, const float* p0,
                                 const float* v0, float* R, void* stream);
int se3mpc_dynamics_residual_f64(const se3mpc_params* p, int B, int ld, const double* X, const double* p0,
                                 const double* v0, double* R, void* stream);

/* Sphere-obstacle inequality residuals, replaces _obstacle_constraints (planner.py:499-514).
 * spheres: device [K][4] = (cx, cy, cz, radius), 0 <= K <= SE3MPC_MAX_SPHERES, staged in LDS.
 * C: [N*K][ld] (row k*K + j) or NULL; cmin: [B] = min over (k,j) or NULL;
 * viol: [B] = sum over (k,j) of max(0, -c) or NULL (the in-kernel reduced forms). */
int se3mpc_obstacle_residual_f32(const se3mpc_params* p, int B, int ld, const float* X, const float* spheres,
                                 int K, float* C, float* cmin, float* viol, void* stream);
int se3mpc_obstacle_residual_f64(const se3mpc_params* p, int B, int ld, const double* X, const double* spheres,
                                 int K, double* C, double* cmin, double* viol, void* stream);

/* Replaces _physical_constraints (planner.py:472-497).  C: [4N][ld]: N velocity rows,
 * N acceleration rows, then per k (max_thrust^2 - |T|^2, |T|^2 - min_thrust^2). */
int se3mpc_physical_constraints_f32(const se3mpc_params* p, int B, int ld, const float* X, float* C, void* stream);
int se3mpc_physical_constraints_f64(const se3mpc_params* p, int B, int ld, const double* X, double* C, void* stream);

/* Replaces _extract_solution_from_result + _compute_attitudes_and_rates (planner.py:582-654).
 * T: [3N][ld] (the T block of X).  acc, att, rates: [3N][ld]; thrust: [N][ld]; any may be NULL. */
int se3mpc_extract_f32(const se3mpc_params* p, int B, int ld, const float* T, float* acc, float* att,
                       float* rates, float* thrust, void* stream);
int se3mpc_extract_f64(const se3mpc_params* p, int B, int ld, const double* T, double* acc, double* att,
                       double* rates, double* thrust, void* stream);

/* Shooting form (the build's canonical "rollout"): forward rollout of the
 * recurrence that planner.py:449-460 writes as residuals, the objective of planner.py:516-550
 * on the rolled-out states, and its exact gradient wrt the thrust sequence by the reverse
 * sweep.  cost: [B]; gradT: [3N][ld] or NULL; P, V: [3N][ld] or both NULL (rolled-out states).
 * wave_keys: NULL, or [ceil(B/64)] device words: the kernel WRITES slot w = min over trajectories
 * 64w..64w+63 of (orderable(cost[b]) << 32 | (index_base + b)) -- the batch argmin fused into the
 * rollout, one plain store per wavefront, no atomics, no pre-initialisation; se3mpc_reduce_keys folds
 * the slots into one key per batch (same packed key as se3mpc_argmin_*). */
int se3mpc_rollout_cost_grad_f32(const se3mpc_params* p, int B, int ld, const float* p0, const float* v0,
                                 const float* goal, const float* T, float* cost, float* gradT, float* P,
                                 float* V, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_rollout_cost_grad_f64(const se3mpc_params* p, int B, int ld, const double* p0, const double* v0,
                                 const double* goal, const double* T, double* cost, double* gradT, double* P,
                                 double* V, uint64_t* wave_keys, uint32_t index_base, void* stream);

/* Multi-batch launch of the same kernel: `nbatch` independent batches in ONE launch (grid.y), laid out as
 * consecutive blocks -- p0, v0, goal: [nbatch][3][ld]; T, gradT: [nbatch][3N][ld]; cost: [nbatch][ld];
 * wave_keys: NULL or [nbatch][ceil(B/64)].  For callers that hold many independent sample
 * batches (Monte-Carlo sweeps, several planners): one launch amortises the launch latency that
 * dominates a single 8192-rollout batch.  1 <= nbatch <= 65535. */
int se3mpc_rollout_cost_grad_batched_f32(const se3mpc_params* p, int B, int ld, int nbatch, const float* p0,
                                         const float* v0, const float* goal, const float* T, float* cost,
                                         float* gradT, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_rollout_cost_grad_batched_f64(const se3mpc_params* p, int B, int ld, int nbatch, const double* p0,
                                         const double* v0, const double* goal, const double* T, double* cost,
                                         double* gradT, uint64_t* wave_keys, uint32_t index_base, void* stream);

/* The same rollout fused with the sphere-obstacle residuals of planner.py:499-514 evaluated on the ROLLED-OUT
 * positions (BASELINE.json config 3): per-step positions staged in LDS, sphere table in LDS, cmin[b] = min over
 * (k, j) of |P_k - c_j|^2 - (r_j + margin)^2 and viol[b] = sum of max(0, -residual); neither the states nor the
 * N*K residuals are written to HBM.  spheres: [K][4]; cmin, viol: [B] (either may be NULL); gradT may be NULL. */
int se3mpc_rollout_obstacles_f32(const se3mpc_params* p, int B, int ld, const float* p0, const float* v0,
                                 const float* goal, const float* T, float* cost, float* gradT, const float* spheres,
                                 int K, float* cmin, float* viol, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_rollout_obstacles_f64(const se3mpc_params* p, int B, int ld, const double* p0, const double* v0,
                                 const double* goal, const double* T, double* cost, double* gradT, const double* spheres,
                                 int K, double* cmin, double* viol, uint64_t* wave_keys, uint32_t index_base, void* stream);

/* Multi-batch launch of the fused kernel (grid.y = batch), operands laid out as in se3mpc_rollout_cost_grad_batched_*:
 * p0, v0, goal: [nbatch][3][ld]; T, gradT: [nbatch][3N][ld]; cost, cmin, viol: [nbatch][ld]; wave_keys: NULL or
 * [nbatch][ceil(B/64)]; ONE sphere table [K][4] shared by all batches.  1 <= nbatch <= 65535. */
int se3mpc_rollout_obstacles_batched_f32(const se3mpc_params* p, int B, int ld, int nbatch, const float* p0,
                                         const float* v0, const float* goal, const float* T, float* cost, float* gradT,
                                         const float* spheres, int K, float* cmin, float* viol, uint64_t* wave_keys,
                                         uint32_t index_base, void* stream);
int se3mpc_rollout_obstacles_batched_f64(const se3mpc_params* p, int B, int ld, int nbatch, const double* p0,
                                         const double* v0, const double* goal, const double* T, double* cost, double* gradT,
                                         const double* spheres, int K, double* cmin, double* viol, uint64_t* wave_keys,
                                         uint32_t index_base, void* stream);

/* `iters` iterations of the shooting form in ONE launch -- the on-device replacement of a host-driven optimisation loop over the
 * rollout (north_star: "replacing DART-Planner's ... optimisation loop"; reference loop site planner.py:256-268): projected gradient
 * descent on every trajectory's thrust sequence,  T <- clip(T - step * dcost/dT, thrust box of planner.py:390-400),  `iters` times,
 * then one last evaluation at the final T.  The thrust sequence stays in registers between iterations: only iteration 0 reads T_in
 * and only the last evaluation writes (T_out, cost, gradT) -- HBM traffic per launch is that of ONE rollout whatever `iters` is.
 * iters = 0 is se3mpc_rollout_cost_grad_* plus a copy of T.  T_out may alias T_in.  cost_first: NULL or [B] = the cost at T_in;
 * cost: [B] = the cost at T_out; gradT: NULL or [3N][ld] = the gradient at T_out; wave_keys as in se3mpc_rollout_cost_grad_*.
 * Multi-batch (grid.y): operands as in se3mpc_rollout_cost_grad_batched_* with T_out / cost_first laid out like T / cost.
 * se3mpc_projected_step_* is one descent step as its own launch (T_out = clip(T - step * gradT)): iterating
 * se3mpc_rollout_cost_grad_* + se3mpc_projected_step_* from the host is what this entry point replaces. */
int se3mpc_rollout_iterate_f32(const se3mpc_params* p, int B, int ld, int nbatch, int iters, double step, const float* p0,
                               const float* v0, const float* goal, const float* T_in, float* T_out, float* cost_first, float* cost,
                               float* gradT, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_rollout_iterate_f64(const se3mpc_params* p, int B, int ld, int nbatch, int iters, double step, const double* p0,
                               const double* v0, const double* goal, const double* T_in, double* T_out, double* cost_first, double* cost,
                               double* gradT, uint64_t* wave_keys, uint32_t index_base, void* stream);
/* The obstacle-aware iteration loop (BASELINE config 3 inside the loop): se3mpc_rollout_iterate_* on the objective
 *     running cost (planner.py:516-550 on the rolled-out states) + obstacle_weight * sum_k sum_j max(0, -c_kj)^2,
 *     c_kj = |P_k - c_j|^2 - (r_j + safety_margin)^2      (the reference's sphere residual, planner.py:499-514)
 * -- the BUILD'S EXTENSION: the reference forms these residuals and never hands them to its solver (:250 vs :256-268; its
 * obstacle_weight, :63 = 1000, is never read).  The exact gradient of the penalty reaches the thrust sequence through the same
 * adjoint sweep.  spheres: [K][4] rows (cx, cy, cz, r) as se3mpc_rollout_obstacles_* (0 <= K <= SE3MPC_MAX_SPHERES; K = 0 is the
 * plain loop's numbers at the obstacle form's speed).  cost / cost_first include the penalty; penalty: NULL or [B] = the penalty
 * alone at T_out (0: the plan keeps the margin of every sphere at every step).  Everything else as se3mpc_rollout_iterate_*.
 * wave_keys: written inside the launch (the library presets the [nbatch][ceil(B/64)] slots itself: workgroups of 32 trajectories fold
 * pairwise into a slot). */
int se3mpc_rollout_iterate_obstacles_f32(const se3mpc_params* p, int B, int ld, int nbatch, int iters, double step, const float* p0,
                                         const float* v0, const float* goal, const float* T_in, float* T_out, float* cost_first,
                                         float* cost, float* gradT, const float* spheres, int K, double obstacle_weight,
                                         float* penalty, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_rollout_iterate_obstacles_f64(const se3mpc_params* p, int B, int ld, int nbatch, int iters, double step, const double* p0,
                                         const double* v0, const double* goal, const double* T_in, double* T_out, double* cost_first,
                                         double* cost, double* gradT, const double* spheres, int K, double obstacle_weight,
                                         double* penalty, uint64_t* wave_keys, uint32_t index_base, void* stream);
int se3mpc_projected_step_f32(const se3mpc_params* p, int B, int ld, double step, const float* T, const float* gradT, float* T_out,
                              void* stream);
int se3mpc_projected_step_f64(const se3mpc_params* p, int B, int ld, double step, const double* T, const double* gradT, double* T_out,
                              void* stream);

/* The tail of a shooting-form plan in ONE launch of one wavefront (the caller side of se3mpc_rollout_iterate_* / _obstacles_*, i.e. what
 * _solve_se3_mpc does after its optimiser returns, planner.py:270-280 -> _extract_solution_from_result :582-602 -> _compute_attitudes_and_rates
 * :604-654): fold the descent launch's wave_keys[n_slots] to the winning key (stored to key_out if not NULL), take column
 * (key & 0xffffffff) - index_base of T ([3N][ld], the descent's T_out) as doubles, roll it out from state = (p0, v0, goal) [9 doubles; goal ignored
 * when has_goal == 0] with the recurrence of planner.py:449-460 and the objective of :516-550, extract accelerations, attitudes, body rates and
 * thrust magnitudes, and -- K > 0 -- the sphere penalty obstacle_weight * sum max(0, -c_kj)^2 left at that plan (spheres: [K][4] doubles, rows
 * (cx, cy, cz, r) before the safety margin).  out: 19 N + 3 doubles = [P | V | T | acc | att | rates (N x 3 each) | thrust (N) | cost | penalty |
 * cost + penalty]; out, key_out and state may be pinned host memory (the kernel's loads / stores then are the copies).  B >= 1. */
int se3mpc_shooting_finish_f32(const se3mpc_params* p, int B, int ld, const float* T, const uint64_t* wave_keys, int n_slots,
                               uint32_t index_base, const double* state, const double* spheres, int K, double obstacle_weight, double* out,
                               uint64_t* key_out, void* stream);
int se3mpc_shooting_finish_f64(const se3mpc_params* p, int B, int ld, const double* T, const uint64_t* wave_keys, int n_slots,
                               uint32_t index_base, const double* state, const double* spheres, int K, double obstacle_weight, double* out,
                               uint64_t* key_out, void* stream);

/* keys_out[i] = min over wave_keys[i][0..per_batch) for i < nbatch (one small workgroup per batch). */
int se3mpc_reduce_keys(const uint64_t* wave_keys, int per_batch, int nbatch, uint64_t* keys_out, void* stream);

/* Tuning knob for measurements: which implementation of the rollout the entry point above
 * launches.  0 = auto (default), 1 = exact-N register arrays (N in {6,20,30,50}; falls back to
 * 3 otherwise), 2 = per-step state tiles staged in LDS, 3 = register-light reversible sweep, 4 / 5 = 1 / 3
 * with one wavefront looping over the three axes, 6 = 32-step register bucket with guarded steps (f32, horizons 17..32;
 * else 3).  variant + 8 * (flags + 1) forces the memory-policy
 * flags of the benchmarked instantiation (horizon 30, f32, gradient): bit 0 nt loads, bit 1 nt stores,
 * bit 2 XCD-contiguous block order; the default is all three (7).  + 128 / + 256 / + 384 forces the workgroup of
 * se3mpc_rollout_obstacles_* to 3 / 8 / 4 wavefronts (default: 8 while 8 x workgroups <= 1024, else 4 for the exact-N = 50 register
 * sweep -- three axis wavefronts and a helper -- and 3 otherwise); + 128 also forces
 * se3mpc_rollout_iterate_obstacles_* to its narrow shape (3 wavefronts on 64 trajectories, sphere table in LDS; default: 7 wavefronts on
 * 32 trajectories, four of them helpers with the table in registers).  + 512 / + 1024: the
 * write-heavy float32 lane kernels never / always take their 16-byte-per-lane form (four trajectories per lane; needs B and ld
 * multiples of 4 and 16-byte aligned operands; default: from 262144 trajectories up).
 * + 2048: se3mpc_rollout_obstacles_f32 forms its residuals on the matrix core (v_mfma_f32_16x16x4_f32 on the expanded form |P|^2 - 2 P.c + |c|^2 - R^2:
 * measured evidence for DESIGN.md 5.3, no faster than the default packed-VALU difference form and four digits less precise next to a sphere's surface).
 * All compute the same quantities. */
int se3mpc_set_rollout_variant(int variant);

/* Replaces is_plan_valid (planner.py:717-737): valid[b] = 1 iff all positions finite,
 * z >= 0.1 and |v| <= 20.  P, V: [3N][ld] (V may be NULL). */
int se3mpc_is_plan_valid_f32(const se3mpc_params* p, int B, int ld, const float* P, const float* V,
                             int32_t* valid, void* stream);
int se3mpc_is_plan_valid_f64(const se3mpc_params* p, int B, int ld, const double* P, const double* V,
                             int32_t* valid, void* stream);

/* Batch argmin: *key = min over b of (orderable(cost[b]) << 32 | (index_base + b)); the packed
 * 64-bit key is what the multi-GPU path all-reduces with MIN.
 * The function resets *key itself (stream-ordered).  Decode with se3mpc_key_*. */
int se3mpc_argmin_f32(int B, const float* cost, uint32_t index_base, uint64_t* key, void* stream);
int se3mpc_argmin_f64(int B, const double* cost, uint32_t index_base, uint64_t* key, void* stream);
uint32_t se3mpc_key_index(uint64_t key);
float se3mpc_key_cost(uint64_t key);   /* f64 costs are ordered through their float rounding */

/* Population sums of a lane-layout block -- the other exchange of the multi-GPU path ("for an
 * MPPI-style averaged gradient: all-reduce(SUM) of 3N floats"): out[r] = sum_b w_b * X[r][b] for r < rows and
 * out[rows] = sum_b w_b, accumulated in float64 with a fixed summation tree.  w_b = 1 when cost == NULL (plain
 * mean of, e.g., the rollout's thrust gradient), else the MPPI weight exp(-(cost[b] - cost_ref) / temperature),
 * where cost_ref is the host argument, or -- when ref_key != NULL -- the cost held in that device key (the fused /
 * all-reduced argmin key: the global minimum, so every rank weighs on the same scale).  One all-reduce(SUM) of the
 * rows + 1 doubles, then out[r] / out[rows], gives the population mean over all ranks.
 * workspace: device double[se3mpc_population_workspace(rows, B)]. */
int se3mpc_population_workspace(int rows, int B);
int se3mpc_population_sums_f32(int rows, int B, int ld, const float* X, const float* cost, double cost_ref,
                               const uint64_t* ref_key, double temperature, double* out, double* workspace, void* stream);
int se3mpc_population_sums_f64(int rows, int B, int ld, const double* X, const double* cost, double cost_ref,
                               const uint64_t* ref_key, double temperature, double* out, double* workspace, void* stream);

/* Obstacle source: the sphere table of se3mpc_obstacle_residual_* straight from a local
 * occupancy grid, replacing the host loop of cloud/main_improved_threelayer.py:387-398 (target 20) and
 * tests/test_se3_mpc_with_mapper.py:29-33 (target 10): occupied = cells with occupancy > threshold in grid
 * order, step = max(1, n_occupied // target), spheres = every step-th occupied cell, all with `radius`.
 * positions: [M][3] (grid order of ExplicitGeometricMapper.get_local_occupancy_grid,
 * perception/explicit_geometric_mapper.py:221-248), occupancy: [M]; spheres: [cap][4] (at most 2*target-1
 * are produced; extra ones are dropped at cap), *count = number written. */
int se3mpc_spheres_from_grid_f32(const float* positions, const float* occupancy, int M, double threshold, int target,
                                 double radius, float* spheres, int cap, int32_t* count, void* stream);
int se3mpc_spheres_from_grid_f64(const double* positions, const double* occupancy, int M, double threshold, int target,
                                 double radius, double* spheres, int cap, int32_t* count, void* stream);

/* [rows][ld_in] <-> [cols][ld_out] tiled transpose through LDS (layout conversion). */
int se3mpc_transpose_f32(int rows, int cols, const float* in, int ld_in, float* out, int ld_out, void* stream);
int se3mpc_transpose_f64(int rows, int cols, const double* in, int ld_in, double* out, int ld_out, void* stream);

/* ------------------------------------------------------------------ voxel map (obstacle source)
 * Device-resident counterpart of ExplicitGeometricMapper (src/dart_planner/perception/explicit_geometric_mapper.py,
 * "mapper.py" below): the sparse voxel dict (mapper.py:74-76) becomes an open-addressing hash table in HBM owned by
 * the caller.  A voxel index (ix, iy, iz) = floor(position / resolution) (mapper.py:93-96, float64) is packed as
 * three 21-bit fields (|index| < 2^20: +-200 km at 0.2 m); SE3MPC_VOXEL_EMPTY marks a free slot.  Positions outside
 * that range read as unknown (prior).  All entry points are stream-ordered and allocate nothing. */
#define SE3MPC_VOXEL_EMPTY 0xFFFFFFFFFFFFFFFFull
#define SE3MPC_VOXEL_MAX_CELLS 1024   /* cells per axis of a local grid (axis tables staged in LDS) */

typedef struct se3mpc_voxel_map {
  uint64_t* keys;      /* [capacity] packed voxel index                                   */
  double* prob;        /* [capacity] occupancy_probability (mapper.py:31)                 */
  int32_t* count;      /* [capacity] observation_count     (mapper.py:33)                 */
  int32_t capacity;    /* power of two, >= 64; keep the load factor under ~0.7            */
  int32_t reserved;
  double resolution;   /* voxel edge in metres (mapper.py:66, default 0.2)                */
  double prior;        /* occupancy of unknown space: prob_prior = 0.5 (mapper.py:83)     */
} se3mpc_voxel_map;

/* keys = EMPTY, prob = prior, count = 0. */
int se3mpc_voxel_clear(const se3mpc_voxel_map* m, void* stream);
/* Create-or-overwrite M voxels (ijk: [M][3] voxel indices; prob: [M], or NULL for `value` everywhere; count_in:
 * [M] observation counts, or NULL to leave counts untouched) -- what add_obstacle does with 0.9
 * (mapper.py:424-447), how a host-built map is uploaded and how a table is re-hashed into a larger one.
 * failed: device int32[2], incremented (not reset): [0] per voxel that could not be stored (table full / index
 * out of range), [1] per voxel that did not exist before. */
int se3mpc_voxel_insert(const se3mpc_voxel_map* m, const int32_t* ijk, const double* prob, double value,
                        const int32_t* count_in, int M, int32_t* failed, void* stream);
/* update_map (mapper.py:102-153) for M <= SE3MPC_VOXEL_MAX_RAYS observations in their order (longer scans: call
 * again with the next chunk): origin [M][3], direction [M][3] (UNIT vectors: the caller normalises, as
 * mapper.py:265 does), distance [M] = min(hit_distance or max_range, map max_range) (:111-112), hit [M] = 1 when the
 * observation carries a hit distance (the ray's last voxel is then updated with like_hit = prob_hit, every other
 * voxel with like_miss = 1 - prob_miss, :319-323).  Voxels are walked with the reference's DDA (:251-312); the
 * clamped Bayesian update (:325-337) does not commute, so every voxel applies its observations in ray order (voxels
 * are independent of each other and proceed in parallel).
 * Workspace (device): ray_keys uint64 [M][max_len] and ray_len int32 [M] (max_len >= 3 * ceil(distance /
 * resolution) + 8 never truncates); slot_rows int32 [capacity] and row_bits uint64
 * [se3mpc_voxel_update_row_words(M, max_len)], both ALL ZERO on entry and left all zero on return.
 * stats: device int32[4] = {voxels walked, walked voxels that could not be stored, truncated rays, voxels created}
 * (reset by the call). */
#define SE3MPC_VOXEL_MAX_RAYS 1024
int se3mpc_voxel_update_rays(const se3mpc_voxel_map* m, const double* origin, const double* direction,
                             const double* distance, const int32_t* hit, int M, double like_hit, double like_miss,
                             uint64_t* ray_keys, int32_t* ray_len, int max_len, int32_t* slot_rows, uint64_t* row_bits,
                             int32_t* stats, void* stream);
long long se3mpc_voxel_update_row_words(int M, int max_len);
/* _trace_ray (mapper.py:251-312) alone, for M rays: ray_keys[ray][0 .. ray_len[ray]) = the packed indices of the
 * walked voxels in walk order (index a = ((key >> (42 - 21 a)) & 0x1FFFFF) - 2^20 for a = 0, 1, 2; SE3MPC_VOXEL_EMPTY
 * for a voxel outside the packable range).  The table is not touched (only m->resolution is used).  stats as in
 * se3mpc_voxel_update_rays ([0] voxels walked, [1] rays dropped, [2] rays truncated). */
int se3mpc_voxel_trace_rays(const se3mpc_voxel_map* m, const double* origin, const double* direction,
                            const double* distance, int M, uint64_t* ray_keys, int32_t* ray_len, int max_len,
                            int32_t* stats, void* stream);
/* Compact the occupied slots (order unspecified): ijk_out [capacity][3], prob_out / count_out [capacity] (may be
 * NULL), *n_out = number of voxels (the caller zeroes nothing: the function resets *n_out itself). */
int se3mpc_voxel_export(const se3mpc_voxel_map* m, int32_t* ijk_out, double* prob_out, int32_t* count_out,
                        int32_t* n_out, void* stream);
/* query_occupancy_batch (mapper.py:155-183): positions [M][3] -> occupancy [M] (float64 as stored). */
int se3mpc_voxel_query_f32(const se3mpc_voxel_map* m, const float* positions, int M, double* occupancy, void* stream);
int se3mpc_voxel_query_f64(const se3mpc_voxel_map* m, const double* positions, int M, double* occupancy, void* stream);
/* is_trajectory_safe (mapper.py:185-219, :339-353) for B trajectories at once: trajectory b = N rows of 3 at
 * P + b * stride (elements; stride = 9N reads the P block of se3mpc_solve_*'s X directly).  Each position is
 * checked with its 6 axis neighbours at +-margin; safe[b] = 1/0, first[b] = index of the first colliding
 * position or -1.  One wavefront per trajectory. */
int se3mpc_voxel_trajectory_safe_f32(const se3mpc_voxel_map* m, const float* P, int B, int N, long long stride,
                                     double margin, double threshold, int32_t* safe, int32_t* first, void* stream);
int se3mpc_voxel_trajectory_safe_f64(const se3mpc_voxel_map* m, const double* P, int B, int N, long long stride,
                                     double margin, double threshold, int32_t* safe, int32_t* first, void* stream);
/* get_local_occupancy_grid (mapper.py:221-248) fused with the grid -> sphere selection of
 * cloud/main_improved_threelayer.py:387-398 / tests/test_se3_mpc_with_mapper.py:29-33, without materialising the
 * grid: n = int(size / resolution) cells per axis at linspace(centre - size/2, centre + size/2, n) (numpy's
 * i * step + start, last = stop), cell order (iz, ix, iy) with iy fastest, occupied = occupancy > threshold,
 * step = max(1, n_occupied // target), every step-th occupied cell becomes (x, y, z, radius).
 * centre: HOST pointer to 3 doubles.  spheres: [cap][4]; count: device int32[2] = {spheres written, n_occupied};
 * workspace: device int32[se3mpc_voxel_local_workspace(n)]. */
int se3mpc_voxel_local_workspace(int cells_per_axis);
int se3mpc_voxel_local_spheres_f32(const se3mpc_voxel_map* m, const double* centre, double size, double threshold,
                                   int target, double radius, float* spheres, int cap, int32_t* count,
                                   int32_t* workspace, void* stream);
int se3mpc_voxel_local_spheres_f64(const se3mpc_voxel_map* m, const double* centre, double size, double threshold,
                                   int target, double radius, double* spheres, int cap, int32_t* count,
                                   int32_t* workspace, void* stream);

/* ------------------------------------------------------------------ consumer side of the contract
 * Plan sample -> geometric controller -> simulator, one drone per lane.  Reference ("controller.py" =
 * src/dart_planner/control/geometric_controller.py, "onboard.py" = src/dart_planner/control/onboard_controller.py,
 * "simulator.py" = src/dart_planner/utils/drone_simulator.py), unit-stripped and reproduced with its quirks.
 * All arrays are per-drone rows: pos, vel, att (roll, pitch, yaw), omega: [B][3]; time: double [B] (DroneState.timestamp).
 */

/* GeometricControllerConfig (controller.py:26-77) after _apply_tuning_profile (:140-158).  se3mpc_controller_default_params
 * fills the "sitl_optimized" profile (control_config.py:95-111) with the vehicle constants the reference instantiates. */
typedef struct se3mpc_controller_params {
  double kp_pos[3], ki_pos[3], kd_pos[3];      /* position PID gains                     (controller.py:38-42)   */
  double kp_att[3], kd_att[3];                 /* attitude PD gains                      (:45-47)                */
  double inertia[3];                           /* diagonal inertia, kg m^2               (:49)                   */
  double max_torque_xyz[3];                    /* per-axis torque limits, N m            (:51)                   */
  double max_integral_per_axis[3];             /* per-axis integral limits               (:68)                   */
  double max_integral_pos;                     /* norm limit of the integral             (:56)                   */
  double max_tilt_angle;                       /* rad                                    (:57)                   */
  double mass, gravity;                        /* (:59-60)                                                       */
  double max_thrust, min_thrust;               /* N; lower limit = min_thrust*mass*gravity (:61-62, :469)        */
  double tracking_error_threshold, velocity_error_threshold;   /* (:64-65, _check_tracking_performance :650-658) */
  double back_calculation_gain;                /* Kb                                     (:69)                   */
  double integral_decay_factor;                /* (:70)                                                          */
  double saturation_threshold;                 /* (:71)                                                          */
  double yaw_singularity_threshold;            /* |yaw_vector . b3| at or above which yaw is singular (:73)      */
  double default_heading_yaw;                  /* rad, for yaw_fallback_method 1         (:75)                   */
  int32_t anti_windup_method;                  /* 0 "clamping", 1 "back_calculation", 2 none of them (:67)       */
  int32_t yaw_fallback_method;                 /* 0 "skip_yaw", 1 "default_heading", 2 "maintain_current", 3 any other string (:74) */
} se3mpc_controller_params;

/* DroneSimulator.__init__ (simulator.py:41-50). */
typedef struct se3mpc_simulator_params {
  double mass, gravity, inertia[3], max_thrust, max_torque;
} se3mpc_simulator_params;

/* Mutable controller members (controller.py:87-105), SE3MPC_CONTROLLER_STATE_WORDS doubles per drone:
 * [0..2] integral_vel_error, [3] last_time (NaN = None), [4] last_valid_thrust, [5] unsaturated_thrust, [6..8] unsaturated_torque,
 * [9] failsafe_count, [10] number of failsafe gain halvings (:817-821), [11] bit set: 1 failsafe_active, 2 last_thrust_saturated,
 * 4 / 8 / 16 last_torque_saturated x / y / z. */
#define SE3MPC_CONTROLLER_STATE_WORDS 12

int se3mpc_controller_default_params(se3mpc_controller_params* out);
int se3mpc_simulator_default_params(se3mpc_simulator_params* out);
/* GeometricController.__init__ / reset() (controller.py:87-105, :840-860) for B drones: state = device double[B][12]. */
int se3mpc_controller_reset(const se3mpc_controller_params* cp, int B, double* state, void* stream);

/* compute_control (controller.py:413-512) and, when body_thrust / body_rates are given, compute_body_rate_command (:706-726) for B
 * drones: desired position / velocity / acceleration dpos, dvel, dacc [B][3] (dacc NULL = 0), yaw, yaw_rate [B] (NULL = 0).
 * Outputs (any may be NULL): thrust [B], torque [B][3], body_thrust [B] (normalised), body_rates [B][3], flags int32 [B] (bit 0
 * failsafe command returned, 1 invalid dt, 2 thrust saturated, 3 yaw singularity, 4 tilt limited, 5..7 torque saturated x/y/z).
 * `state` is read and updated. */
int se3mpc_control_f32(const se3mpc_controller_params* cp, int B, const double* time, const float* pos, const float* vel,
                       const float* att, const float* omega, const float* dpos, const float* dvel, const float* dacc,
                       const float* yaw, const float* yaw_rate, double* state, float* thrust, float* torque, float* body_thrust,
                       float* body_rates, int32_t* flags, void* stream);
int se3mpc_control_f64(const se3mpc_controller_params* cp, int B, const double* time, const double* pos, const double* vel,
                       const double* att, const double* omega, const double* dpos, const double* dvel, const double* dacc,
                       const double* yaw, const double* yaw_rate, double* state, double* thrust, double* torque, double* body_thrust,
                       double* body_rates, int32_t* flags, void* stream);

/* compute_control_fast / compute_control_from_fast_state (controller.py:253-411, :728-768), the unit-free path of the reference's
 * 400 Hz hardware loop (hardware/pixhawk_interface.py:401), for B drones: dt (seconds) is an argument; an invalid one (dt <= 0 or
 * dt > 0.1) returns vehicle_mass * vehicle_gravity and zero torque and changes nothing (flag bit 1; no failsafe).  Gravity and the
 * lower thrust limit (min_thrust * vehicle_mass * vehicle_gravity) come from the vehicle constants (common/vehicle_params.py:19-23:
 * 1.0 kg, 9.80665 m/s^2), not from cp->mass / cp->gravity.  Same `state` record as se3mpc_control_* (the two paths share integral,
 * halved gains and saturation flags); last_time, last_valid_thrust, failsafe_active and failsafe_count are not written.  flags as
 * se3mpc_control_* (bits 2, 5..7 are what the reference adds to its _thrust_saturation_count / _torque_saturation_count). */
int se3mpc_control_fast_f32(const se3mpc_controller_params* cp, double vehicle_mass, double vehicle_gravity, int B, double dt,
                            const float* pos, const float* vel, const float* att, const float* omega, const float* dpos,
                            const float* dvel, const float* dacc, const float* yaw, const float* yaw_rate, double* state, float* thrust,
                            float* torque, int32_t* flags, void* stream);
int se3mpc_control_fast_f64(const se3mpc_controller_params* cp, double vehicle_mass, double vehicle_gravity, int B, double dt,
                            const double* pos, const double* vel, const double* att, const double* omega, const double* dpos,
                            const double* dvel, const double* dacc, const double* yaw, const double* yaw_rate, double* state,
                            double* thrust, double* torque, int32_t* flags, void* stream);

/* The controller's building blocks, as the reference's own controller tests call them (tests/control/
 * test_geometric_controller_anti_windup.py, test_geometric_controller_yaw_singularity.py, tests/test_controller_torque_calculation.py),
 * for B drones on the same `state` record:
 *  - integral_update = _update_integral_error(vel_error, dt, thrust_saturated, torque_saturated) (controller.py:536-564) with its
 *    anti-windup method and _clamp_integral_per_axis: vel_error [B][3]; saturation int32 [B] (bit 0 thrust, bits 1..3 torque x/y/z;
 *    NULL = none); the unsaturated thrust / torques of the back-calculation are read from the record;
 *  - attitude_torque = _geometric_attitude_control / _fast_geometric_attitude_control(att, ang_vel, b3_des, yaw_des, yaw_rate_des)
 *    (:643-704, :348-411): torque [B][3], flags as se3mpc_control_* (bits 3, 5..7); writes the record's unsaturated torques and torque
 *    saturation flags.  inertia: NULL = diag(cp->inertia), or host double[9] row-major (the tests assign a full matrix to _fast_inertia);
 *  - desired_frame = _detect_yaw_singularity(yaw_vector, b3_des) (:160-189) -> cos_angle [B], singular int32 [B], and frame [B][9] =
 *    (b1, b2, b3): method < 0 what _geometric_attitude_control builds (cp's fallback only when singular); method 0..3 =
 *    _handle_yaw_singularity(yaw_vector, b3_des, current_yaw, method) (:191-252) unconditionally.  Any output may be NULL. */
int se3mpc_controller_integral_update_f32(const se3mpc_controller_params* cp, int B, const float* vel_error, double dt,
                                          const int32_t* saturation, double* state, void* stream);
int se3mpc_controller_integral_update_f64(const se3mpc_controller_params* cp, int B, const double* vel_error, double dt,
                                          const int32_t* saturation, double* state, void* stream);
int se3mpc_controller_attitude_torque_f32(const se3mpc_controller_params* cp, int B, const float* att, const float* omega,
                                          const float* b3_des, const float* yaw, const float* yaw_rate, const double* inertia,
                                          double* state, float* torque, int32_t* flags, void* stream);
int se3mpc_controller_attitude_torque_f64(const se3mpc_controller_params* cp, int B, const double* att, const double* omega,
                                          const double* b3_des, const double* yaw, const double* yaw_rate, const double* inertia,
                                          double* state, double* torque, int32_t* flags, void* stream);
int se3mpc_controller_desired_frame_f32(const se3mpc_controller_params* cp, int B, int method, const float* yaw_vector,
                                        const float* b3_des, const float* current_yaw, float* frame, float* cos_angle,
                                        int32_t* singular, void* stream);
int se3mpc_controller_desired_frame_f64(const se3mpc_controller_params* cp, int B, int method, const double* yaw_vector,
                                        const double* b3_des, const double* current_yaw, double* frame, double* cos_angle,
                                        int32_t* singular, void* stream);

/* compute_control_from_trajectory(state, trajectory, t) / compute_body_rate_from_trajectory -- stubs in the reference
 * (controller.py:873-875; the second does not exist), the glue its contract test calls: target = the plan sampled at sample_time[b]
 * with OnboardController._interpolate_trajectory (onboard.py:43-93), then compute_control(state, target, yaw 0, yaw rate 0).
 * Plans as in se3mpc_closed_loop_*.  target: NULL or [B][9] = the sampled (position, velocity, acceleration). */
int se3mpc_control_plan_f32(const se3mpc_controller_params* cp, int B, const double* time, const double* sample_time, const float* pos,
                            const float* vel, const float* att, const float* omega, int N, const double* timestamps, long long ts_stride,
                            const float* P, long long strideP, const float* V, long long strideV, const float* A, long long strideA,
                            double* state, float* thrust, float* torque, float* body_thrust, float* body_rates, int32_t* flags,
                            float* target, void* stream);
int se3mpc_control_plan_f64(const se3mpc_controller_params* cp, int B, const double* time, const double* sample_time, const double* pos,
                            const double* vel, const double* att, const double* omega, int N, const double* timestamps, long long ts_stride,
                            const double* P, long long strideP, const double* V, long long strideV, const double* A, long long strideA,
                            double* state, double* thrust, double* torque, double* body_thrust, double* body_rates, int32_t* flags,
                            double* target, void* stream);

/* DroneSimulator.step (simulator.py:52-72) for B drones with given commands: thrust [B] (newtons), torque [B][3]; wind as in
 * se3mpc_closed_loop_*; time, pos, vel, att, omega are advanced in place by dt. */
int se3mpc_simulator_step_f32(const se3mpc_simulator_params* sp, int B, double dt, const float* thrust, const float* torque,
                              const float* wind, long long wind_stride, double* time, float* pos, float* vel, float* att,
                              float* omega, void* stream);
int se3mpc_simulator_step_f64(const se3mpc_simulator_params* sp, int B, double dt, const double* thrust, const double* torque,
                              const double* wind, long long wind_stride, double* time, double* pos, double* vel, double* att,
                              double* omega, void* stream);

/* The closed loop of the reference's contract tests (tests/test_planner_controller_contract.py:115-162, :255-316), `nsteps` times per
 * drone in ONE launch:  t = state.timestamp;  [stop_at_plan_end: a drone whose t > timestamps[N-1] stops for good (`break`)];
 * target = plan sampled at t (onboard.py:43-93);  cmd = compute_control(state, target, yaw 0);  [step == gust_step: the wind becomes
 * gust_wind, host double[3]];  state = DroneSimulator.step(state, cmd, sim_dt) (simulator.py:52-72).
 * Plan of drone b: timestamps + b*ts_stride (double [N]), P + b*strideP, V + b*strideV, A + b*strideA ([N][3] rows; V, A may be
 * NULL = zeros; stride 0 = one plan shared by all drones; strideP = strideV = 9N on X / X + 3N and strideA = 3N on `acc` read the
 * outputs of se3mpc_solve_* in place).  wind: NULL, or newtons at wind + b*wind_stride (stride 0 = one vector).
 * time, pos, vel, att, omega, state are updated in place.  Logs (any may be NULL): log_state [nsteps][B][12] = (pos, vel, att, omega)
 * BEFORE each step, log_cmd [nsteps][B][4] = (thrust, torque) (NaN for a stopped drone), log_time double [nsteps][B],
 * steps_taken int32 [B]. */
int se3mpc_closed_loop_f32(const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int nsteps, double sim_dt,
                           int N, const double* timestamps, long long ts_stride, const float* P, long long strideP, const float* V,
                           long long strideV, const float* A, long long strideA, double* time, float* pos, float* vel, float* att,
                           float* omega, double* state, const float* wind, long long wind_stride, int gust_step,
                           const double* gust_wind, int stop_at_plan_end, float* log_state, float* log_cmd, double* log_time,
                           int32_t* steps_taken, void* stream);
int se3mpc_closed_loop_f64(const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int nsteps, double sim_dt,
                           int N, const double* timestamps, long long ts_stride, const double* P, long long strideP, const double* V,
                           long long strideV, const double* A, long long strideA, double* time, double* pos, double* vel, double* att,
                           double* omega, double* state, const double* wind, long long wind_stride, int gust_step,
                           const double* gust_wind, int stop_at_plan_end, double* log_state, double* log_cmd, double* log_time,
                           int32_t* steps_taken, void* stream);

/* The receding-horizon closed-loop Monte-Carlo of BASELINE config 5's named test shape (tests/test_monte_carlo_sim.py:24-72) in ONE
 * launch: for each of B drones, `cycles` times { se3mpc_solve_* from the drone's own (pos, vel) with the reference's cold start;
 * `substeps` x se3mpc_closed_loop_*'s step (plan sample -> compute_control -> DroneSimulator.step at sim_dt) against the fresh plan,
 * whose row k is stamped (cycle * substeps * sim_dt) + k * params->dt }.  The same code as the two entry points it fuses, hence the
 * same bits as alternating them (dart_planner_amd/control/closed_loop.py) -- without 2 x cycles kernel boundaries at each of which every
 * drone waits for the slowest one.  goal [B][3]; wind NULL or rows of wind_stride (0 = one shared row); time [B], pos / vel / att /
 * omega [B][3], state [B][SE3MPC_CONTROLLER_STATE_WORDS]: in / out as se3mpc_closed_loop_*; X_last [B][9N], acc_last [B][3N],
 * info_last [B] (each may be NULL): the last cycle's plan; overflowed: device int32, set to the number of solves that needed more
 * L-BFGS pairs than this launch's LDS image holds (never with the reference's options) -- when it is not 0 the run must be repeated
 * with the two-launch form. */
int se3mpc_monte_carlo_f32(const se3mpc_params* p, const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int cycles,
                           int substeps, double sim_dt, const float* goal, const float* wind, long long wind_stride, double* time,
                           float* pos, float* vel, float* att, float* omega, double* state, float* X_last, float* acc_last,
                           se3mpc_solve_info* info_last, int32_t* overflowed, void* stream);
int se3mpc_monte_carlo_f64(const se3mpc_params* p, const se3mpc_controller_params* cp, const se3mpc_simulator_params* sp, int B, int cycles,
                           int substeps, double sim_dt, const double* goal, const double* wind, long long wind_stride, double* time,
                           double* pos, double* vel, double* att, double* omega, double* state, double* X_last, double* acc_last,
                           se3mpc_solve_info* info_last, int32_t* overflowed, void* stream);

/* ------------------------------------------------------------------ problem layout: [b][row]
 * The batched solve: replaces _solve_se3_mpc (planner.py:230-280) = cold start (or a caller
 * x0), box, scipy.optimize.minimize(method="L-BFGS-B", jac=..., bounds=..., maxiter, gtol,
 * ftol) and _extract_solution_from_result, for B independent problems, one wavefront each.
 *   p0, v0, goal : [B][3]
 *   x0           : [B][9N] warm start, or NULL for the reference cold start
 *   X            : [B][9N] solution (result.x)
 *   info         : [B] se3mpc_solve_info
 *   acc, att, rates : [B][N][3], thrust : [B][N]   (any may be NULL)
 * All accumulations that feed a branch of L-BFGS-B run in double; `_f32` keeps the vectors in
 * float. */
/* Which generalized-Cauchy-point search se3mpc_solve_* runs while the L-BFGS memory is empty (B = theta I: the first iterate of every
 * solve and every iterate after a memory refresh).  0 (default): the closed form -- with B = theta I the published search crosses
 * exactly the breakpoints t_i <= 1/theta and stops at t = 1/theta, so the point is the projected step P(x - g/theta), one parallel
 * pass.  1: the published sequential breakpoint search (Byrd-Lu-Nocedal-Zhu 1995, algorithm CP) everywhere, as SciPy's L-BFGS-B
 * runs it: same iterates up to the rounding of its accumulated dtm = -f1/f2 (1e-8 N on thrust entries when nearly all of sum g^2
 * sits on variables that reach their bounds), ~5x slower on the first iterate.  With 1 the f64 solve reproduces the reference's
 * thrust block to 1e-9. */
int se3mpc_set_solver_variant(int variant);

int se3mpc_solve_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                     const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att,
                     float* rates, float* thrust, void* stream);
int se3mpc_solve_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                     const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att,
                     double* rates, double* thrust, void* stream);

/* Host-latency form of the solve: what SE3MPCPlanner.plan_trajectory (planner.py:215-228, one problem, the caller blocks until the
 * plan exists) binds.  ONE call = launch + wait: the same arguments as se3mpc_solve_* -- all of them HOST-PINNED, DEVICE-MAPPED
 * buffers (hipHostMalloc / torch pin_memory): the kernel reads its 72 bytes of input and writes its results in place -- plus
 *   done    : 8-byte aligned word in host-pinned, device-mapped memory
 *   ticket  : any value other than *done's current one (a per-call counter)
 *   timeout_us : after this long the wait falls back to hipStreamSynchronize
 * The kernel's last act is a system-scope fence and a store of `ticket` into *done; this function spins on that word instead of
 * paying hipStreamSynchronize's wake-up.  The batch must fit ONE wavefront (B <= 64 / lanes-per-problem: 8 problems at horizon <= 8,
 * 2 at horizon <= 32, 1 beyond; SE3MPC_ERR_SHAPE otherwise -- use se3mpc_solve_* then).  Returns when the results are in the buffers. */
int se3mpc_plan_host_f32(const se3mpc_params* p, int B, const float* p0, const float* v0, const float* goal,
                         const float* x0, float* X, se3mpc_solve_info* info, float* acc, float* att, float* rates,
                         float* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us, void* stream);
int se3mpc_plan_host_f64(const se3mpc_params* p, int B, const double* p0, const double* v0, const double* goal,
                         const double* x0, double* X, se3mpc_solve_info* info, double* acc, double* att, double* rates,
                         double* thrust, unsigned long long* done, unsigned long long ticket, double timeout_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SE3MPC_H */
