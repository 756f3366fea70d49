/*
 * qln_evaluator.h -- C ABI of the MI355X-native batched NLP evaluator for the
 * hybrid planar-quadruped landing problem.
 *
 * This is the drop-in boundary for the reference's evaluator path
 *   src/moi.jl:1-33          (MOI.eval_objective / eval_objective_gradient /
 *                             eval_constraint / eval_constraint_jacobian /
 *                             jacobian_structure on HybridNLP)
 *   src/costs.jl:6-34        (eval_f, grad_f!)
 *   src/constraints.jl:145-158, 212-291   (eval_c!, jac_c!)
 *   src/nlp.jl:13-87         (HybridNLP index maps, bounds, sizes)
 * generalised from ONE problem to a batch of B independent landing problems.
 * Plain pointers and sizes only; every entry point returns an int status
 * (QLN_OK == 0, negative = error) and never throws or exits.  The Julia `ccall`
 * veneer a maintainer would add is integration/julia/HybridNLPHIP.jl; the Python
 * ctypes mirror used by the tests is quadruped_landing_amd/_lib.py.
 *
 * Threading: a handle is not thread-safe; distinct handles are independent.  Batched (device-pointer)
 * calls are asynchronous and ordered on the handle's stream; MOI-mode (host-pointer) calls return
 * after the results are in the caller's buffers.  The library never frees, reallocates or keeps caller
 * buffers; it owns only its handle (descriptor copies and, for MOI mode, lazily allocated staging) and what the
 * caller explicitly asks it to allocate with qln_vals_alloc_placed.
 *
 * Layouts (all FP64, all offsets/strides in doubles):
 *   Z     problem b at Z + b*z_stride, length n_nlp = 20N-5,
 *         [x_1 u_1 x_2 u_2 ... x_{N-1} u_{N-1} x_N]           (src/nlp.jl:38-39,94-102)
 *   c     problem b at c + c_off[b], length m_nlp(b) = 18N - k_trans(b) + 16, in the
 *         reference's cinds order (src/nlp.jl:48-63): init 15 | term 14 | dyn 15(N-1) |
 *         contact-init N | contact-other N-k_trans+1 | final-control 1 | clearance N
 *   vals  problem b at vals + j_off[b], length nnz(b); block-COO of the reference's
 *         jac_c! write-set, state-dependent entries first:
 *           [ (N-1) step blocks, 15x20 column-major each            src/constraints.jl:186-198 ]
 *           [ N clearance d/dtheta entries                          src/constraints.jl:269-273 ]
 *           [ I(15) 15x15 col-major | I(15)[1:14,:] 14x15           src/constraints.jl:228-229 ]
 *           [ 15(N-1) diagonal -1 of the -I(n) blocks               src/constraints.jl:200     ]
 *           [ contact-init N ones | contact-other N-k_trans+1 ones  src/constraints.jl:235-256 ]
 *           [ final-control 2 ones | clearance d/dyb N ones         src/constraints.jl:259-265 ]
 *         The first 300(N-1)+N values depend on Z; the rest are constants and are
 *         written only when QLN_JAC_WRITE_CONSTANTS is set (or once by
 *         qln_jacobian_init_constants).
 *         With jac_format = QLN_JAC_FORMAT_STRUCTURAL only the first section differs: a step block holds
 *         just the entries that can be non-zero in the knot's contact mode, in column-major order of
 *         that pattern -- 71 values in contact modes 1/2, 56 at the transition knot k_trans-1 (the jump
 *         mask zeroes rows 5, 7, 11-15, src/planar_quadruped.jl:262-263), 57 in mode 3 -- so the section
 *         has 71*n1 + 56*nj + 57*n3 values (n1 = knots before the transition knot, nj = 0/1, n3 = knots
 *         from k_trans on).  Entries left out are exactly 0.0 in the reference's ForwardDiff Jacobian for
 *         every Z; qln_jacobian_structure lists the entries that remain.
 *   grad  same layout as Z.        f: one double per problem.
 */
#ifndef QLN_EVALUATOR_H
#define QLN_EVALUATOR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLN_NX 15          /* state dim   (src/planar_quadruped.jl:25) */
#define QLN_NU 5           /* control dim (src/planar_quadruped.jl:26) */
#define QLN_NZ 20
#define QLN_COST_STRIDE 41 /* Q[15] R[5] q[15] r[5] c  (src/quadratic_cost.jl:16-22) */

#define QLN_OK 0
#define QLN_ERR_INVALID_ARGUMENT (-1)
#define QLN_ERR_HIP (-2)
#define QLN_ERR_NO_DEVICE (-3)
#define QLN_ERR_UNSUPPORTED (-4)

/* flags for the Jacobian entry points */
#define QLN_JAC_WRITE_CONSTANTS 1u

/* qln_batch_desc.jac_format: layout of the step-block section of vals (see "Layouts") */
#define QLN_JAC_FORMAT_DENSE_BLOCKS 0 /* one dense 15x20 block per dynamics knot: the reference's unit of work */
#define QLN_JAC_FORMAT_STRUCTURAL 1   /* only the structurally non-zero entries of every block */

typedef struct qln_handle qln_handle;

/* PlanarQuadruped (src/planar_quadruped.jl:11-20) */
typedef struct qln_model {
    double g, mb, mf, lb, l1, l2;
} qln_model;

/* A batch of B landing problems with a common horizon N.  All pointers are HOST
 * pointers; qln_create copies them to the device.  Mirrors the arguments of
 * HybridNLP(model, obj, init_mode, k_trans, N, x0, xf) (src/nlp.jl:34-37). */
typedef struct qln_batch_desc {
    int32_t B;                /* problems */
    int32_t N;                /* knot points per problem (>= 2) */
    qln_model model;
    const int32_t* k_trans;   /* [B], 1-based start index of mode 3, 1 <= k_trans <= N+1 */
    const int32_t* init_mode; /* [B], 1 or 2 */
    const double* x0;         /* [B][15] */
    const double* xf;         /* [B][15] */
    const double* cost;       /* [cost_batch][N][41] per-knot diagonal QuadraticCost (obj vector); NULL = none yet:
                                 build it on the device with qln_set_lqr_cost before evaluating f / grad */
    int32_t cost_batch;       /* 1 = one table shared by all problems, or B */
    int64_t z_stride;         /* doubles between consecutive problems in Z/grad; 0 -> n_nlp (dense) */
    int32_t align;            /* c_off/j_off are rounded up to a multiple of this many doubles;
                                 0 -> 16 (128 B).  j_off is always kept even. */
    int32_t jac_format;       /* QLN_JAC_FORMAT_* */
} qln_batch_desc;

typedef struct qln_dims {
    int32_t B, N;
    int32_t n_nlp;        /* 20N-5                           (src/nlp.jl:86) */
    int32_t m_nlp_max;    /* max_b 18N - k_trans(b) + 16     (src/nlp.jl:87) */
    int32_t nnz_max;      /* max_b nnz(b) */
    int32_t nnz_dynamic;  /* state-dependent values at the head of a problem's vals segment: 300(N-1)+N with dense
                             blocks; the largest per-problem count of the batch in the structural format */
    int64_t z_stride;
    int64_t z_total;      /* doubles to allocate for Z / grad */
    int64_t c_total;      /* doubles to allocate for c */
    int64_t j_total;      /* doubles to allocate for vals */
} qln_dims;

const char* qln_last_error(void);        /* thread-local message of the last failing call */
/* for the companion libraries of this ABI (qln_multi.h): records `msg` in the same slot and returns `code` */
int qln_set_last_error(int code, const char* msg);
const char* qln_version(void);

/* Lifetime.  `device` is a HIP device ordinal. */
int qln_create(const qln_batch_desc* desc, int device, qln_handle** out);
int qln_destroy(qln_handle* h);
/* What qln_create will lay out for `desc`, without a device: the sizes and the per-problem offsets (exclusive scans of
 * m_nlp(b) / nnz(b) rounded up to desc->align; src/nlp.jl:48-87 per problem).  Host arithmetic only -- the same code
 * qln_create runs first; desc->x0 / xf / cost are not read.  c_off, j_off: [B] or NULL. */
int qln_layout(const qln_batch_desc* desc, qln_dims* dims, int64_t* c_off, int64_t* j_off);
/* All launches go to `hip_stream` (a hipStream_t; NULL = the default stream). */
int qln_set_stream(qln_handle* h, void* hip_stream);
int qln_synchronize(qln_handle* h);

/* Sizes and index maps (num_primals/num_duals/cinds/lb/ub of src/nlp.jl:48-87). */
int qln_get_dims(const qln_handle* h, qln_dims* out);
int qln_get_offsets(const qln_handle* h, int64_t* c_off /*[B]*/, int64_t* j_off /*[B]*/);
int qln_problem_dims(const qln_handle* h, int32_t b, int32_t* m_nlp, int32_t* nnz);
/* number of state-dependent values at the head of problem b's vals segment (step blocks + N clearance entries) */
int qln_problem_nnz_dynamic(const qln_handle* h, int32_t b, int32_t* nnz_dynamic);
int qln_constraint_index_ranges(const qln_handle* h, int32_t b, int32_t cinds[14]); /* 1-based [start,end] x 7 */
int qln_constraint_bounds(const qln_handle* h, int32_t b, double* lb, double* ub);
/* 0-based (row, col) of every entry of problem b's vals segment (MOI.jacobian_structure, src/moi.jl:31-33,
 * restricted to the jac_c! write-set). */
int qln_jacobian_structure(const qln_handle* h, int32_t b, int32_t* rows, int32_t* cols);

/* Batched, device-pointer, stream-ordered (asynchronous) evaluation. */
int qln_eval_objective(qln_handle* h, const double* Z, double* f);                 /* src/costs.jl:6-16  */
int qln_eval_objective_gradient(qln_handle* h, const double* Z, double* grad);     /* src/costs.jl:23-34 */
int qln_eval_constraint(qln_handle* h, const double* Z, double* c);                /* src/constraints.jl:145-158 */
int qln_eval_constraint_jacobian(qln_handle* h, const double* Z, double* vals, uint32_t flags); /* :212-291 */
/* The fused hot path: eval_c! and jac_c! of every knot of every problem in one launch. */
int qln_eval_constraint_and_jacobian(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags);
/* Everything an NLP iteration asks of the evaluator from ONE read of Z: f = eval_f, grad = grad_f!, c = eval_c!,
 * vals = jac_c! (src/costs.jl:6-34, src/constraints.jl:145-158, 212-291) in a single launch -- the wave that has a
 * problem's slice of Z in LDS for the constraint rows also forms the objective terms and the gradient from it.  Same
 * layouts and the same bits as the separate entry points.  Needs a cost table. */
int qln_eval_all(qln_handle* h, const double* Z, double* f, double* grad, double* c, double* vals, uint32_t flags);
/* The pair a line search -- or Ipopt's filter at a trial point, which calls eval_f and eval_g at the same x (src/moi.jl:1-3,
 * 10-13; 1696 + 1696 of the reference run's 4222 callbacks, src/main.ipynb:717-722) -- asks for: f = eval_f and c = eval_c!
 * from ONE read of Z in one launch, no derivatives.  Same layouts and the same bits as qln_eval_objective and
 * qln_eval_constraint.  Needs a cost table. */
int qln_eval_objective_and_constraint(qln_handle* h, const double* Z, double* f, double* c);
int qln_jacobian_init_constants(qln_handle* h, double* vals);
/* Products with the constraint Jacobian of jac_c! (src/constraints.jl:212-291) at Z, for the caller side of the path
 * (solver iterations on the GPU).  The Jacobian is not read from memory: every step block is re-derived
 * from Z in registers, so no vals buffer is involved and the result is the same for either jac_format.
 *   qln_eval_constraint_jvp:  y = J(Z) v      v: layout of Z (z_stride), y: layout of c (c_off)
 *   qln_eval_constraint_vjp:  g = J(Z)^T lam  lam: layout of c, g: layout of Z (entries past n_nlp are not touched)
 * Device pointers, stream-ordered. */
int qln_eval_constraint_jvp(qln_handle* h, const double* Z, const double* v, double* y);
int qln_eval_constraint_vjp(qln_handle* h, const double* Z, const double* lam, double* g);
/* One Gauss-Newton step on the constraint violation for every problem of the batch (the solver
 * iteration on the GPU, consuming the Jacobian where it is produced).  For problem b
 *     dZ_b = D x,  x = the minimum-norm minimiser of || A D x + rho ||_2   (subject to ||x|| <= radius[b] if given),
 * rho_i = c_i on the equality rows, min(c_i, 0) on the clearance rows (bounds of src/nlp.jl:66-69), A = jac_c(Z) without
 * the rows of satisfied clearance constraints, D = diag(col_scale) (NULL = identity; a zero holds a variable fixed).
 * Computed by at most max_iters CGLS iterations, stopped early once ||(AD)'(A dZ + rho)|| <= rel_tol * ||(AD)' rho|| or
 * when the iterate leaves the trust radius (it is then cut at the boundary, Steihaug-Toint).  One wavefront per
 * problem with every vector in LDS; the Jacobian is re-derived from Z inside the products and never stored.
 * c = eval_c!(Z) as written by qln_eval_constraint.  radius: [B] or NULL; col_scale: [n_nlp] or NULL (shared by all
 * problems).  info (may be NULL): QLN_GN_INFO_STRIDE doubles per problem {iterations done, ||(AD)' rho||^2,
 * ||(AD)'(A dZ + rho)||^2, ||A dZ + rho||^2 (the linear model's prediction), ||rho||^2, 1 if cut at the radius,
 * ||x||, 0}.  dZ has the layout of Z.  QLN_ERR_UNSUPPORTED if a problem does not fit the 160 KB of LDS of a CU
 * (N > 149).  Device pointers, stream-ordered. */
#define QLN_GN_INFO_STRIDE 8
int qln_gauss_newton_step(qln_handle* h, const double* Z, const double* c, double* dZ, int32_t max_iters, double rel_tol,
                          const double* radius /*[B] or NULL*/, const double* col_scale /*[n_nlp] or NULL*/,
                          double* info /*[B][QLN_GN_INFO_STRIDE] or NULL*/);
/* Batched solve of the reference NLP on the GPU -- the caller of this evaluator in the reference, solve() of
 * src/moi.jl:46-103 (objective eval_f, constraints eval_c! with the bounds of src/nlp.jl:66-69, the variable bounds
 * of src/moi.jl:51-67 incl. quirk Q6), for every problem of the batch at once, one wavefront per problem.
 * Method (quadruped_landing_amd/csrc/qln_ilqr_kernels.hip): augmented-Lagrangian iLQR -- the states are eliminated by
 * rolling the controls out from x0 with the evaluator's own RK4 step, a Newton-type step is a Riccati sweep over the
 * knots in LDS (the step blocks are the evaluator's closed form), terminal / final-control rows, clearance rows and
 * the state bounds carry multipliers + penalty, the bounds on the step length h are kept inside the sweep.
 * Z (device, layout of Z): in = initial guess, of which the CONTROLS u_k are used (h clamped to its bounds; the
 * states are rolled out from x0); out = the solution: its dynamics / initial-condition rows are zero to the last bit
 * by construction, the remaining rows to options->tol_violation when status = 0.
 * info (device, may be NULL): QLN_SOLVE_INFO_STRIDE doubles per problem {outer iterations, iLQR iterations, objective
 * f of the returned Z (the bits qln_eval_objective gives for it), violation of the returned Z (what
 * qln_constraint_violation gives for its eval_c!, and solve()'s bounds on theta and quirk Q6's on the knots after the
 * first), final penalty rho, status (0 = converged to tol_violation, 1 = iteration limit, 2 = no descent at the largest
 * penalty), augmented cost, last accepted step length, sum of h, LM mu at exit, five phase timers, 1 if the rescue phase
 * ran}.  f and the violation are measured on the trajectory that is handed back (the RK4 roll-out); status is the solver's
 * stop test on its own closed-form roll-out of the same controls, which differs from it by rounding (1e-15 per step): with
 * status = 0 the reported violation can exceed tol_violation by that much.  max_outer = 0 and rescue_outer = 0 make the
 * call a roll-out-and-report of the given controls (status 1).
 * There is no reference oracle for the iterates (the reference hands its callbacks to Ipopt); the result is judged by
 * this evaluator: qln_eval_constraint + qln_constraint_violation and qln_eval_objective on the returned Z.
 * Needs a cost table.  QLN_ERR_UNSUPPORTED if a problem does not fit the LDS of a CU (N > ~650).  Stream-ordered.
 * The first call on a handle allocates the solver's device scratch, 494 N doubles per problem (158 KB at N = 40: step
 * entries, feedback laws, the sixteen trial trajectories, multipliers), kept until qln_destroy -- a 10-GB hipMalloc at
 * B = 65 536 that took between a few ms and 2 s on this pool's boxes: make one throw-away call before timing. */
typedef struct qln_solve_options {
    int32_t max_outer;        /* multiplier updates                                   default 80   */
    int32_t max_inner;        /* iLQR iterations per multiplier update (inexact inner solves pay)   default 6 */
    double tol_violation;     /* stop when the violation is <= this                   default 1e-6 (solve()'s c_tol) */
    double inner_tol;         /* inner loop ends when the cost decrease is below inner_tol (1 + |J|)   default 1e-7 */
    double rho0, rho_factor, rho_max;  /* penalty schedule                            default 3, 5, 1e8 */
    double h_min, h_max;      /* bounds on the step length (src/moi.jl:58-61)         default 0.001, 0.02 */
    double theta_min, theta_max;       /* bounds on the body angle (src/moi.jl:54-56) default -pi/2, pi/2 */
    int32_t q6_bounds;        /* the lower bounds of quirk Q6 (yb_{k+1}, x1_{k+1} >= 0, src/moi.jl:64-65)  default 1 */
    int32_t exact_h_gradient; /* 0: objective gradient as the reference's grad_f!, which has no d(h l)/dh (quirk Q2) --
                                 the stage weights h_k are frozen within an iteration; 1: the exact gradient (couples the
                                 step lengths to the cost: use tighter inner solves, max_inner ~30)             default 0 */
    double h_prox;            /* proximal weight on the step lengths in the Newton system (it vanishes at a fixed point):
                                 with the reference's gradient the objective does not see h, the h_k are fixed by the
                                 constraints alone and wander along flat directions without it          default 1e4 */
    int32_t rescue_outer;     /* a problem still above the tolerance after max_outer updates gets this many more, with
                                 accurate inner solves (60 iterations) from a penalty of at most 1e6; info[15] = 1
                                 where that phase ran                                                     default 20 */
} qln_solve_options;
#define QLN_SOLVE_INFO_STRIDE 16
int qln_solve_default_options(qln_solve_options* opt);
/* The variable bounds solve() hands to Ipopt (src/moi.jl:51-67), for a caller that keeps Ipopt: x_l / x_u are host
 * arrays of n_nlp = 20N-5 doubles, -inf / +inf where the reference sets none.  theta in [theta_min, theta_max] at every
 * knot, h in [h_min, h_max] at every dynamics knot, and -- with q6_bounds != 0, the default -- the two lower bounds of
 * src/moi.jl:64-65 exactly where the reference puts them: 1-based 22+20(k-1) and 24+20(k-1), i.e. yb_{k+1} >= 0 and
 * x1_{k+1} >= 0 (quirk Q6; the source comment says "F", the Ipopt header of the shipped run counts 120 such variables,
 * src/main.ipynb:222).  opt = NULL: the defaults, which are the reference's literals.  Needs no handle and no GPU. */
int qln_variable_bounds(int32_t N, const qln_solve_options* opt /* NULL = defaults */, double* x_l, double* x_u);
int qln_solve(qln_handle* h, double* Z, const qln_solve_options* opt /* NULL = defaults */, double* info);
/* the same with HOST pointers (MOI-mode style: Z copied in, solved on the GPU, copied back; synchronous) -- what the
 * Julia veneer calls in place of `solve(Z0, nlp)`.  info (host, may be NULL): [B][QLN_SOLVE_INFO_STRIDE]. */
int qln_solve_host(qln_handle* h, double* Z, const qln_solve_options* opt, double* info);
/* OPT-IN EXTENSION WITHOUT A REFERENCE ORACLE.  The leg-length ("kinematic") constraint group exists in the reference
 * only as commented-out code (src/constraints.jl:115-138 values, :276-288 Jacobian, src/nlp.jl:60,70 index range and
 * bounds): the reference never computes it, so nothing can be compared with it, and it is NOT part of c / vals above.
 *   d[b][2k]   = |pb_k - p1_k|,  d[b][2k+1] = |pb_k - p2_k|      k = 0..N-1   (as the commented source defines them)
 *   bounds     0 <= d <= l1 + l2 + lb/2                                        (qln_kinematic_bounds)
 *   jac[b][2k+i][4] = d(row)/d(xb, yb, x_foot, y_foot) = (+e, -e) with e = (pb - p_foot)/|pb - p_foot|: the
 *   mathematically correct Jacobian, in the columns 20k + {0, 1, 3, 4} (foot 1) / 20k + {0, 1, 5, 6} (foot 2) of Z --
 *   the commented Jacobian indexes x[7:8] / x[9:10], which are not the feet, and is not reproduced.
 * Checked against an independent numpy statement of the formula and its complex-step derivative only
 * (tests/test_gpu_kinematic.py).  d: [B][2N] doubles, jac (may be NULL): [B][2N][4].  Device pointers, stream-ordered. */
int qln_eval_kinematic_constraint(qln_handle* h, const double* Z, double* d, double* jac);
int qln_kinematic_bounds(const qln_handle* h, double* lower, double* upper); /* the two scalars */
/* OPT-IN EXTENSION WITHOUT A REFERENCE ORACLE.  The reference has no friction constraint at all (its NLP lets the ground
 * pull on a foot); the path's stated scope names one, so it is offered as a group of its own, NOT part of c / vals above:
 * the friction pyramid |F_x| <= mu F_y of every foot that stands on the ground during dynamics knot k = 0..N-2, as two
 * linear rows per foot
 *   d[b][4k + 0] = mu F1y_k - F1x_k,  d[b][4k + 1] = mu F1y_k + F1x_k,  d[b][4k + 2], [4k + 3]: the same for foot 2,
 *   bounds 0 <= d < +inf (the two rows of a foot add up to 2 mu F_y >= 0: the ground only pushes).
 * Which feet stand is the mode schedule of the dynamics rows (src/constraints.jl:23-37): before the transition knot only
 * the foot of init_mode, from it on both; the rows of a foot in flight are 0 with zero derivatives (its force entries
 * drive the leg, src/planar_quadruped.jl:36-79).
 *   jac[b][4k + i][2] = d(row)/d(F_x, F_y) of that foot = (-1, mu) / (+1, mu), in the columns 20k + 15 + {0, 1} (foot 1) /
 *   20k + 15 + {2, 3} (foot 2) of Z.
 * Checked against a numpy statement of the same formula (tests/test_gpu_kinematic.py).  d: [B][4(N-1)] doubles, jac (may be
 * NULL): [B][4(N-1)][2].  Device pointers, stream-ordered. */
int qln_eval_friction_cone(qln_handle* h, const double* Z, double mu, double* d, double* jac);
/* viol[b] = largest violation of problem b's constraint bounds (src/nlp.jl:66-69) by c: max |c_i| over the equality
 * rows, max(0, -c_i) over the clearance rows -- the "Constraint violation" Ipopt prints for the reference's solve
 * (src/main.ipynb:712).  Device pointers; c as written by qln_eval_constraint. */
int qln_constraint_violation(qln_handle* h, const double* c, double* viol /*[B]*/);
/* The step before the path: the notebook's initial guess Z0 = packZ(nlp, Xguess, Uref)
 * (src/main.ipynb:181-198, src/nlp.jl:94-102, src/ref_traj.jl:19-34) for every problem of the batch, written
 * on the device in the handle's Z layout.  Needs k_trans >= 2 for every problem (the notebook divides by
 * k_trans - 1). */
int qln_initial_guess(qln_handle* h, double* Z);
/* The synthetic workload of SURVEY.md 8d generated on the device (8f-3): per-problem random drop states
 *   theta0 ~ U(theta_deg) degrees, y2_0 ~ U(y2), vby0 = -sqrt(two_g * H) with H ~ U(drop_height), omega0 ~ U(omega),
 * written into x0_template's entries 3, 7, 9, 10 (1-based; the notebook's xinit, src/main.ipynb:114-124) and mirrored for
 * init_mode 2.  The uniform draws are those of numpy.random.default_rng(seed) (PCG64) consumed in the host recipe's
 * order -- theta0[B], y2_0[B], H[B], omega0[B] starting `stream_offset` draws into the stream -- bit for bit: pcg_state /
 * pcg_inc are numpy.random.PCG64(seed).state's 128-bit `state` and `inc` as {high, low} words.  Replaces the x0 the
 * handle was created with. */
typedef struct qln_drop_state_sampler {
    uint64_t pcg_state[2], pcg_inc[2];
    int64_t stream_offset;
    double x0_template[15];
    double theta_deg[2], y2[2], drop_height[2], omega[2];  /* {low, high} */
    double two_g;                                          /* 2 * 9.81 in the recipe */
} qln_drop_state_sampler;
int qln_sample_drop_states(qln_handle* h, const qln_drop_state_sampler* s);
/* The ragged workload's per-problem descriptors drawn on the device (k_trans ~ U{2..N-1},
 * init_mode ~ U{1,2}; 8f-3): numpy.random.Generator.integers(low, high, size = count) on the PCG64 stream -- Lemire's
 * multiply-shift on 32-bit draws, the low half of a 64-bit output first -- starting `draw_offset` 32-BIT draws into the
 * stream.  out: HOST array [count] (the draws are made on `device` and copied back: they are the descriptors qln_create is
 * given).  numpy REJECTS a draw with probability < (high - low) / 2^32 and redraws, which shifts every later position of
 * the stream by one: *rejected returns how many draws of this call numpy would have rejected, and unless it is 0 the
 * output -- and every stream position behind this call, e.g. a qln_drop_state_sampler.stream_offset computed from the
 * draw count -- is NOT numpy's: fall back to the host generator for that seed.  (Two such calls of count B each consume
 * exactly B 64-bit outputs: the drop states then start at stream_offset = B.  A range of ONE value consumes no draw at
 * all, as in numpy.)  Needs no handle. */
int qln_sample_bounded_integers(int device, const uint64_t pcg_state[2], const uint64_t pcg_inc[2], int64_t draw_offset, int32_t low,
                                int32_t high /* exclusive */, int64_t count, int32_t* out, int64_t* rejected);
/* Z <- Z + N(0, sigma^2) on every entry, step lengths h then clipped to [h_min, h_max] (redraw_h = 0) or redrawn
 * U(h_min, h_max) (redraw_h != 0) -- the evaluation point of SURVEY.md 8d from qln_initial_guess's Z0.  The normal
 * draws are Box-Muller on the sampler's stream from `stream_offset`: the recipe's distribution, not numpy's numbers. */
int qln_perturb_point(qln_handle* h, const qln_drop_state_sampler* s, double* Z, double sigma, double h_min, double h_max,
                      int redraw_h);
/* the handle's current boundary states, copied to host arrays ([B][15] each; either may be NULL) */
int qln_get_boundary_states(qln_handle* h, double* x0, double* xf);
/* The notebook's objective built on the device: obj[k] = LQRCost(Q, R, Xref[k], Uref[k]), k < N, and
 * obj[N] = LQRCost(Qf, R*0, Xref[N], Uref[1]) (src/main.ipynb:158-161, src/quadratic_cost.jl:33-42) with Xref/Uref of
 * reference_trajectory(model, N, k_trans, xf, init_mode, dt) (src/ref_traj.jl:6-39).  Q, Qf: 15 diagonal entries,
 * R: 5.  per_problem != 0 builds one table per problem (k_trans / init_mode / xf may differ), else one shared table
 * from problem 0.  Replaces any previous cost table of the handle. */
int qln_set_lqr_cost(qln_handle* h, const double* Qdiag, const double* Rdiag, const double* Qfdiag, double dt,
                     int per_problem);
/* Copies the handle's cost table ([cost_batch][N][41]) to a host buffer; cost_batch is returned through *cost_batch. */
int qln_get_cost(qln_handle* h, double* cost_host, int32_t* cost_batch);

/* MOI mode: HOST pointers, synchronous.  Same layouts.  Batches of up to 8 MB per callback are evaluated on mapped
 * pinned host memory directly (one launch, no copies); larger ones are staged through device memory. */
int qln_eval_objective_host(qln_handle* h, const double* Z, double* f);
int qln_eval_objective_gradient_host(qln_handle* h, const double* Z, double* grad);
int qln_eval_constraint_host(qln_handle* h, const double* Z, double* c);
int qln_eval_constraint_jacobian_host(qln_handle* h, const double* Z, double* vals);
/* Reference-compatible dense Jacobian of ONE problem (src/moi.jl:15-24): `jac` is a host,
 * column-major m_nlp x n_nlp buffer; exactly the jac_c! write-set is assigned (explicit zeros of
 * the identity blocks included), every other entry is left untouched.  `b` selects the problem. */
int qln_eval_constraint_jacobian_dense_host(qln_handle* h, int32_t b, const double* Z, double* jac);

/* Placement-aware allocation of the Jacobian buffer (optional; every entry point also accepts plain hipMalloc memory).
 * On MI355X the store bandwidth a buffer sustains depends on where it lies physically: device memory behaves as
 * 32-GiB regions, and the evaluator's eight write fronts (one per XCD, the first four in the first half of vals) run
 * ~20 % faster when the two halves of the buffer lie in different regions (DESIGN.md section 5,
 * profiles/r01_placement_regions.txt).  This call builds such a buffer with the HIP virtual-memory API: it maps
 * j_total doubles + 64 GiB of physical memory (less if less is free) in 256-MiB chunks behind one virtual range,
 * times the fused launch on windows of that range, keeps the fastest window and returns every chunk outside it to the
 * driver.  Buffers under 1 GiB are mapped as they come.  Z as for qln_eval_constraint_and_jacobian; c: the constraint
 * buffer the timed launches write -- it is OVERWRITTEN -- or NULL to have the call use a scratch buffer of its own.
 * *vals is at least 2-MiB aligned and holds j_total doubles; constants are not written.  ms_best (may be NULL): launch
 * time on the window kept.  QLN_ERR_HIP if the memory or the virtual-memory API is not available; nothing stays
 * allocated or mapped after a failure.
 * Release with qln_vals_free_placed (qln_destroy releases what is left).  Both wait for the whole DEVICE to go idle
 * before unmapping -- the buffer is plain memory to its users, so work on any stream may still be touching it -- and
 * report a failing unmap / release instead of hiding it.  The physical memory goes back to the driver; the virtual
 * range stays reserved for the life of the process, because on ROCm 7.2 an address that has been unmapped keeps
 * serving accesses through its old translations if it is ever mapped again (bench/vmm_va_reuse.cpp,
 * profiles/r02_vmm_va_reuse.txt): ~70 GiB of address space per call, of 128 TiB. */
int qln_vals_alloc_placed(qln_handle* h, const double* Z, double* c, double** vals, float* ms_best);
/* The same with the transient memory the scan may map beyond j_total doubles chosen by the caller (qln_vals_alloc_placed
 * = 64 GiB: two region lengths, so the slab contains two region boundaries and the scan has two chances of a clean
 * straddling window).  32 GiB gives one boundary, 0 maps the buffer as it comes (no scan: a plain allocation's luck).
 * Measured consequence: profiles/r03_placement_budget.txt.  The virtual range reserved (and retired, see below) is
 * j_total * 8 + transient_bytes. */
int qln_vals_alloc_placed_budget(qln_handle* h, const double* Z, double* c, int64_t transient_bytes, double** vals, float* ms_best);
/* Address space retired by placed allocations in this process so far (their virtual ranges are never returned, see
 * above) and the cap at which qln_vals_alloc_placed refuses with QLN_ERR_UNSUPPORTED and a message that says so (64 TiB of
 * the 128 TiB: ~900 default-size calls).  Either pointer may be NULL.  Needs no handle. */
int qln_vals_placed_address_space(int64_t* retired_bytes, int64_t* cap_bytes);
int qln_vals_free_placed(qln_handle* h, double* vals);
/* What the placement scan of a buffer returned by qln_vals_alloc_placed did (each out pointer may be NULL). */
int qln_vals_placed_info(const qln_handle* h, const double* vals, int64_t* chunk_bytes, int64_t* chunks_scanned,
                         int64_t* window_first_chunk);

/* Measurement helper for bench.py: runs `warmup` + `iters` launches of the fused hot path on the
 * handle's stream and returns each timed launch's duration from HIP events (milliseconds). */
int qln_time_constraint_and_jacobian(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags,
                                     int32_t warmup, int32_t iters, float* ms_each /*[iters]*/);
/* The same launches back to back with nothing between them: ONE pair of HIP events around all `iters` launches;
 * *ms_total = elapsed ms of the whole run (the launches may overlap at their edges: the next one's first workgroups start while
 * the previous one's last drain).  Average launch time = *ms_total / iters. */
int qln_time_constraint_and_jacobian_total(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags,
                                           int32_t warmup, int32_t iters, float* ms_total);

#ifdef __cplusplus
}
#endif
#endif
