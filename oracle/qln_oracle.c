/*
 * qln_oracle.c -- see qln_oracle.h.  TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * Every function follows the reference's Julia source operation by operation
 * (file:line cited per function, paths relative to /root/reference).  Julia
 * semantics that matter for bit-level agreement of the VALUE path:
 *   - `a + b + c + d` on Float64 (and elementwise on arrays) folds left to right;
 *   - `-a * b` is (-a)*b, `-a / b` is (-a)/b, `0.5 * h * f` is (0.5*h)*f;
 *   - no FMA contraction (compile this file with -ffp-contract=off);
 *   - StaticArrays dot products are unrolled left-to-right sums.
 * The derivative path restates ForwardDiff.jacobian as forward-mode dual numbers
 * with 20 partials carried through the same RK4 code.
 */
#include "qln_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

void orc_default_model(orc_model* m) {
    /* src/planar_quadruped.jl:11-20 */
    m->g = -9.81;
    m->mb = 10.0;
    m->mf = 0.1;
    m->lb = 0.5;
    m->l1 = 0.25;
    m->l2 = 0.25;
}

/* ------------------------------------------------------------------ sizes */

int32_t orc_num_primals(int32_t N) { return ORC_NX * N + ORC_NU * (N - 1); } /* src/nlp.jl:86 */

void orc_cinds(int32_t N, int32_t k_trans, int32_t out[14]) {
    /* src/nlp.jl:48-63, 1-based inclusive ranges */
    const int n = ORC_NX;
    int32_t s, e;
    s = 1;
    e = n;
    out[0] = s;
    out[1] = e; /* c_init_inds */
    s = e + 1;
    e = e + n - 1;
    out[2] = s;
    out[3] = e; /* c_term_inds */
    s = e + 1;
    e = e + (N - 1) * n;
    out[4] = s;
    out[5] = e; /* c_dyn_inds */
    s = e + 1;
    e = e + N;
    out[6] = s;
    out[7] = e; /* c_init_contact_inds */
    s = e + 1;
    e = e + N - k_trans + 1;
    out[8] = s;
    out[9] = e; /* c_another_contact_inds */
    s = e + 1;
    e = e + 1;
    out[10] = s;
    out[11] = e; /* c_final_ctrl_inds */
    s = e + 1;
    e = e + N;
    out[12] = s;
    out[13] = e; /* c_body_pos_inds */
}

int32_t orc_num_duals(int32_t N, int32_t k_trans) {
    int32_t ci[14];
    orc_cinds(N, k_trans, ci);
    return ci[13]; /* src/nlp.jl:87: cinds[end][end] */
}

void orc_constraint_bounds(int32_t N, int32_t k_trans, double* lb, double* ub) {
    /* src/nlp.jl:66-69 */
    int32_t ci[14];
    orc_cinds(N, k_trans, ci);
    int32_t m = ci[13];
    for (int32_t i = 0; i < m; ++i) {
        lb[i] = 0.0;
        ub[i] = 0.0;
    }
    for (int32_t i = ci[12]; i <= ci[13]; ++i) ub[i - 1] = INFINITY;
}

/* --------------------------------------------------------------- dynamics */

void orc_contact_dynamics(const orc_model* m, int mode, const double* x, const double* u, double* xd) {
    /* src/planar_quadruped.jl:36-79 (mode 1), :89-132 (mode 2), :142-185 (mode 3) */
    const double g = m->g, mb = m->mb, lb = m->lb, mf = m->mf;
    const double Ib = mb * (lb * lb) / 12; /* mb * lb^2 / 12 */
    const double pbx = x[0], pby = x[1];
    const double p1x = x[3], p1y = x[4];
    const double p2x = x[5], p2y = x[6];
    const double F1x = u[0], F1y = u[1], F2x = u[2], F2y = u[3];

    const double body_acc_x = (F1x + F2x) / mb;
    const double body_acc_y = (F1y + F2y) / mb + g;
    const double tauF = -F1x * (p1y - pby) + F1y * (p1x - pbx) - F2x * (p2y - pby) + F2y * (p2x - pbx);
    const double body_w = tauF / Ib;

    xd[0] = x[7]; /* vb */
    xd[1] = x[8];
    xd[2] = x[9]; /* omega */
    if (mode == 2) { /* foot 1 free: v1 */
        xd[3] = x[10];
        xd[4] = x[11];
    } else { /* foot_1_v = zeros(2) */
        xd[3] = 0.0;
        xd[4] = 0.0;
    }
    if (mode == 1) { /* foot 2 free: v2 */
        xd[5] = x[12];
        xd[6] = x[13];
    } else {
        xd[5] = 0.0;
        xd[6] = 0.0;
    }
    xd[7] = body_acc_x;
    xd[8] = body_acc_y;
    xd[9] = body_w;
    if (mode == 2) {
        xd[10] = -F1x / mf;
        xd[11] = -F1y / mf + g;
    } else {
        xd[10] = 0.0;
        xd[11] = 0.0;
    }
    if (mode == 1) {
        xd[12] = -F2x / mf;
        xd[13] = -F2y / mf + g;
    } else {
        xd[12] = 0.0;
        xd[13] = 0.0;
    }
}

void orc_contact_dynamics_rk4(const orc_model* m, int mode, const double* x, const double* u, double* xn) {
    /* src/planar_quadruped.jl:189-197 (and :201-221 for modes 2, 3) */
    const double h = u[4];
    double f1[14], f2[14], f3[14], f4[14], s[14];
    const double hh = 0.5 * h;
    orc_contact_dynamics(m, mode, x, u, f1);
    for (int i = 0; i < 14; ++i) s[i] = x[i] + hh * f1[i];
    orc_contact_dynamics(m, mode, s, u, f2);
    for (int i = 0; i < 14; ++i) s[i] = x[i] + hh * f2[i];
    orc_contact_dynamics(m, mode, s, u, f3);
    for (int i = 0; i < 14; ++i) s[i] = x[i] + h * f3[i];
    orc_contact_dynamics(m, mode, s, u, f4);
    const double h6 = h / 6.0;
    for (int i = 0; i < 14; ++i) xn[i] = x[i] + h6 * (((f1[i] + 2 * f2[i]) + 2 * f3[i]) + f4[i]);
    xn[14] = x[14] + u[4];
}

void orc_jump_map(const double* x, double* xn) {
    /* src/planar_quadruped.jl:250-260: [x[1:4]; 0.0; x[6]; 0.0; x[8:10]; zeros(4); x[15]] */
    for (int i = 0; i < ORC_NX; ++i) xn[i] = x[i];
    xn[4] = 0.0;
    xn[6] = 0.0;
    xn[10] = xn[11] = xn[12] = xn[13] = 0.0;
}

void orc_jump_jacobian_diag(double d[ORC_NX]) {
    /* src/planar_quadruped.jl:262-263 -- note the 0 in slot 15 (quirk Q1) */
    static const double k[ORC_NX] = {1, 1, 1, 1, 0, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0};
    memcpy(d, k, sizeof(k));
}

/* ---- forward-mode duals: restatement of ForwardDiff.jacobian (src/planar_quadruped.jl:225-248) */

typedef struct dual {
    double v;
    double d[ORC_NZ];
} dual;

static dual d_const(double v) {
    dual r;
    r.v = v;
    memset(r.d, 0, sizeof(r.d));
    return r;
}
static dual d_add(dual a, dual b) {
    dual r;
    r.v = a.v + b.v;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
static dual d_sub(dual a, dual b) {
    dual r;
    r.v = a.v - b.v;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
static dual d_neg(dual a) {
    dual r;
    r.v = -a.v;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = -a.d[i];
    return r;
}
static dual d_mul(dual a, dual b) {
    dual r;
    r.v = a.v * b.v;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = b.v * a.d[i] + a.v * b.d[i];
    return r;
}
static dual d_rmul(double a, dual b) { /* Real * Dual */
    dual r;
    r.v = a * b.v;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = a * b.d[i];
    return r;
}
static dual d_divr(dual a, double b) { /* Dual / Real */
    dual r;
    r.v = a.v / b;
    for (int i = 0; i < ORC_NZ; ++i) r.d[i] = a.d[i] / b;
    return r;
}
static dual d_addr(dual a, double b) {
    a.v = a.v + b;
    return a;
}

static void dual_dynamics(const orc_model* m, int mode, const dual* x, const dual* u, dual* xd) {
    /* same statements as orc_contact_dynamics, on duals */
    const double g = m->g, mb = m->mb, lb = m->lb, mf = m->mf;
    const double Ib = mb * (lb * lb) / 12;
    const dual zero = d_const(0.0);
    dual body_acc_x = d_divr(d_add(u[0], u[2]), mb);
    dual body_acc_y = d_addr(d_divr(d_add(u[1], u[3]), mb), g);
    dual t1 = d_mul(d_neg(u[0]), d_sub(x[4], x[1]));
    dual t2 = d_mul(u[1], d_sub(x[3], x[0]));
    dual t3 = d_mul(u[2], d_sub(x[6], x[1]));
    dual t4 = d_mul(u[3], d_sub(x[5], x[0]));
    dual tauF = d_add(d_sub(d_add(t1, t2), t3), t4);
    dual body_w = d_divr(tauF, Ib);
    xd[0] = x[7];
    xd[1] = x[8];
    xd[2] = x[9];
    xd[3] = (mode == 2) ? x[10] : zero;
    xd[4] = (mode == 2) ? x[11] : zero;
    xd[5] = (mode == 1) ? x[12] : zero;
    xd[6] = (mode == 1) ? x[13] : zero;
    xd[7] = body_acc_x;
    xd[8] = body_acc_y;
    xd[9] = body_w;
    if (mode == 2) {
        xd[10] = d_divr(d_neg(u[0]), mf);
        xd[11] = d_addr(d_divr(d_neg(u[1]), mf), g);
    } else {
        xd[10] = zero;
        xd[11] = zero;
    }
    if (mode == 1) {
        xd[12] = d_divr(d_neg(u[2]), mf);
        xd[13] = d_addr(d_divr(d_neg(u[3]), mf), g);
    } else {
        xd[12] = zero;
        xd[13] = zero;
    }
}

void orc_contact_jacobian(const orc_model* m, int mode, const double* x, const double* u, double* J) {
    /* src/planar_quadruped.jl:225-230: ForwardDiff.jacobian(z -> rk4(z[1:15], z[16:20]), [x;u]) */
    dual z[ORC_NZ];
    for (int i = 0; i < ORC_NZ; ++i) {
        z[i] = d_const(i < ORC_NX ? x[i] : u[i - ORC_NX]);
        z[i].d[i] = 1.0;
    }
    const dual* xs = z;
    const dual* us = z + ORC_NX;
    dual f1[14], f2[14], f3[14], f4[14], s[14], out[ORC_NX];
    dual h = us[4];
    dual hh = d_rmul(0.5, h);
    dual_dynamics(m, mode, xs, us, f1);
    for (int i = 0; i < 14; ++i) s[i] = d_add(xs[i], d_mul(hh, f1[i]));
    dual_dynamics(m, mode, s, us, f2);
    for (int i = 0; i < 14; ++i) s[i] = d_add(xs[i], d_mul(hh, f2[i]));
    dual_dynamics(m, mode, s, us, f3);
    for (int i = 0; i < 14; ++i) s[i] = d_add(xs[i], d_mul(h, f3[i]));
    dual_dynamics(m, mode, s, us, f4);
    dual h6 = d_divr(h, 6.0);
    for (int i = 0; i < 14; ++i) {
        dual acc = d_add(d_add(d_add(f1[i], d_rmul(2.0, f2[i])), d_rmul(2.0, f3[i])), f4[i]);
        out[i] = d_add(xs[i], d_mul(h6, acc));
    }
    out[14] = d_add(xs[14], us[4]);
    for (int c = 0; c < ORC_NZ; ++c)
        for (int r = 0; r < ORC_NX; ++r) J[r + ORC_NX * c] = out[r].d[c];
}

/* ------------------------------------------------------ reference trajectory + cost */

void orc_reference_trajectory(const orc_model* m, int32_t N, int32_t k_trans, const double* xterm,
                              int32_t init_mode, double dt, double* Xref, double* Uref) {
    /* src/ref_traj.jl:6-39.  Xref[end, :] = range(0, dt*(N-1), length=N): Julia builds the range
     * in twice precision; element k is the correctly rounded (k-1)*stop/(N-1) for these sizes.  The
     * clock entry is weighted by Q[15,15] = 0 in every shipped cost, so its last bit never reaches
     * an output. */
    const double g = m->g, mb = m->mb;
    const double stop = dt * (N - 1);
    for (int32_t k = 0; k < N; ++k) {
        for (int i = 0; i < ORC_NX; ++i) Xref[(int64_t)k * ORC_NX + i] = xterm[i];
        Xref[(int64_t)k * ORC_NX + 14] = (N > 1) ? (double)((long double)k * (long double)stop / (long double)(N - 1)) : 0.0;
    }
    for (int32_t k = 0; k < N - 1; ++k) {
        double* u = Uref + (int64_t)k * ORC_NU;
        for (int i = 0; i < ORC_NU; ++i) u[i] = 0.0;
        const int K = k + 1; /* 1-based */
        if (init_mode == 1) {
            if (K <= k_trans - 1) {
                u[1] = -mb * g;
            } else {
                u[1] = -mb * g / 2;
                u[3] = -mb * g / 2;
            }
        } else {
            if (K <= k_trans - 1) {
                u[3] = -mb * g;
            } else {
                u[3] = -mb * g / 2;
                u[1] = -mb * g / 2;
            }
        }
        u[4] = (K <= k_trans - 1) ? 0.001 : 0.02;
    }
}

void orc_lqr_cost(const double* Qd, const double* Rd, const double* xf, const double* uf, double* out) {
    /* src/quadratic_cost.jl:33-42: q = -Q*xf, r = -R*uf, c = 0.5*xf'Q*xf + 0.5*uf'R*uf */
    double* Q = out;
    double* R = out + 15;
    double* q = out + 20;
    double* r = out + 35;
    for (int i = 0; i < 15; ++i) {
        Q[i] = Qd[i];
        q[i] = (-Qd[i]) * xf[i];
    }
    for (int i = 0; i < 5; ++i) {
        R[i] = Rd[i];
        r[i] = (-Rd[i]) * uf[i];
    }
    double a = 0.0, b = 0.0;
    for (int i = 0; i < 15; ++i) {
        double t = (0.5 * (Qd[i] * xf[i])) * xf[i];
        a = (i == 0) ? t : a + t;
    }
    for (int i = 0; i < 5; ++i) {
        double t = (0.5 * (Rd[i] * uf[i])) * uf[i];
        b = (i == 0) ? t : b + t;
    }
    out[40] = a + b;
}

static double quad_form(const double* D, const double* x, int n) {
    /* 0.5 * x'D * x  ==  dot(0.5 * (x .* diag(D)), x), unrolled left to right */
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = (0.5 * (D[i] * x[i])) * x[i];
        s = (i == 0) ? t : s + t;
    }
    return s;
}
static double dotn(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = a[i] * b[i];
        s = (i == 0) ? t : s + t;
    }
    return s;
}

double orc_stagecost(const double* cost, const double* x, const double* u) {
    /* src/quadratic_cost.jl:44-47: 0.5*x'Q*x + q'x + 0.5*u'R*u + r'u + c */
    const double *Q = cost, *R = cost + 15, *q = cost + 20, *r = cost + 35;
    return (((quad_form(Q, x, 15) + dotn(q, x, 15)) + quad_form(R, u, 5)) + dotn(r, u, 5)) + cost[40];
}

double orc_termcost(const double* cost, const double* x) {
    /* src/quadratic_cost.jl:49-52: 0.5*x'Q*x + q'x + c */
    const double *Q = cost, *q = cost + 20;
    return (quad_form(Q, x, 15) + dotn(q, x, 15)) + cost[40];
}

/* ------------------------------------------------------------ objective */

double orc_eval_f(const orc_problem* p, const double* Z) {
    /* src/costs.jl:6-16 */
    double J = 0.0;
    const int32_t N = p->N;
    for (int32_t k = 0; k < N - 1; ++k) {
        const double* x = Z + (int64_t)k * ORC_NZ;
        const double* u = x + ORC_NX;
        const double hk = u[4];
        J += hk * orc_stagecost(p->cost + (int64_t)k * ORC_COST_STRIDE, x, u);
    }
    J += orc_termcost(p->cost + (int64_t)(N - 1) * ORC_COST_STRIDE, Z + (int64_t)(N - 1) * ORC_NZ);
    return J;
}

void orc_grad_f(const orc_problem* p, double* grad, const double* Z) {
    /* src/costs.jl:23-34 (quirk Q2: no d(h*l)/dh term) */
    const int32_t N = p->N;
    for (int32_t k = 0; k < N - 1; ++k) {
        const double* x = Z + (int64_t)k * ORC_NZ;
        const double* u = x + ORC_NX;
        const double* cost = p->cost + (int64_t)k * ORC_COST_STRIDE;
        const double hk = u[4];
        double* gx = grad + (int64_t)k * ORC_NZ;
        for (int i = 0; i < 15; ++i) gx[i] = hk * (cost[i] * x[i] + cost[20 + i]);
        for (int i = 0; i < 5; ++i) gx[15 + i] = hk * (cost[15 + i] * u[i] + cost[35 + i]);
    }
    const double* x = Z + (int64_t)(N - 1) * ORC_NZ;
    const double* cost = p->cost + (int64_t)(N - 1) * ORC_COST_STRIDE;
    double* gx = grad + (int64_t)(N - 1) * ORC_NZ;
    for (int i = 0; i < 15; ++i) gx[i] = cost[i] * x[i] + cost[20 + i];
}

/* ---------------------------------------------------------- constraints */

static int knot_mode(const orc_problem* p, int32_t K /*1-based*/, int* jump) {
    /* src/constraints.jl:23-37 / :184-198 */
    if (K < p->k_trans - 1) {
        *jump = 0;
        return p->init_mode;
    } else if (K == p->k_trans - 1) {
        *jump = 1;
        return p->init_mode;
    }
    *jump = 0;
    return 3;
}

void orc_eval_c(const orc_problem* p, double* c, const double* Z) {
    /* src/constraints.jl:145-158 */
    const int32_t N = p->N;
    int32_t ci[14];
    orc_cinds(N, p->k_trans, ci);
    const double* xN = Z + (int64_t)(N - 1) * ORC_NZ;
    /* :149 */
    for (int i = 0; i < 15; ++i) c[ci[0] - 1 + i] = Z[i] - p->x0[i];
    /* :150 */
    for (int i = 0; i < 14; ++i) c[ci[2] - 1 + i] = xN[i] - p->xf[i];
    /* :151 -> :6-41 dynamics_constraint! */
    for (int32_t k = 0; k < N - 1; ++k) {
        const double* x = Z + (int64_t)k * ORC_NZ;
        const double* u = x + ORC_NX;
        const double* xnext = x + ORC_NZ;
        int jump;
        int mode = knot_mode(p, k + 1, &jump);
        double xn[15], xj[15];
        orc_contact_dynamics_rk4(&p->model, mode, x, u, xn);
        const double* r = xn;
        if (jump) {
            orc_jump_map(xn, xj);
            r = xj;
        }
        double* d = c + (ci[4] - 1) + (int64_t)k * ORC_NX;
        for (int i = 0; i < 15; ++i) d[i] = r[i] - xnext[i];
    }
    /* :152 -> :48-65 contact_init_constraints! */
    for (int32_t k = 0; k < N; ++k) {
        const double* x = Z + (int64_t)k * ORC_NZ;
        c[ci[6] - 1 + k] = (p->init_mode == 1) ? x[4] : x[6];
    }
    /* :153 -> :72-91 contact_another_constraints! */
    for (int32_t k = 1; k <= N - p->k_trans + 1; ++k) {
        int32_t i = k + p->k_trans - 1; /* 1-based knot */
        const double* x = Z + (int64_t)(i - 1) * ORC_NZ;
        c[ci[8] - 1 + (k - 1)] = (p->init_mode == 1) ? x[6] : x[4];
    }
    /* :154 */
    {
        const double* u = Z + (int64_t)(N - 2) * ORC_NZ + ORC_NX;
        c[ci[10] - 1] = u[1] + u[3] + p->model.mb * p->model.g;
    }
    /* :155 -> :98-113 body_pos_constraints! */
    for (int32_t k = 0; k < N; ++k) {
        const double* x = Z + (int64_t)k * ORC_NZ;
        c[ci[12] - 1 + k] = x[1] - p->model.lb / 2 * fabs(sin(x[2]));
    }
}

/* step block of knot k (0-based) with the jump mask applied: src/constraints.jl:181-198 */
static void step_block(const orc_problem* p, const double* Z, int32_t k, double* J) {
    const double* x = Z + (int64_t)k * ORC_NZ;
    const double* u = x + ORC_NX;
    int jump;
    int mode = knot_mode(p, k + 1, &jump);
    orc_contact_jacobian(&p->model, mode, x, u, J);
    if (jump) { /* jump1_jacobian() * J : Diagonal * Matrix scales rows */
        double d[ORC_NX];
        orc_jump_jacobian_diag(d);
        for (int c = 0; c < ORC_NZ; ++c)
            for (int r = 0; r < ORC_NX; ++r) J[r + ORC_NX * c] = d[r] * J[r + ORC_NX * c];
    }
}

static double clearance_dtheta(const orc_problem* p, double theta) {
    /* src/constraints.jl:269-273 (quirk Q3: theta == 0 takes the + branch) */
    const double lb = p->model.lb;
    if (theta > 0) return -lb / 2 * cos(theta);
    return lb / 2 * cos(theta);
}

void orc_jac_c_dense(const orc_problem* p, double* jac, const double* Z) {
    /* src/constraints.jl:212-291; jac is column-major m_nlp x n_nlp; only the write-set is assigned */
    const int32_t N = p->N;
    int32_t ci[14];
    orc_cinds(N, p->k_trans, ci);
    const int64_t m = ci[13];
#define JAC(r0, c0) jac[(int64_t)(r0) + m * (int64_t)(c0)]
    const int64_t xN0 = (int64_t)(N - 1) * ORC_NZ;
    /* :228 jac_init .= I(n) */
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 15; ++r) JAC(ci[0] - 1 + r, c) = (r == c) ? 1.0 : 0.0;
    /* :229 jac_term .= I(n)[1:n-1, :] */
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 14; ++r) JAC(ci[2] - 1 + r, xN0 + c) = (r == c) ? 1.0 : 0.0;
    /* :232 -> :168-205 dynamics_jacobian! */
    for (int32_t k = 0; k < N - 1; ++k) {
        double J[ORC_NX * ORC_NZ];
        step_block(p, Z, k, J);
        const int64_t r0 = (ci[4] - 1) + (int64_t)k * ORC_NX;
        const int64_t c0 = (int64_t)k * ORC_NZ;
        for (int c = 0; c < ORC_NZ; ++c)
            for (int r = 0; r < ORC_NX; ++r) JAC(r0 + r, c0 + c) = J[r + ORC_NX * c];
        /* :200 D[ci, xi[k+1]] .= -I(n) */
        for (int c = 0; c < 15; ++c)
            for (int r = 0; r < 15; ++r) JAC(r0 + r, c0 + ORC_NZ + c) = (r == c) ? -1.0 : 0.0;
    }
    /* :235-243 */
    for (int32_t k = 0; k < N; ++k) JAC(ci[6] - 1 + k, (int64_t)k * ORC_NZ + ((p->init_mode == 1) ? 4 : 6)) = 1.0;
    /* :246-256 */
    for (int32_t K = p->k_trans; K <= N; ++K) {
        int32_t i = K - p->k_trans + 1;
        JAC(ci[8] - 1 + (i - 1), (int64_t)(K - 1) * ORC_NZ + ((p->init_mode == 1) ? 6 : 4)) = 1.0;
    }
    /* :259-260 */
    JAC(ci[10] - 1, (int64_t)(N - 2) * ORC_NZ + ORC_NX + 1) = 1.0;
    JAC(ci[10] - 1, (int64_t)(N - 2) * ORC_NZ + ORC_NX + 3) = 1.0;
    /* :263-274 */
    for (int32_t k = 0; k < N; ++k) {
        const double theta = Z[(int64_t)k * ORC_NZ + 2];
        JAC(ci[12] - 1 + k, (int64_t)k * ORC_NZ + 1) = 1.0;
        JAC(ci[12] - 1 + k, (int64_t)k * ORC_NZ + 2) = clearance_dtheta(p, theta);
    }
#undef JAC
}

/* ----------------------------------------------------------- block-COO */

int32_t orc_jac_nnz_dynamic(int32_t N) { return 300 * (N - 1) + N; }

int32_t orc_jac_nnz(int32_t N, int32_t k_trans) {
    return orc_jac_nnz_dynamic(N) + 225 + 210 + 15 * (N - 1) + N + (N - k_trans + 1) + 2 + N;
}

void orc_jac_structure(const orc_problem* p, int32_t* rows, int32_t* cols) {
    const int32_t N = p->N;
    int32_t ci[14];
    orc_cinds(N, p->k_trans, ci);
    int64_t e = 0;
    for (int32_t k = 0; k < N - 1; ++k)
        for (int c = 0; c < ORC_NZ; ++c)
            for (int r = 0; r < ORC_NX; ++r) {
                rows[e] = ci[4] - 1 + k * ORC_NX + r;
                cols[e] = k * ORC_NZ + c;
                ++e;
            }
    for (int32_t k = 0; k < N; ++k) {
        rows[e] = ci[12] - 1 + k;
        cols[e] = k * ORC_NZ + 2;
        ++e;
    }
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 15; ++r) {
            rows[e] = ci[0] - 1 + r;
            cols[e] = c;
            ++e;
        }
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 14; ++r) {
            rows[e] = ci[2] - 1 + r;
            cols[e] = (N - 1) * ORC_NZ + c;
            ++e;
        }
    for (int32_t k = 0; k < N - 1; ++k)
        for (int r = 0; r < 15; ++r) {
            rows[e] = ci[4] - 1 + k * ORC_NX + r;
            cols[e] = (k + 1) * ORC_NZ + r;
            ++e;
        }
    for (int32_t k = 0; k < N; ++k) {
        rows[e] = ci[6] - 1 + k;
        cols[e] = k * ORC_NZ + ((p->init_mode == 1) ? 4 : 6);
        ++e;
    }
    for (int32_t K = p->k_trans; K <= N; ++K) {
        rows[e] = ci[8] - 1 + (K - p->k_trans);
        cols[e] = (K - 1) * ORC_NZ + ((p->init_mode == 1) ? 6 : 4);
        ++e;
    }
    rows[e] = ci[10] - 1;
    cols[e] = (N - 2) * ORC_NZ + ORC_NX + 1;
    ++e;
    rows[e] = ci[10] - 1;
    cols[e] = (N - 2) * ORC_NZ + ORC_NX + 3;
    ++e;
    for (int32_t k = 0; k < N; ++k) {
        rows[e] = ci[12] - 1 + k;
        cols[e] = k * ORC_NZ + 1;
        ++e;
    }
}

void orc_jac_c_coo(const orc_problem* p, double* vals, const double* Z) {
    const int32_t N = p->N;
    int64_t e = 0;
    for (int32_t k = 0; k < N - 1; ++k) {
        step_block(p, Z, k, vals + e);
        e += ORC_NX * ORC_NZ;
    }
    for (int32_t k = 0; k < N; ++k) vals[e++] = clearance_dtheta(p, Z[(int64_t)k * ORC_NZ + 2]);
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 15; ++r) vals[e++] = (r == c) ? 1.0 : 0.0;
    for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 14; ++r) vals[e++] = (r == c) ? 1.0 : 0.0;
    for (int32_t k = 0; k < N - 1; ++k)
        for (int r = 0; r < 15; ++r) vals[e++] = -1.0;
    for (int32_t k = 0; k < N; ++k) vals[e++] = 1.0;
    for (int32_t K = p->k_trans; K <= N; ++K) vals[e++] = 1.0;
    vals[e++] = 1.0;
    vals[e++] = 1.0;
    for (int32_t k = 0; k < N; ++k) vals[e++] = 1.0;
}

/* -------------------------------------------------------------- batched */

static void batch_problem(const orc_batch* b, int32_t i, orc_problem* p) {
    p->N = b->N;
    p->k_trans = b->k_trans[i];
    p->init_mode = b->init_mode[i];
    p->model = b->model;
    memcpy(p->x0, b->x0 + (int64_t)i * ORC_NX, sizeof(p->x0));
    memcpy(p->xf, b->xf + (int64_t)i * ORC_NX, sizeof(p->xf));
    p->cost = b->cost + (b->cost_batch > 1 ? (int64_t)i * b->N * ORC_COST_STRIDE : 0);
}

void orc_batch_eval_c_jac(const orc_batch* b, const double* Z, double* c, double* vals, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int32_t i = 0; i < b->B; ++i) {
        orc_problem p;
        batch_problem(b, i, &p);
        const double* Zi = Z + (int64_t)i * b->z_stride;
        if (c) orc_eval_c(&p, c + b->c_off[i], Zi);
        if (vals) orc_jac_c_coo(&p, vals + b->j_off[i], Zi);
    }
}

void orc_batch_eval_f_grad(const orc_batch* b, const double* Z, double* f, double* grad, int nthreads) {
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int32_t i = 0; i < b->B; ++i) {
        orc_problem p;
        batch_problem(b, i, &p);
        const double* Zi = Z + (int64_t)i * b->z_stride;
        if (f) f[i] = orc_eval_f(&p, Zi);
        if (grad) orc_grad_f(&p, grad + (int64_t)i * b->z_stride, Zi);
    }
}
