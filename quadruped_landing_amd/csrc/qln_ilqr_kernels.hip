// qln_ilqr_kernels.hip -- the caller side of the evaluator on the GPU: a batched solve of the
// reference NLP -- what solve() hands to Ipopt in src/moi.jl:46-103 -- with one wavefront per landing problem.
//
// Method: augmented-Lagrangian iLQR.  The NLP is an optimal-control problem: the dynamics rows
// (src/constraints.jl:6-41) tie x_{k+1} to (x_k, u_k), so the states are eliminated by rolling the controls out from
// x0 (the iterations with the step in closed form, the result once more with the evaluator's own RK4 step: the
// solution's dynamics / initial-condition / contact rows are then zero to the last bit), and a Newton-type step on the
// controls is a Riccati sweep over the knots -- 15x15 / 15x5 / 5x5 blocks that live in the wave's LDS.  What the roll-out does not satisfy by construction -- the terminal rows
// (src/constraints.jl:150), the final-control row (:154), the clearance inequalities (:98-113) and solve()'s variable
// bounds on theta and on the entries quirk Q6 bounds (src/moi.jl:51-67) -- carries multipliers and a quadratic penalty;
// the bounds on the step length h (a control) are kept inside the sweep (clamped feed-forward, gain row zeroed).
//   outer loop  : multipliers  lam <- max(0, lam + rho g)  /  lam + rho e ;  rho x rho_factor while the violation stalls
//   inner loop  : backward Riccati sweep with Levenberg-Marquardt mu on Quu, Gauss-Newton Hessian of the penalty terms;
//                 forward: sixteen step lengths alpha = 1, 1/2, ... 2^-15 tried at once, one lane per alpha, closed-loop
//                 roll-outs; the one with the lowest augmented cost is taken if it lowers it (a few per multiplier
//                 update: inexact inner solves).
// Objective gradient: by default the reference's own grad_f! (src/costs.jl:23-34), which has no d(h_k l_k)/dh_k
// (quirk Q2): within an iteration the stage weights h_k are frozen -- the fixed point is the kind of point Ipopt's
// run tends to with that gradient.  exact_h adds the missing term (a stationary point of the true objective).
//
// The step Jacobians A_k, B_k are the closed-form blocks of the evaluator (QLN_STEP_BASE / QLN_STEP_ENTRIES), derived
// once per sweep by lane = knot and parked in a global scratch.  Per problem that scratch holds the step entries
// (88 N doubles), the feedback law of every knot (80 N), the sixteen trial trajectories (320 N) and the inequality
// multipliers (6 N); the wave's LDS holds the current trajectory, 9 scalars per knot and the sweep's matrices
// (29 N + 1.4 k doubles: 20.3 KB at N = 40, eight waves per CU).
//
// The reference holds nothing to compare the iterates with (it hands its callbacks to Ipopt 3.13 + MUMPS); the result
// is checked by the evaluator itself: constraint violation and objective of the returned Z (tests/test_gpu_solve.py).
#include "qln_kernel_common.h"

#include <cmath>
#include <cstdlib>
#include <utility>

namespace qln {
namespace {

constexpr int kCUs = 256;                  // MI355X
constexpr size_t kLdsPerCU = 160 * 1024;
#ifndef QLN_ALPHAS
#define QLN_ALPHAS 16
#define QLN_ALPHA_STEP 1
#endif
constexpr int kAlphas = QLN_ALPHAS;        // step lengths tried per iteration: alpha_a = 2^(-a kAlphaStep), a = 0 .. kAlphas-1
                                           // (16 = one DPP row: the roll-outs take the feedback law by row broadcast; only
                                           // QLN_ALPHA_STEP is still a free knob)
constexpr int kAlphaStep = QLN_ALPHA_STEP;
constexpr int kPerTraj = kWave / kAlphas;  // lanes that share the merit evaluation of one trial trajectory
constexpr int kIneq = 6;       // inequality rows per knot
constexpr int kEnt = 88;       // doubles per knot in the step-entry scratch (85 used)
constexpr int kKn = 9;         // per-knot scalars kept in LDS
constexpr int kLd = 17;        // leading dimension of the sweep's LDS matrices: 16 columns + 1 (a stride of 16 doubles puts a column in two LDS banks)
constexpr int kKg = 80;        // doubles per knot of the feedback law in the scratch: 5 rows of [15 gains, feed-forward]
// The sweep needs four entries of A = d x+/d x per column c -- the diagonal A(c, c), A(2, c), A(9, c) and the coupling
// A(c-7, c) (step_structure_ok below) -- in T = P A by column and in Qxx = .. + A'T by row, the same four either way.  Its LDS
// image is therefore not the 15x15 matrix but the compact table AC[16][4] = {A(c,c), A(2,c) (0 for c = 2), A(9,c) (0 for c = 9),
// A(c-7,c) (0 where there is no coupling)}: a column's four are two 16-byte reads at one address instead of four reads at four
// computed ones.  Row 15 and the slots no step block fills hold 0.0 for the whole sweep, as does B(1, 0) of the B image
// (15x5, behind the table): "value or zero" is then ONE read with a selected address -- hipcc turns a select between a loaded
// value and a constant into a branch around the load, with a wait of its own, and the reads of a phase stop overlapping.
constexpr int kAcB = 64;       // offset of the B image behind the table
constexpr int kAZero = 60;     // AC[15][0]
constexpr int kBZero = 5;      // B(1, 0)
__host__ __device__ constexpr int ac_slot(int row, int col) {  // where A(row, col) lies in the table (see above)
    return 4 * col + (row == col ? 0 : row == 2 ? 1 : row == 9 ? 2 : 3);
}

// per-knot scalars (lane = knot phase -> backward sweep)
enum { KN_W = 0, KN_T0 = 1, /* t0..t5 = max(0, lam + rho g); the row is active where t > 0 */ KN_CQ = 7 /* (lb/2) cos(theta) */, KN_ELL = 8 };
// (the final-control row's lam + rho e lives in leq[15])

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
__device__ __forceinline__ double act(double t) { return t > 0 ? 1.0 : 0.0; }
// sum over the four lanes of a quad (all four active), the same bits in each of them
template <int CTRL>
__device__ __forceinline__ double dpp_quad_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_quad_perm<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v += dpp_quad_perm<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    return v;
}
// lane I of each row of sixteen lanes, in every lane of that row (DPP row_newbcast: one v_mov_b64_dpp, no LDS, no wait).
// The source lane must be enabled.
template <int I>
__device__ __forceinline__ double row_bcast(double v) {
    static_assert(I >= 0 && I < 16, "a lane of the row");
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + I, 0xf, 0xf, false);
}
// u[j] += K[j](lane I of the row) * dx for the five rows of a feedback law: v_fmac_f64 with the DPP operand -- the broadcast
// costs no instruction of its own (hipcc does not fold a v_mov_b64_dpp into the multiply-add, hence the assembly; fused like
// fma()).  The block opens with the wait states an instruction with a DPP operand needs after one of its registers (two) or
// EXEC (five: FIRST, the block that follows the branch into the roll-out lanes' code) was last written by a VALU
// instruction -- the compiler does not see through inline assembly to insert them.
template <int I, bool FIRST>
__device__ __forceinline__ void fmac5_row_bcast(double (&u)[5], const double (&K)[5], double dx) {
    static_assert(I >= 0 && I < 16, "a lane of the row");
    if constexpr (FIRST) {
        asm("s_nop 4\n\t"
            "v_fmac_f64_dpp %0, %5, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %7, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %3, %8, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %4, %9, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4])
            : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(dx), "n"(I));
    } else {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %5, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %6, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %7, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %3, %8, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %4, %9, %10 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
            : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4])
            : "v"(K[0]), "v"(K[1]), "v"(K[2]), "v"(K[3]), "v"(K[4]), "v"(dx), "n"(I));
    }
}
// acc += (lane I0 + i of the row's `rec`) * v[i], i = 0 .. n-1: the same DPP-operand multiply-add, for the stage cost of the
// merit evaluation -- a knot's 41-double cost record lies across the sixteen lanes of the lane group that evaluates it
// (three registers: rec[0..15], rec[16..31], rec[32..40]), fetched with three coalesced loads a batch ahead, instead of every
// lane issuing 41 loads of its own where the value is needed.  One leading s_nop covers the DPP hazards (see above).
template <int I0, int NOP>
__device__ __forceinline__ void fmac_row_bcast_8(double& acc, double rec, double v0, double v1, double v2, double v3, double v4,
                                                 double v5, double v6, double v7) {
    static_assert(I0 >= 0 && I0 + 7 < 16, "lanes of the row");
    asm("s_nop %10\n\t"
        "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %6 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %7 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %8 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %1, %9 row_newbcast:%18 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(rec), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7), "n"(NOP), "n"(I0), "n"(I0 + 1),
          "n"(I0 + 2), "n"(I0 + 3), "n"(I0 + 4), "n"(I0 + 5), "n"(I0 + 6), "n"(I0 + 7));
}
template <int I, int NOP>
__device__ __forceinline__ void fmac_row_bcast_1(double& acc, double rec, double v) {
    asm("s_nop %3\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rec), "v"(v), "n"(NOP), "n"(I));
}
// l_k(x, u) of a stage knot from the record held across the row (see above): sum_e D_e (z_e^2 / 2) + d_e z_e + c, z = [x; u].
// Eight terms per block, so that only eight squares are live at a time; two independent chains.
__device__ __forceinline__ double stage_ell_row(const double (&cr)[3], const double (&x)[15], const double (&u)[5]) {
    auto hsq = [](double z) { return 0.5 * z * z; };
    double a0 = 0.0, a1 = 0.0;
    // rec[0..15] = D_0 .. D_15 (the squares of x_0..x_14, u_0)
    fmac_row_bcast_8<0, 4>(a0, cr[0], hsq(x[0]), hsq(x[1]), hsq(x[2]), hsq(x[3]), hsq(x[4]), hsq(x[5]), hsq(x[6]), hsq(x[7]));
    // rec[20..31] = d_0 .. d_11 at lanes 4..15 of the second register
    fmac_row_bcast_8<8, 1>(a1, cr[1], x[4], x[5], x[6], x[7], x[8], x[9], x[10], x[11]);
    fmac_row_bcast_8<8, 1>(a0, cr[0], hsq(x[8]), hsq(x[9]), hsq(x[10]), hsq(x[11]), hsq(x[12]), hsq(x[13]), hsq(x[14]), hsq(u[0]));
    // rec[16..19] = D_16 .. D_19 (u_1..u_4), rec[20..23] = d_0 .. d_3
    fmac_row_bcast_8<0, 1>(a1, cr[1], hsq(u[1]), hsq(u[2]), hsq(u[3]), hsq(u[4]), x[0], x[1], x[2], x[3]);
    // rec[32..39] = d_12 .. d_19, rec[40] = c
    fmac_row_bcast_8<0, 1>(a0, cr[2], x[12], x[13], x[14], u[0], u[1], u[2], u[3], u[4]);
    fmac_row_bcast_1<8, 1>(a1, cr[2], 1.0);
    return a0 + a1;
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// the value lane `src` holds, in every lane (wave-uniform: it lives in scalar registers)
__device__ __forceinline__ double read_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kWave));
    return v;
}

struct StageIn {
    const double* rec;   // cost record [41] (global, wave-uniform address)
    const double* lam5;  // inequality multipliers of the knot (global scratch)
    double rho, w;       // penalty, weight on the stage cost (h_k, frozen h_k, or 1 for the terminal knot)
    bool has_u;          // k < N-1
    bool ineq;           // k >= 1: the knot's inequalities are live (x_1 = x0 is data)
    bool q6;             // bounds yb >= 0, x1 >= 0 of quirk Q6
    bool final_ctrl;     // k == N-2
    bool terminal;       // k == N-1
    double lam_fc;       // multiplier of the final-control row
    const double* lam_term;  // [14]
    const double* xf;        // [15]
    double mbg, lb, th_lo, th_hi;
};
struct StageOut {
    double val;    // augmented stage cost
    double ell;    // l_k(x, u) (no weight)
    double t[kIneq];   // max(0, lam_j + rho g_j)
    double g[kIneq];   // constraint values g_j <= 0
    double cq;         // (lb/2) cos(theta): d g0 / d theta = +cq, d g1 / d theta = -cq
    double e_fc;   // final-control residual
    double viol;   // largest violation of this knot's constraints
};

// Augmented-Lagrangian stage cost of one knot (src/costs.jl:6-16 for the objective part).  x[15], u[5] in registers.
template <bool ELL_GIVEN = false>
__device__ __forceinline__ void stage_eval(const StageIn& I, const double (&x)[15], const double (&u)[5], StageOut& o, double ell_in = 0.0) {
    double ell = ell_in;
    if constexpr (!ELL_GIVEN) {
        ell = I.rec[40];
#pragma unroll
        for (int i = 0; i < 15; ++i) ell += (0.5 * (I.rec[i] * x[i])) * x[i] + I.rec[20 + i] * x[i];
        if (I.has_u) {
#pragma unroll
            for (int i = 0; i < 5; ++i) ell += (0.5 * (I.rec[15 + i] * u[i])) * u[i] + I.rec[35 + i] * u[i];
        }
    }
    o.ell = ell;
    double val = I.w * ell;
    double viol = 0.0;
    // Clearance yb - (lb/2)|sin theta| >= 0 (src/constraints.jl:98-113) as the two smooth rows it is the intersection
    // of, yb -+ (lb/2) sin theta >= 0: same feasible set, no kink at theta = 0 -- where landed trajectories live and
    // where the kinked row stalls every Newton-type step.  With zero multipliers and yb > lb/2 both rows are inactive
    // whatever theta is (|sin| <= 1): the trigonometry is skipped and the result is the same.
    if (I.ineq && (I.lam5[0] > 0 || I.lam5[1] > 0 || !(x[1] > I.lb / 2))) {
        double sth, cth;
        sincos(x[2], &sth, &cth);
        o.g[0] = -(x[1] - I.lb / 2 * sth);
        o.g[1] = -(x[1] + I.lb / 2 * sth);
        o.cq = (I.lb / 2) * cth;
    } else {
        o.g[0] = o.g[1] = I.lb / 2 - x[1];  // an upper bound of both rows, negative here
        o.cq = 0.0;
    }
    o.g[2] = x[2] - I.th_hi;                          // theta <= pi/2                  (src/moi.jl:55-56)
    o.g[3] = I.th_lo - x[2];
    o.g[4] = -x[1];                                   // Q6: "22 + 20(k-1)" = yb_{k+1}   (src/moi.jl:64)
    o.g[5] = -x[3];                                   // Q6: "24 + 20(k-1)" = x1_{k+1}   (src/moi.jl:65)
#pragma unroll
    for (int j = 0; j < kIneq; ++j) {
        const bool on = I.ineq && (j < 4 || I.q6);
        const double l = I.lam5[j];
        const double t = on ? fmax(0.0, l + I.rho * o.g[j]) : 0.0;
        o.t[j] = t;
        if (on) {
            val += (t * t - l * l) / (2 * I.rho);
            viol = fmax(viol, o.g[j]);
        }
    }
    o.e_fc = 0.0;
    if (I.final_ctrl) {                               // F1y + F2y + mb g = 0               (src/constraints.jl:154)
        o.e_fc = u[1] + u[3] + I.mbg;
        val += I.lam_fc * o.e_fc + 0.5 * I.rho * o.e_fc * o.e_fc;
        viol = fmax(viol, fabs(o.e_fc));
    }
    if (I.terminal) {                                 // x_N[1:14] = xf[1:14]               (src/constraints.jl:150)
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const double e = x[i] - I.xf[i];
            val += I.lam_term[i] * e + 0.5 * I.rho * e * e;
            viol = fmax(viol, fabs(e));
        }
    }
    o.val = val;
    o.viol = viol;
}

// one dynamics knot of the roll-out: the evaluator's RK4 step + jump map (src/constraints.jl:19-38)
__device__ __forceinline__ void step_forward(const BatchParams& P, int k, int kt, int im, double Ib, const double (&x)[15],
                                             const double (&u)[5], double (&xn)[15]) {
    const int K = k + 1;
    const int mode = (K <= kt - 1) ? im : 3;
    const bool jump = (K == kt - 1);
    const bool f1free = (mode == 2), f2free = (mode == 1);
    StepConst sc;
    sc.abx = (u[0] + u[2]) / P.mb;
    sc.aby = (u[1] + u[3]) / P.mb + P.g;
    sc.a1x = f1free ? (-u[0] / P.mf) : 0.0;
    sc.a1y = f1free ? (-u[1] / P.mf + P.g) : 0.0;
    sc.a2x = f2free ? (-u[2] / P.mf) : 0.0;
    sc.a2y = f2free ? (-u[3] / P.mf + P.g) : 0.0;
    rk4_step(x, u, sc, f1free, f2free, Ib, xn);
    if (jump) {
        xn[4] = 0.0;
        xn[6] = 0.0;
        xn[10] = xn[11] = xn[12] = xn[13] = 0.0;
    }
}


// The same knot in closed form, for the TRIAL roll-outs.  With zero-order-hold forces every acceleration but the body's
// angular one is constant over the step, and RK4 reproduces p+ = p + h v + h^2 a/2, v+ = v + h a, and the cubic / quartic
// in h for omega / theta (file header of qln_kernels.hip) up to rounding: 1.3e-15 against the RK4 step on random states.
// ~70 flops instead of ~400, reciprocals instead of divisions.  The trajectory handed back to the caller is rolled out once
// more with the evaluator's RK4 step at the very end, so that its dynamics rows are still zero to the last bit.
struct FastStep {
    double imb, imf, iIb, g;
};
__device__ __forceinline__ void step_fast(const FastStep& C, int k, int kt, int im, const double (&x)[15], const double (&u)[5],
                                          double (&xn)[15]) {
    const int K = k + 1;
    const int mode = (K <= kt - 1) ? im : 3;
    const bool jump = (K == kt - 1);
    const double m1 = (mode == 2) ? 1.0 : 0.0, m2 = (mode == 1) ? 1.0 : 0.0;
    const double h = u[4], h2 = h * h, hh2 = 0.5 * h2, h3_6 = h2 * h * (1.0 / 6.0), h4_24 = h2 * h2 * (1.0 / 24.0);
    const double abx = (u[0] + u[2]) * C.imb, aby = (u[1] + u[3]) * C.imb + C.g;
    const double a1x = m1 * (-u[0] * C.imf), a1y = m1 * (-u[1] * C.imf + C.g);
    const double a2x = m2 * (-u[2] * C.imf), a2y = m2 * (-u[3] * C.imf + C.g);
    const double r1x = x[3] - x[0], r1y = x[4] - x[1], r2x = x[5] - x[0], r2y = x[6] - x[1];
    const double w1x = m1 * x[10] - x[7], w1y = m1 * x[11] - x[8], w2x = m2 * x[12] - x[7], w2y = m2 * x[13] - x[8];
    const double tau0 = r1x * u[1] - r1y * u[0] + r2x * u[3] - r2y * u[2];
    const double tauv = w1x * u[1] - w1y * u[0] + w2x * u[3] - w2y * u[2];
    const double taua = C.g * ((1.0 - m1) * u[0] + (1.0 - m2) * u[2]);
    xn[0] = x[0] + h * x[7] + hh2 * abx;
    xn[1] = x[1] + h * x[8] + hh2 * aby;
    xn[2] = x[2] + h * x[9] + C.iIb * (hh2 * tau0 + h3_6 * tauv + h4_24 * taua);
    xn[3] = x[3] + m1 * (h * x[10]) + hh2 * a1x;
    xn[4] = x[4] + m1 * (h * x[11]) + hh2 * a1y;
    xn[5] = x[5] + m2 * (h * x[12]) + hh2 * a2x;
    xn[6] = x[6] + m2 * (h * x[13]) + hh2 * a2y;
    xn[7] = x[7] + h * abx;
    xn[8] = x[8] + h * aby;
    xn[9] = x[9] + C.iIb * (h * tau0 + hh2 * tauv + h3_6 * taua);
    xn[10] = x[10] + h * a1x;
    xn[11] = x[11] + h * a1y;
    xn[12] = x[12] + h * a2x;
    xn[13] = x[13] + h * a2y;
    xn[14] = x[14] + h;
    if (jump) {
        xn[4] = 0.0;
        xn[6] = 0.0;
        xn[10] = xn[11] = xn[12] = xn[13] = 0.0;
    }
}

// Structure of a step block that the Riccati sweep relies on (checked against the union pattern of qln_device.h):
//   d x+/d x (15x15): the diagonal, rows 2 (theta) and 9 (omega), and (c-7, c) for c in {7, 8, 10, 11, 12, 13}
//                     (a position picks up h times its velocity);
//   d x+/d F (columns 15-18): six rows each -- b_rows(j);   d x+/d h (column 19): dense.
__host__ __device__ constexpr int a_coupling(int c) { return (c == 7 || c == 8 || (c >= 10 && c <= 13)) ? c - 7 : -1; }
__host__ __device__ constexpr void b_rows(int j, int (&rows)[6]) {
    rows[0] = j & 1;
    rows[1] = 2;
    rows[2] = 3 + j;
    rows[3] = 7 + (j & 1);
    rows[4] = 9;
    rows[5] = 10 + j;
}
constexpr bool step_structure_ok() {
    for (int r = 0; r < 15; ++r) {
        for (int c = 0; c < 15; ++c)
            if (step_union_present(r, c) && !(r == c || r == 2 || r == 9 || r == a_coupling(c))) return false;
        for (int j = 0; j < 4; ++j) {
            int rows[6] = {0, 0, 0, 0, 0, 0};
            b_rows(j, rows);
            bool in = false;
            for (int q = 0; q < 6; ++q) in = in || rows[q] == r;
            if (step_union_present(r, 15 + j) && !in) return false;
        }
    }
    return true;
}
static_assert(step_structure_ok(), "the sparse products of the Riccati sweep cover every possible non-zero of a step block");
static_assert(!step_union_present(1, 15), "the zero slot of the B image is outside the union pattern");

struct Lds {
    double *X, *U, *K, *leq, *kn, *P, *A, *B, *T, *S, *Qxx, *Qux, *Quu, *g, *Hd;
    int* map;  // union-pattern position -> offset in [A | B]
};

// One statement of the LDS layout: the kernel carves its pointers out of the dynamic LDS with it (returned BY VALUE --
// a struct whose address is taken would live in scratch memory, and every LDS access would start with a scratch load),
// the host gets the size (base = null).
struct Carved {
    Lds L;
    size_t doubles;
};
__host__ __device__ __forceinline__ Carved carve(double* base, int N) {
    size_t off = 0;
    Carved cv;
    Lds& t = cv.L;
#define QLN_TAKE(field, n)                 \
    t.field = base ? base + off : nullptr; \
    off += (size_t)(((n) + 1) & ~1);
    // the sweep's matrices first (compile-time offsets), row stride 16 -- see the sweep
    QLN_TAKE(P, 15 * kLd)
    QLN_TAKE(A, 300)  // the table AC[16][4] of A's entries, then B (15x5); the rest belongs to the roll-outs' slots
    t.B = t.A ? t.A + kAcB : nullptr;
    QLN_TAKE(T, 15 * kLd)  // [T | pv]
    QLN_TAKE(Qxx, 15 * kLd)  // [Qxx | Qx]     (P .. Hd: contiguous, the roll-outs' (x, u) slots)
    QLN_TAKE(S, 5 * kLd)
    t.K = t.S;  // gains of the knot being swept, 5 rows of 16 (S is dead once Quu is formed); all knots: global scratch
    QLN_TAKE(Qux, 5 * kLd)   // [Qux | Qu]
    QLN_TAKE(Quu, 26)
    QLN_TAKE(g, 20)
    QLN_TAKE(Hd, 20)
    QLN_TAKE(X, 15 * N)
    QLN_TAKE(U, 5 * N)
    QLN_TAKE(leq, 16)
    QLN_TAKE(kn, kKn * N)
    double* mp = base ? base + off : nullptr;
    off += (size_t)((kStepUnion / 2 + 2 + 1) & ~1);
    t.map = reinterpret_cast<int*>(mp);
#undef QLN_TAKE
    cv.doubles = off;
    return cv;
}

}  // namespace

size_t ilqr_lds_bytes(int32_t N) { return carve(nullptr, N).doubles * sizeof(double); }
// per problem: the step blocks of every knot, and one trial trajectory per step length
size_t ilqr_scratch_doubles(int32_t B, int32_t N) { return (size_t)B * ((size_t)N * (kEnt + kKg + kIneq) + (size_t)kAlphas * 20 * (size_t)N); }

namespace {

// info[b][16] (15: 1 if the rescue phase ran): 0 outer iterations, 1 iLQR iterations, 2 objective f of the returned Z, 3 constraint violation (the
// solver's own measure: c rows it penalises + the bounds it penalises), 4 final rho, 5 status (0 = converged to tol,
// 1 = outer limit reached, 2 = no descent step found at the last penalty), 6 augmented cost, 7 last accepted alpha,
// 8 sum of h, 9 LM mu at exit
template <int OCC>
__global__ __launch_bounds__(kWave, OCC) void k_al_ilqr(BatchParams P, SolveParams S, double* __restrict__ Zio,
                                                      double* __restrict__ info, double* __restrict__ scratch) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;  // wave-uniform
    const int N = P.N;
    const ProblemDesc pd = P.desc[b];
    const int kt = pd.k_trans, im = pd.init_mode;
    const double Ib = P.mb * (P.lb * P.lb) / 12;
    const double mbg = P.mb * P.g;
    const double p_lb = P.lb;
    FastStep FS;
    FS.imb = 1.0 / P.mb, FS.imf = 1.0 / P.mf, FS.iIb = 1.0 / Ib, FS.g = P.g;
    const double h_lo = S.h_lo, h_hi = S.h_hi, th_lo = S.th_lo, th_hi = S.th_hi, h_prox = S.h_prox;
    const double mu0 = S.mu0, mu_min = S.mu_min, mu_max = S.mu_max, tol = S.tol, inner_tol = S.inner_tol;
    const double rho_factor = S.rho_factor, rho_max = S.rho_max;
    const int max_outer = S.max_outer, max_inner = S.max_inner;
    const bool q6 = S.q6 != 0, exact_h = S.exact_h != 0;
    const Lds L = carve(lds, N).L;
    double* __restrict__ Zb = Zio + (int64_t)b * P.z_stride;
    double* __restrict__ ent = scratch + (int64_t)b * ((int64_t)N * (kEnt + kKg + kIneq) + (int64_t)kAlphas * 20 * N);
    double* __restrict__ Kg = ent + (int64_t)N * kEnt;    // [N][5][16]: feedback gains of every knot (L2-resident)
    double* __restrict__ traj = Kg + (int64_t)N * kKg;    // [20 N][kAlphas]: the trial roll-outs, entry of Z x step length
                                                          // (the sixteen lanes of a store or load share a 128-byte line)
    double* __restrict__ lamg = traj + (int64_t)kAlphas * 20 * N;  // [N][kIneq]: multipliers of the inequality rows
    const double* __restrict__ costg = P.cost + (P.cost_batch > 1 ? (int64_t)b * N * 41 : 0);  // wave-uniform reads
    const double* __restrict__ x0g = P.bnd + (int64_t)b * 30;

    // ---- load: controls of the initial guess, cost records, boundary states; multipliers start at zero ----
    {
        for (int i = lane; i < 5 * (N - 1); i += kWave) {
            const int k = i / 5, j = i - 5 * k;
            double v = Zb[20 * k + 15 + j];
            if (j == 4) v = fmin(fmax(v, h_lo), h_hi);
            L.U[i] = v;
        }
        for (int i = lane; i < kIneq * N; i += kWave) lamg[i] = 0.0;
        if (lane < 16) L.leq[lane] = 0.0;
        if (lane < 15) L.X[lane] = x0g[lane];
        // where the p-th entry of a step block's union pattern goes in [A | B] (the value expressions of the
        // statements are not expanded here: the macro parameter is unused)
#define JW(row, col, val)                                                                        \
    {                                                                                            \
        constexpr int pos_ = step_union_pos(row, col);                                           \
        if (lane == (pos_ & 63)) L.map[pos_] = ((col) < 15) ? ac_slot(row, col) : kAcB + 5 * (row) + ((col) - 15); \
    }
        QLN_STEP_ENTRIES();
#undef JW
    }
    wave_lds_sync();
    double xf[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) xf[i] = x0g[15 + i];

    auto stage_in = [=](int k, double rho, double w) {
        StageIn I;
        I.rec = costg + 41 * k;
        I.lam5 = lamg + kIneq * k;
        I.rho = rho;
        I.w = w;
        I.has_u = k < N - 1;
        I.ineq = k >= 1;
        I.q6 = q6;
        I.final_ctrl = (k == N - 2);
        I.terminal = (k == N - 1);
        I.lam_fc = L.leq[14];
        I.lam_term = L.leq;
        I.xf = xf;
        I.mbg = mbg;
        I.lb = p_lb;
        I.th_lo = th_lo;
        I.th_hi = th_hi;
        return I;
    };

    // ---- initial roll-out of the guess's controls from x0 (every lane redundantly; lane 0's copy is kept) ----
    {
        double x[15], u[5], xn[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) x[i] = L.X[i];
        for (int k = 0; k < N - 1; ++k) {
#pragma unroll
            for (int j = 0; j < 5; ++j) u[j] = L.U[5 * k + j];
            step_forward(P, k, kt, im, Ib, x, u, xn);
#pragma unroll
            for (int i = 0; i < 15; ++i) x[i] = xn[i];
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 15; ++i) L.X[15 * (k + 1) + i] = xn[i];
            }
        }
    }
    wave_lds_sync();

    // where a problem's time goes: s_memtime ticks per phase, reported in info[10..14]
    unsigned long long tk_refresh = 0, tk_blocks = 0, tk_sweep = 0, tk_roll = 0, tk_accept = 0, tk0;
#define QLN_TICK(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tk0; tk0 = t_; } while (0)
#ifdef QLN_ROLL_STAMPS  // tuning build only: info[10..14] = law + u / slots + stores + step / barrier / merit / (unused) of a roll-out knot
    unsigned long long rl0 = 0;
#define QLN_ROLL_BEGIN() rl0 = __builtin_amdgcn_s_memtime()
#define QLN_ROLL_TICK(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - rl0; rl0 = t_; } while (0)
#else
#define QLN_ROLL_BEGIN() do {} while (0)
#define QLN_ROLL_TICK(acc) do {} while (0)
#endif
#ifdef QLN_SWEEP_STAMPS  // tuning build only: info[10..14] = the five phases of a sweep knot instead of the five phases of an iteration
    unsigned long long sw0 = 0;
#define QLN_SWEEP_BEGIN() sw0 = __builtin_amdgcn_s_memtime()
#define QLN_SWEEP_TICK(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - sw0; sw0 = t_; } while (0)
#define QLN_ITER_TICK(acc) do { tk0 = __builtin_amdgcn_s_memtime(); } while (0)
#elif defined(QLN_ROLL_STAMPS)
#define QLN_SWEEP_BEGIN() do {} while (0)
#define QLN_SWEEP_TICK(acc) do {} while (0)
#define QLN_ITER_TICK(acc) do { tk0 = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QLN_SWEEP_BEGIN() do {} while (0)
#define QLN_SWEEP_TICK(acc) do {} while (0)
#define QLN_ITER_TICK(acc) QLN_TICK(acc)
#endif
    double rho = S.rho0, mu = mu0;
    double prev_viol = INFINITY;
    double J_cur = 0.0, viol = INFINITY, last_alpha = 0.0;
    int outer = 0, iters = 0, status = 1;

    // Per-knot scalars at the current trajectory (lane = knot): weights, active multipliers, clearance slope, l_k.
    // Returns the augmented cost J and the violation (wave-uniform).
    auto refresh = [=](double rho, double& J, double& vmax) {
        double Jl = 0.0, vl = 0.0;
        for (int k0 = 0; k0 < N; k0 += kWave) {
            const int k = k0 + lane;
            if (k < N) {
                double x[15], u[5] = {0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 15; ++i) x[i] = L.X[15 * k + i];
                if (k < N - 1) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) u[j] = L.U[5 * k + j];
                }
                const double w = (k < N - 1) ? u[4] : 1.0;
                StageIn I = stage_in(k, rho, w);
                StageOut o;
                stage_eval(I, x, u, o);
                double* kn = L.kn + kKn * k;
                kn[KN_W] = w;
#pragma unroll
                for (int j = 0; j < kIneq; ++j) {
                    kn[KN_T0 + j] = o.t[j];
                }
                kn[KN_CQ] = o.cq;
                kn[KN_ELL] = o.ell;
                if (k == N - 2) L.leq[15] = L.leq[14] + rho * o.e_fc;
                Jl += o.val;
                vl = fmax(vl, o.viol);
            }
        }
        J = wsum(Jl);
        vmax = wmax(vl);
        wave_lds_sync();
    };

    refresh(rho, J_cur, viol);
    // Second wind: a problem that has not reached the tolerance when the schedule runs out (about 3 in 100 000 random
    // landings) gets `rescue_outer` more multiplier updates with accurate inner solves (60 iterations) from a moderate
    // penalty -- the multipliers are nearly right by then, it is the inexact inner solves at a saturated penalty that stall.
    const int total_outer = max_outer + S.rescue_outer;
    int rescued = 0;
    for (outer = 0; outer < total_outer; ++outer) {
        const bool rescue = outer >= max_outer;
        if (outer == max_outer) {
            rho = fmin(rho, 1e6);
            prev_viol = INFINITY;
            rescued = 1;
        }
        const int inner_cap = rescue ? 60 : max_inner;
        mu = mu0;
        bool stalled = false;
        for (int it = 0; it < inner_cap; ++it) {
            tk0 = __builtin_amdgcn_s_memtime();
            refresh(rho, J_cur, viol);
            ++iters;
            QLN_ITER_TICK(tk_refresh);
            // ---- step blocks of every knot (lane = knot), closed form, to the scratch ----
            for (int k0 = 0; k0 < N - 1; k0 += kWave) {
                const int k = k0 + lane;
                if (k < N - 1) {
                    const double* zk = L.X + 15 * k;
                    double x[14];
#pragma unroll
                    for (int i = 0; i < 14; ++i) x[i] = zk[i];
                    const double F1x = L.U[5 * k], F1y = L.U[5 * k + 1], F2x = L.U[5 * k + 2], F2y = L.U[5 * k + 3], h = L.U[5 * k + 4];
                    const int K = k + 1;
                    const int mode = (K <= kt - 1) ? im : 3;
                    const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
                    const double g = P.g, mb = P.mb, mf = P.mf;
                    QLN_STEP_BASE();
                    double* e = ent + (int64_t)k * kEnt;
#define JW(row, col, val)                              \
    {                                                  \
        constexpr int pos_ = step_union_pos(row, col); \
        e[pos_] = (val);                               \
    }
                    QLN_STEP_ENTRIES();
#undef JW
                }
            }
            __threadfence();  // the entries are read back by other lanes of this wave
            wave_lds_sync();
            QLN_ITER_TICK(tk_blocks);

            // ---- backward Riccati sweep ----
            // LDS matrices of the sweep have a row stride of 16: an entry index splits into (row, column) by a shift and a
            // mask, a lane keeps its column over the passes of a phase (what depends on the column only is read once),
            // and the sixteenth column carries the vector that goes through the same product: T[:,15] = pv (so that
            // [Qxx | Qx] = [Hxx | gx] + A'[T | pv] and [Qux | Qu] = [0 | gu] + B'[T | pv] are one loop each).
            bool pd_ok = true;
            {
                // terminal knot: P = Hxx(N-1), pv = gx(N-1)
                const int k = N - 1;
                const double* kn = L.kn + kKn * k;
                const double* rec = costg + 41 * k;
                for (int e = lane; e < 15 * kLd; e += kWave) L.P[e] = 0.0;
                wave_lds_sync();
                if (lane < 15) {
                    const int i = lane;
                    const double xi = L.X[15 * k + i];
                    double gi = rec[i] * xi + rec[20 + i];
                    double hi = rec[i];
                    if (i < 14) {
                        gi += L.leq[i] + rho * (xi - xf[i]);
                        hi += rho;
                    }
                    if (i == 1) {
                        gi += -kn[KN_T0 + 0] - kn[KN_T0 + 1] - kn[KN_T0 + 4];
                        hi += rho * (act(kn[KN_T0 + 0]) + act(kn[KN_T0 + 1]) + act(kn[KN_T0 + 4]));
                    }
                    if (i == 2) {
                        gi += (kn[KN_T0 + 0] - kn[KN_T0 + 1]) * kn[KN_CQ] + kn[KN_T0 + 2] - kn[KN_T0 + 3];
                        hi += rho * ((act(kn[KN_T0 + 0]) + act(kn[KN_T0 + 1])) * kn[KN_CQ] * kn[KN_CQ] + act(kn[KN_T0 + 2]) + act(kn[KN_T0 + 3]));
                    }
                    if (i == 3) {
                        gi += -kn[KN_T0 + 5];
                        hi += rho * act(kn[KN_T0 + 5]);
                    }
                    L.T[kLd * i + 15] = gi;
                    L.P[(kLd + 1) * i] = hi;
                }
                if (lane == 0) {
                    const double off = -rho * (act(kn[KN_T0 + 0]) - act(kn[KN_T0 + 1])) * kn[KN_CQ];
                    L.P[kLd * 1 + 2] = off;
                    L.P[kLd * 2 + 1] = off;
                }
                wave_lds_sync();
            }
            // A and B hold zeros wherever the union pattern has no entry: written once, the per-knot scatter below touches
            // exactly the pattern's 85 positions.  What a knot needs from global memory -- its step-block entries and its
            // cost record -- is requested two knots ahead (it is staged one knot ahead, below), so that the L2 round trip is off
            // the sweep's dependency chain.
            for (int e = lane; e < 300; e += kWave) L.A[e] = 0.0;
            double pf_e0, pf_e1, pf_D, pf_d;
            auto prefetch = [=](int k, double& e0, double& e1, double& cD, double& cd) {
                const double* e = ent + (int64_t)k * kEnt;
                e0 = e[lane];
                e1 = e[min(lane + 64, kStepUnion - 1)];
                const double* rec = costg + 41 * k;
                cD = rec[min(lane, 19)];
                cd = rec[20 + min(lane, 19)];
            };
            prefetch(N - 2, pf_e0, pf_e1, pf_D, pf_d);
            // where this lane's entries of a step block go in [A | B] (constant over the knots)
            const int m0 = L.map[lane], m1 = L.map[min(lane + 64, kStepUnion - 1)];
            // What knot k needs in LDS before its products start: A (15x15), B (15x5) scattered from the prefetched entries, the
            // stage gradient (20) and the diagonal of the Gauss-Newton Hessian.  None of it depends on the knot behind, so it
            // is staged one knot AHEAD -- for knot k-1 inside the phase in which knot k factors Quu -- and has no barrier and no
            // LDS round trip of its own on the sweep's critical path.  No branches around LDS reads: the knot's scalars are
            // read by every lane (one address), the three special entries are selects.
            auto stage_knot = [=](int k, int ln, double v0, double v1, double recD, double recd) {
                L.A[m0] = v0;
                L.A[m1] = v1;  // lanes past the end of the pattern repeat its last entry: the same value to the same address
                const double* kn = L.kn + kKn * k;
                const double w = kn[KN_W];
                const double t0 = kn[KN_T0 + 0], t1 = kn[KN_T0 + 1], t2 = kn[KN_T0 + 2], t3 = kn[KN_T0 + 3], t4 = kn[KN_T0 + 4],
                             t5 = kn[KN_T0 + 5], cq = kn[KN_CQ], ell = kn[KN_ELL];
                const double lfc = L.leq[15];
                const int i = min(ln, 19);
                const double zi = *((i < 15) ? L.X + 15 * k + i : L.U + 5 * k + (i - 15));
                double gi = w * (recD * zi + recd);
                double hi = w * recD;
                // selects, not branches: the block stays straight-line code that the scheduler can weave into the factorisation
                const bool live = k >= 1;  // x_1 = x0 is data: its inequality rows are not live
                const double g1 = gi + (-t0 - t1 - t4), h1 = hi + rho * (act(t0) + act(t1) + act(t4));
                const double g2 = gi + ((t0 - t1) * cq + t2 - t3), h2 = hi + rho * ((act(t0) + act(t1)) * cq * cq + act(t2) + act(t3));
                const double g3 = gi + -t5, h3 = hi + rho * act(t5);
                gi = (live && i == 1) ? g1 : (live && i == 2) ? g2 : (live && i == 3) ? g3 : gi;
                hi = (live && i == 1) ? h1 : (live && i == 2) ? h2 : (live && i == 3) ? h3 : hi;
                const bool fc = (k == N - 2) && (i == 16 || i == 18);
                gi = fc ? gi + lfc : gi;
                hi = fc ? hi + rho : hi;
                gi = (exact_h && i == 19) ? gi + ell : gi;  // d(h l)/dh, the term grad_f! leaves out (quirk Q2)
                // lanes 20 .. 63 have nothing to store: they write into the body of T, which is dead between the products that read
                // it and the next knot's T = P A (the vector in its sixteenth column is not touched)
                *((ln < 20) ? L.g + i : L.T + (ln & 7)) = gi;
                *((ln < 20) ? L.Hd + i : L.T + kLd + (ln & 7)) = hi;
            };
            stage_knot(N - 2, lane, pf_e0, pf_e1, pf_D, pf_d);
            if (N >= 3) prefetch(N - 3, pf_e0, pf_e1, pf_D, pf_d);
            wave_lds_sync();
            for (int k = N - 2; k >= 0 && pd_ok; --k) {
                // the lane index, opaque to loop-invariant code motion in the two-waves-per-SIMD build: the LDS addresses of the
                // phases below are a few integer operations each; hoisted out of the knot loop they would occupy (and spill)
                // registers there.  The one-wave-per-SIMD build has 512 registers and lets them be hoisted: its sweep is a
                // fifth shorter for it (310 k against 377 k cycles of 39 knots).
                int ln = lane;
#ifndef QLN_HOIST_OCC2  // tuning build only
                if constexpr (OCC == 2) asm volatile("" : "+v"(ln));
#endif
                QLN_SWEEP_BEGIN();
                const double* kn = L.kn + kKn * k;
                // knot k-1's entries and cost record (requested a knot ago) are staged during this knot's factorisation; knot
                // k-2's are requested now
                const double v0 = pf_e0, v1 = pf_e1, recD = pf_D, recd = pf_d;
                if (k > 1) prefetch(k - 2, pf_e0, pf_e1, pf_D, pf_d);
                QLN_SWEEP_TICK(tk_refresh);
                const double h12 = (k >= 1) ? -rho * (act(kn[KN_T0 + 0]) - act(kn[KN_T0 + 1])) * kn[KN_CQ] : 0.0;  // d2/d(yb)d(theta)
                const double hfc = (k == N - 2) ? rho : 0.0;                              // d2/d(F1y)d(F2y)
                // The products of the sweep use the structure of the step blocks instead of dense 15-term sums (see
                // step_structure_ok above): A = diagonal + rows 2 (theta) and 9 (omega) + the six position <- velocity
                // couplings (c-7, c); a force column of B has six rows; only the h column of B is dense -- its 15-term
                // sums are split over the four lanes of a quad (terms i = p, p+4, p+8, p+12) and added up by DPP.
                // In the one-wave-per-SIMD build every phase below does ALL its reads and arithmetic first and its (predicated) stores
                // last: a predicated store ends a basic block, and hipcc does not move the reads of one product above the stores
                // of another -- written product by product, a phase is a chain of read / wait / store segments.  The
                // two-waves-per-SIMD build keeps the product-by-product order: with all reads of a phase in flight its 256
                // registers overflow and the spills land in the roll-out loop (measured: sweep -9 %, roll-outs +15..48 %).
                constexpr bool kStoresLast = (OCC == 1);
                // ---- T = P A (entry (r, c): ln keeps c, r = (ln >> 4) + 4 pass), S = P B ----
                {
                    const int c = ln & 15, cp = a_coupling(c);
                    const bool cv = c < 15;
                    const int cc = cv ? c : 0;
                    const int cq = max(cp, 0);
                    const double* acp = L.A + 4 * c;  // (row 15 of the table is zero)
                    const double a0 = acp[0], a2 = acp[1], a9 = acp[2], ac = acp[3];
                    // the four passes are independent: every lane loads (clamped row / column, always inside P) and only the
                    // stores are predicated, so that the LDS reads of all passes are in flight together instead of one
                    // exec-masked pass (and its waits) after the other
                    double tv[4];
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const double* pr = L.P + kLd * min((ln >> 4) + 4 * it, 14);
                        tv[it] = fma(pr[cq], ac, fma(pr[9], a9, fma(pr[2], a2, pr[cc] * a0)));
                    }
                    auto store_T = [&]() {
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = (ln >> 4) + 4 * it;
                            if (r < 15 && cv) L.T[kLd * r + c] = tv[it];
                        }
                    };
                    if constexpr (!kStoresLast) store_T();
                    const int rs = min(ln >> 2, 14), js = ln & 3;
                    const double* ps = L.P + kLd * rs;
                    int rows[6];
                    b_rows(js, rows);
                    double accS = 0.0, ahS = 0.0;
#pragma unroll
                    for (int q = 0; q < 6; ++q) accS = fma(ps[rows[q]], L.B[5 * rows[q] + js], accS);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = js + 4 * t, ic = min(i, 14);
                        const double bv = L.B[(i < 15) ? 5 * i + 4 : kBZero];
                        ahS = fma(ps[ic], bv, ahS);
                    }
                    ahS = quad_sum(ahS);
                    if constexpr (kStoresLast) store_T();
                    if (ln < 60) {
                        L.S[5 * rs + js] = accS;
                        if (js == 0) L.S[5 * rs + 4] = ahS;
                    }
                }
                wave_lds_sync();
                QLN_SWEEP_TICK(tk_blocks);
                {
                    // ---- [Qxx | Qx] = [Hxx | gx] + A'[T | pv] ----
                    const int c = ln & 15;
                    const double t2 = L.T[2 * kLd + c], t9 = L.T[9 * kLd + c];
                    double qv[4];  // as above: loads of all four passes unconditional (row clamped), stores predicated
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = min((ln >> 4) + 4 * it, 14);
                        const int rp = a_coupling(r), rq = max(rp, 0);
                        const double* arp = L.A + 4 * r;
                        const double a0 = arp[0], a2 = arp[1], a9 = arp[2], ac = arp[3];
                        const double tr = L.T[kLd * r + c], tq = L.T[kLd * rq + c];
                        double acc = *((c == 15) ? L.g + r : (r == c) ? L.Hd + r : L.A + kAZero);
                        if ((r == 1 && c == 2) || (r == 2 && c == 1)) acc += h12;
                        acc = fma(a0, tr, acc);
                        acc = fma(a2, t2, acc);
                        acc = fma(a9, t9, acc);
                        acc = fma(ac, tq, acc);
                        qv[it] = acc;
                    }
                    auto store_Qxx = [&]() {
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = (ln >> 4) + 4 * it;
                            if (r < 15) L.Qxx[kLd * r + c] = qv[it];
                        }
                    };
                    if constexpr (!kStoresLast) store_Qxx();
                    // ---- [Qux | Qu] = [0 | gu] + B'[T | pv]: force rows (ln = 16 j + c), then the h row in quads ----
                    const int jf = (ln >> 4);
                    int rowsf[6];
                    b_rows(jf, rowsf);
                    double quf = *((c == 15) ? L.g + 15 + jf : L.A + kAZero);
#pragma unroll
                    for (int q = 0; q < 6; ++q) quf = fma(L.B[5 * rowsf[q] + jf], L.T[kLd * rowsf[q] + c], quf);
                    if constexpr (!kStoresLast) L.Qux[kLd * jf + c] = quf;
                    const int ch = ln >> 2, p = ln & 3;
                    double quh = 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = p + 4 * t, ic = min(i, 14);
                        const double bv = L.B[(i < 15) ? 5 * i + 4 : kBZero];
                        quh = fma(bv, L.T[kLd * ic + ch], quh);
                    }
                    quh = quad_sum(quh) + *((ch == 15) ? L.g + 19 : L.A + kAZero);
                    if constexpr (!kStoresLast) {
                        if (p == 0) L.Qux[4 * kLd + ch] = quh;
                    }
                    // ---- Quu = Huu + B'S + mu I: the force rows (and, by symmetry, the h row), then (h, h) in a quad.  Every lane
                    // runs both parts on clamped indices; lanes 0-31 keep the force rows, the quad of lanes 32-35 keeps (h, h)
                    const int ju = (ln >> 3) & 3, cu = min(ln & 7, 4);
                    int rowsu[6];
                    b_rows(ju, rowsu);
                    const double hdj = L.Hd[15 + ju], hd19 = L.Hd[19];
                    double quu = (ju == cu) ? hdj + mu : 0.0;
                    if ((ju == 1 && cu == 3) || (ju == 3 && cu == 1)) quu += hfc;
#pragma unroll
                    for (int q = 0; q < 6; ++q) quu = fma(L.B[5 * rowsu[q] + ju], L.S[5 * rowsu[q] + cu], quu);
                    double ahu = 0.0;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int i = p + 4 * t, ic = min(i, 14);
                        const double bv = L.B[(i < 15) ? 5 * i + 4 : kBZero];
                        ahu = fma(bv, L.S[5 * ic + 4], ahu);
                    }
                    ahu = quad_sum(ahu);
                    // ---- the stores ----
                    if constexpr (kStoresLast) {
                        store_Qxx();
                        L.Qux[kLd * jf + c] = quf;
                        if (p == 0) L.Qux[4 * kLd + ch] = quh;
                    }
                    if (ln < 32 && (ln & 7) < 5) {
                        L.Quu[5 * ju + cu] = quu;
                        if (cu == 4) L.Quu[20 + ju] = quu;
                    }
                    if (ln == 32) L.Quu[24] = ahu + hd19 + mu + h_prox;  // + the proximal weight on the step length (see SolveParams)
                }
                wave_lds_sync();
                QLN_SWEEP_TICK(tk_sweep);
                // knot k-1 is staged here, in the shadow of the factorisation below (see stage_knot): A, B, g, Hd were last read
                // before the barrier above and are next read after the one that ends this knot
                stage_knot(max(k - 1, 0), ln, v0, v1, recD, recd);  // (knot 0 is staged twice, with its own data: harmless)
                // LDL' of Quu (5x5, h last so that the leading 4x4 factor serves the clamped case), every ln alike
                double q[5][5], l[5][5], dinv[5];
#pragma unroll
                for (int r = 0; r < 5; ++r)
#pragma unroll
                    for (int c = 0; c < 5; ++c) q[r][c] = L.Quu[5 * r + c];
                // right-hand sides, one per lane: lane j < 15 column j of Qux (its solution is column j of the gains), lanes
                // 15 .. 63 Qu (the feed-forward) -- ONE solve serves both
                double y[5], qu[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    y[i] = -L.Qux[kLd * i + min(ln, 15)];
                    qu[i] = L.Qux[kLd * i + 15];
                }
                const double hk = L.U[5 * k + 4];
                // v[i][j] = l[i][j] d[j] is kept beside l[i][j]: a term of the factorisation is then one fused multiply-add
                // (l_im v_jm), and 1/d_j is the hardware reciprocal refined by two Newton steps (d_j > 1e-300 is checked:
                // none of the scaling an IEEE division carries is needed) -- a third of the instructions of the plain form on
                // the longest dependent chain of a knot.  (Explicit fma: the library is built with -ffp-contract=off for the
                // evaluator's sake.)
                double vv[5][5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    double dj = q[j][j];
#pragma unroll
                    for (int m = 0; m < j; ++m) dj = fma(-l[j][m], vv[j][m], dj);
                    if (!(dj > 1e-300)) pd_ok = false;
                    double rj = __builtin_amdgcn_rcp(dj);
                    rj = fma(fma(-dj, rj, 1.0), rj, rj);
                    rj = fma(fma(-dj, rj, 1.0), rj, rj);
                    dinv[j] = rj;
#pragma unroll
                    for (int i = j + 1; i < 5; ++i) {
                        double v = q[i][j];
#pragma unroll
                        for (int m = 0; m < j; ++m) v = fma(-l[i][m], vv[j][m], v);
                        vv[i][j] = v;
                        l[i][j] = v * rj;
                    }
                }
                if (!pd_ok) break;  // wave-uniform: every ln factors the same matrix
                // solve Quu y = rhs for the first n unknowns (n = 5, or 4 with h clamped)
                auto ldl_solve = [&](double (&y)[5], int n) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        if (i < n) {
#pragma unroll
                            for (int m = 0; m < i; ++m) y[i] = fma(-l[i][m], y[m], y[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 5; ++i)
                        if (i < n) y[i] *= dinv[i];
#pragma unroll
                    for (int i = 4; i >= 0; --i) {
                        if (i < n) {
#pragma unroll
                            for (int m = i + 1; m < 5; ++m)
                                if (m < n) y[i] = fma(-l[m][i], y[m], y[i]);
                        }
                    }
                };
                ldl_solve(y, 5);
                // the feed-forward is lane 15's solution (wave-uniform from here on); the box on h: clamp, then the free 4x4
                // is solved again -- by every lane for its own right-hand side
                double dff[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) dff[i] = read_lane(y[i], 15);
                const double lo = h_lo - hk, hi = h_hi - hk;
                const bool clamped = (dff[4] < lo) || (dff[4] > hi);
                if (clamped) {
                    const double hc = fmin(fmax(dff[4], lo), hi);
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (ln < 15) ? -L.Qux[kLd * i + min(ln, 14)] : -(qu[i] + q[i][4] * hc);
                    y[4] = 0.0;
                    ldl_solve(y, 4);
                    y[4] = 0.0;  // gains: the row of a clamped h is zero
#pragma unroll
                    for (int i = 0; i < 4; ++i) dff[i] = read_lane(y[i], 15);
                    dff[4] = hc;
                }
                // gains: ln j < 15 holds column j
                if (ln < 15) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) L.K[kLd * i + ln] = y[i];
                }
                if (ln < 16) {  // row i of the knot's record: 15 gains, then the feed-forward d_i
#pragma unroll
                    for (int i = 0; i < 5; ++i) Kg[(int64_t)kKg * k + 16 * i + ln] = (ln < 15) ? y[i] : dff[i];
                }
                wave_lds_sync();
                QLN_SWEEP_TICK(tk_roll);
                // Value function of the knot: with K = -Quu_ff^-1 Qux_f on the free controls (zero rows for a clamped h) and
                // d the matching feed-forward, K'(Quu K + Qux) and K'(Quu d + Qu) vanish identically, so
                //     P <- sym(Qxx + Qux' K),   pv <- Qx + Qux' d
                // -- half the products of the symmetric three-term form (measured: the same iterates to the last digit of the
                // iteration counts on 16 384 problems).  Entry (r, c) is the mean of the expression and of its transpose,
                // formed by the same ln and written straight into P (nothing of this phase reads P).
                {
                    const int c = ln & 15;
                    const bool cv = c < 15;
                    const int cc = cv ? c : 0;
                    double qc[5], kcol[5];
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        qc[i] = L.Qux[kLd * i + cc];
                        kcol[i] = L.K[kLd * i + cc];
                    }
                    double pv4[4];  // loads of the four passes unconditional (row clamped), stores predicated
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int r = min((ln >> 4) + 4 * it, 14);
                        double acc = L.Qxx[kLd * r + cc], act_ = L.Qxx[kLd * cc + r];
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            acc = fma(L.Qux[kLd * i + r], kcol[i], acc);
                            act_ = fma(qc[i], L.K[kLd * i + r], act_);
                        }
                        pv4[it] = 0.5 * (acc + act_);
                    }
                    auto store_P = [&]() {
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int r = (ln >> 4) + 4 * it;
                            if (r < 15 && cv) L.P[kLd * r + c] = pv4[it];
                        }
                    };
                    if constexpr (!kStoresLast) store_P();
                    const int lv = min(ln, 14);
                    double pvn = L.Qxx[kLd * lv + 15];
#pragma unroll
                    for (int i = 0; i < 5; ++i) pvn = fma(L.Qux[kLd * i + lv], dff[i], pvn);
                    if constexpr (kStoresLast) store_P();
                    if (ln < 15) L.T[kLd * ln + 15] = pvn;
                }
                wave_lds_sync();
                QLN_SWEEP_TICK(tk_accept);
            }
            if (!pd_ok) {
                mu = fmin(mu * 10.0, mu_max);
                if (mu >= mu_max) {
                    stalled = true;
                    break;
                }
                continue;
            }

            QLN_ITER_TICK(tk_sweep);
            __threadfence();  // the gains written during the sweep are read back by the roll-out lanes
            // ---- forward: one closed-loop roll-out per step length, lane a < 16 tries alpha = 2^-a and keeps its
            //      trajectory in the scratch, so that the accepted one need not be rolled out again.  Only the
            //      state recursion is serial.  A knot's feedback law lies across the sixteen lanes' registers and reaches a
            //      lane by DPP row broadcast (below); the merit of the trial trajectories is evaluated four knots at a time, a
            //      lane group per knot, from (x_k, u_k) the roll-out lanes leave in LDS (the sweep's matrices are free now) --
            //      nothing of a trial trajectory is read back from memory but the accepted one.
            double J_try = INFINITY;
            {
                // merit: a batch of kGroups knots at a time, one lane group per knot -- the roll-out lanes are one of the groups (they
                // would idle while the others evaluate; a batch of four instead of three is a quarter fewer evaluations)
                constexpr int kGroups = kPerTraj;
                static_assert(kGroups >= 1 && kGroups * kAlphas * 20 <= 3 * ((15 * kLd + 1) & ~1) + 300 + 2 * ((5 * kLd + 1) & ~1) + 26 + 20 + 20,
                              "the (x, u) slots live in the sweep's matrices, P .. Hd (contiguous in carve())");
                double* const slots = L.P;  // [kGroups][kAlphas][20]
                const int a = lane & (kAlphas - 1), grp = lane / kAlphas;  // lanes 0 .. kAlphas-1 also roll out
                const double alpha = ldexp(1.0, -a * kAlphaStep);
                double* __restrict__ tz = traj + a;
                double x[15], u[5], xn[15];
                double J = 0.0;
#pragma unroll
                for (int i = 0; i < 15; ++i) x[i] = L.X[i];
                // The feedback law of a knot -- 5 rows of [15 gains, feed-forward] -- is held across a row of sixteen lanes, one
                // register per row of the law: lane i of the roll-out lanes' row has K[j][i] (i = 15: d_j).  A roll-out lane gets
                // the entry it needs by a DPP row broadcast (one v_mov_b64_dpp each): no LDS tile for the gains, no batches of
                // LDS reads each waited for in turn -- those were a quarter of a roll-out knot.  The next knot's law is requested
                // a knot ahead (L2-resident scratch).
                static_assert(kAlphas == 16, "the roll-out lanes are one DPP row");
                double Kr[5], Kn[5];
                auto load_gains = [=](int k, double (&K)[5]) {
                    const double* __restrict__ kp = Kg + (int64_t)kKg * min(k, N - 2) + (lane & 15);
#pragma unroll
                    for (int j = 0; j < 5; ++j) K[j] = kp[16 * j];
                };
                load_gains(0, Kr);
                // a helper lane's multipliers are requested a batch ahead (global scratch)
                double lr[kIneq];
                auto fetch_lam = [=](int kk, double (&lo)[kIneq]) {
                    const int kq = min(max(kk, 0), N - 1);
#pragma unroll
                    for (int j = 0; j < kIneq; ++j) lo[j] = lamg[kIneq * kq + j];
                };
                fetch_lam(grp, lr);
                // ... and so is the cost record of the knot a lane group evaluates: across the group's sixteen lanes, three
                // coalesced loads (stage_ell_row), instead of 41 loads per lane at the point of use
                double cr[3];
                auto fetch_rec = [=](int kk, double (&c3)[3]) {
                    const double* __restrict__ r = costg + 41 * min(max(kk, 0), N - 2);
                    c3[0] = r[a];
                    c3[1] = r[16 + a];
                    c3[2] = r[32 + min(a, 8)];
                };
                fetch_rec(grp, cr);
                wave_lds_sync();
                for (int k = 0; k < N - 1; ++k) {
                    QLN_ROLL_BEGIN();
                    load_gains(k + 1, Kn);  // a whole knot of arithmetic between the request and the first use
                    const int slot = k % kGroups;
                    if (lane < kAlphas) {
#pragma unroll
                        for (int j = 0; j < 5; ++j) u[j] = L.U[5 * k + j] + alpha * row_bcast<15>(Kr[j]);
                        static_for(std::make_integer_sequence<int, 15>{}, [&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            const double dx = x[i] - L.X[15 * k + i];
                            fmac5_row_bcast<i, i == 0>(u, Kr, dx);
                        });
                        u[4] = fmin(fmax(u[4], h_lo), h_hi);
                        QLN_ROLL_TICK(tk_refresh);
                        double* sl = slots + (slot * kAlphas + a) * 20;
#pragma unroll
                        for (int i = 0; i < 15; ++i) sl[i] = x[i];
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            sl[15 + j] = u[j];
                            tz[kAlphas * (20 * k + 15 + j)] = u[j];
                        }
                        step_fast(FS, k, kt, im, x, u, xn);
#pragma unroll
                        for (int i = 0; i < 15; ++i) {
                            x[i] = xn[i];
                            tz[kAlphas * (20 * (k + 1) + i)] = xn[i];
                        }
                    }
                    QLN_ROLL_TICK(tk_blocks);
#pragma unroll
                    for (int j = 0; j < 5; ++j) Kr[j] = Kn[j];
                    wave_lds_sync();
                    QLN_ROLL_TICK(tk_sweep);
                    if (slot == kGroups - 1 || k == N - 2) {  // a batch of knots kb .. k is complete (wave-uniform)
                        const int kb = k - slot, kk = kb + grp;
                        if (kk <= k) {
                            const double* sl = slots + (grp * kAlphas + a) * 20;
                            double xs[15], us[5];
#pragma unroll
                            for (int i = 0; i < 15; ++i) xs[i] = sl[i];
#pragma unroll
                            for (int j = 0; j < 5; ++j) us[j] = sl[15 + j];
                            const double w = exact_h ? us[4] : L.kn[kKn * kk + KN_W];
                            StageIn I = stage_in(kk, rho, w);
                            I.lam5 = lr;
                            StageOut o;
                            stage_eval<true>(I, xs, us, o, stage_ell_row(cr, xs, us));
                            J += o.val;
                        }
                        fetch_lam(kb + kGroups + grp, lr);
                        fetch_rec(kb + kGroups + grp, cr);
                    }
                    QLN_ROLL_TICK(tk_roll);
                }
                if (lane < kAlphas) {  // the terminal knot: the roll-out lanes hold x_N
                    double u0[5] = {0, 0, 0, 0, 0};
                    StageIn I = stage_in(N - 1, rho, 1.0);
                    StageOut o;
                    stage_eval(I, x, u0, o);
                    J += o.val;
                }
#pragma unroll
                for (int off = kAlphas; off < kWave; off <<= 1) J += __shfl_xor(J, off, kWave);
                J_try = J;
            }
            QLN_ITER_TICK(tk_roll);
            // the best of the sixteen candidates, if any lowers the cost (all were computed anyway: one lane each)
            double J_best = ((lane < kAlphas) && (J_try == J_try) && (J_try < J_cur)) ? J_try : INFINITY;
            int a_star = lane;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double Jo = __shfl_xor(J_best, off, kWave);
                const int ao = __shfl_xor(a_star, off, kWave);
                if (Jo < J_best || (Jo == J_best && ao < a_star)) {
                    J_best = Jo;
                    a_star = ao;
                }
            }
            if (!(J_best < J_cur)) {  // no step length gives descent: more regularisation
                mu = fmin(mu * 10.0, mu_max);
                if (mu >= mu_max) {
                    stalled = true;
                    break;
                }
                continue;
            }
            const double J_new = J_best;
            last_alpha = ldexp(1.0, -a_star * kAlphaStep);
            // accept: lane a_star's roll-out becomes the current trajectory (x_1 = x0 stays)
            __threadfence();
            wave_lds_sync();
            {
                const double* __restrict__ tz = traj + a_star;
                for (int i = 15 + lane; i < 20 * (N - 1) + 15; i += kWave) {
                    const int k = i / 20, j = i - 20 * k;
                    const double v = __builtin_nontemporal_load(tz + kAlphas * i);
                    if (j < 15) L.X[15 * k + j] = v;
                    else L.U[5 * k + (j - 15)] = v;
                }
            }
            wave_lds_sync();
            QLN_ITER_TICK(tk_accept);
            mu = fmax(mu / 3.0, mu_min);
            const double dJ = J_cur - J_new;
            if (dJ < inner_tol * (1.0 + fabs(J_new))) break;
        }
        // ---- outer: violation, stop test, multipliers, penalty ----
        refresh(rho, J_cur, viol);
        if (viol <= tol) {
            status = 0;
            ++outer;
            break;
        }
        for (int k0 = 0; k0 < N; k0 += kWave) {
            const int k = k0 + lane;
            if (k < N && k >= 1) {
#pragma unroll
                for (int j = 0; j < kIneq; ++j) lamg[kIneq * k + j] = L.kn[kKn * k + KN_T0 + j];  // max(0, lam + rho g)
            }
        }
        if (lane < 14) L.leq[lane] += rho * (L.X[15 * (N - 1) + lane] - xf[lane]);
        if (lane == 14) L.leq[14] = L.leq[15];
        wave_lds_sync();
        if (viol > 0.25 * prev_viol) rho = fmin(rho * rho_factor, rho_max);
        prev_viol = viol;
        if (stalled && rho >= rho_max) status = 2;
    }

    // ---- write the solution (k_exact_rollout then replaces the states by the RK4 roll-out of these controls) and the report ----
    for (int i = lane; i < 20 * (N - 1); i += kWave) {
        const int k = i / 20, j = i - 20 * k;
        Zb[i] = (j < 15) ? L.X[15 * k + j] : L.U[5 * k + (j - 15)];
    }
    if (lane < 15) Zb[20 * (N - 1) + lane] = L.X[15 * (N - 1) + lane];
    {
        double f = 0.0, hs = 0.0;
        for (int k0 = 0; k0 < N; k0 += kWave) {
            const int k = k0 + lane;
            if (k < N) {
                const double ell = L.kn[kKn * k + KN_ELL];
                const double hk = (k < N - 1) ? L.U[5 * k + 4] : 1.0;
                f += hk * ell;
                if (k < N - 1) hs += hk;
            }
        }
        f = wsum(f);
        hs = wsum(hs);
        if (info && lane == 0) {
            double* o = info + 16 * (int64_t)b;
            o[0] = (double)outer;
            o[1] = (double)iters;
            o[2] = f;
            o[3] = viol;
            o[4] = rho;
            o[5] = (double)status;
            o[6] = J_cur;
            o[7] = last_alpha;
            o[8] = hs;
            o[9] = mu;
            o[10] = (double)tk_refresh;
            o[11] = (double)tk_blocks;
            o[12] = (double)tk_sweep;
            o[13] = (double)tk_roll;
            o[14] = (double)tk_accept;
            o[15] = (double)rescued;
        }
    }
}

// The states handed back to the caller: the solution's controls rolled out from x0 once more with the evaluator's RK4 step
// (the iterations above roll out in closed form, 1e-15 away), so that initial-condition, dynamics and contact rows of the
// returned Z are zero to the last bit when the evaluator looks at them.  One thread per problem; a kernel of its own so
// that the solver kernel's register allocation is not disturbed (with this loop inside it the sweep ran 27 % slower).
// The report is restated on THAT trajectory: info[2] = eval_f of the returned Z in the reference's operation order
// (src/costs.jl:6-16: the same bits qln_eval_objective gives), info[3] = the violation of the returned Z -- the rows of
// eval_c! the roll-out does not zero by construction (terminal, final-control, clearance with the reference's kinked
// |sin|, src/constraints.jl:98-113,150,154; contact rows, which only an infeasible x0 can violate) as
// qln_constraint_violation measures them, and solve()'s variable bounds on theta and quirk Q6's two (src/moi.jl:54-65) on
// the knots after the first.  status (info[5]) stays the solver's verdict on its own closed-form roll-out.
__global__ __launch_bounds__(kWave) void k_exact_rollout(BatchParams P, SolveParams S, double* __restrict__ Zio, double* __restrict__ info) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const ProblemDesc pd = P.desc[b];
    const int N = P.N, kt = pd.k_trans, im = pd.init_mode;
    const double Ib = P.mb * (P.lb * P.lb) / 12;
    double* __restrict__ Zb = Zio + (int64_t)b * P.z_stride;
    const double* __restrict__ x0g = P.bnd + (int64_t)b * 30;
    const double* __restrict__ costg = P.cost + (P.cost_batch > 1 ? (int64_t)b * N * 41 : 0);
    double x[15], u[5], xn[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        x[i] = x0g[i];
        Zb[i] = x[i];
    }
    double f = 0.0, viol = 0.0, poison = 0.0;
    auto up = [&](double v) {  // max that does not swallow a NaN (fmax would): a non-finite row poisons the report
        viol = fmax(viol, v);
        poison += 0.0 * v;
    };
    // rows of one knot's state: contact (the foot of init_mode at every knot, the other from k_trans on), clearance, bounds
    auto state_rows = [&](int k, const double (&x)[15]) {
        const double y_init = (im == 1) ? x[4] : x[6], y_other = (im == 1) ? x[6] : x[4];
        up(fabs(y_init));
        if (k + 1 >= kt) up(fabs(y_other));
        up(-(x[1] - (P.lb / 2) * fabs(sin(x[2]))));
        if (k >= 1) {
            up(fmax(x[2] - S.th_hi, S.th_lo - x[2]));
            if (S.q6) up(fmax(-x[1], -x[3]));
        }
    };
    for (int k = 0; k < N - 1; ++k) {
#pragma unroll
        for (int j = 0; j < 5; ++j) u[j] = Zb[20 * k + 15 + j];
        {   // h_k * stagecost(obj[k], x_k, u_k), added in knot order
            const double* __restrict__ rec = costg + 41 * k;
            double a = (0.5 * (rec[0] * x[0])) * x[0], bb = rec[20] * x[0];
#pragma unroll
            for (int i = 1; i < 15; ++i) {
                a = a + (0.5 * (rec[i] * x[i])) * x[i];
                bb = bb + rec[20 + i] * x[i];
            }
            double cc = (0.5 * (rec[15] * u[0])) * u[0], dd = rec[35] * u[0];
#pragma unroll
            for (int j = 1; j < 5; ++j) {
                cc = cc + (0.5 * (rec[15 + j] * u[j])) * u[j];
                dd = dd + rec[35 + j] * u[j];
            }
            f = f + u[4] * ((((a + bb) + cc) + dd) + rec[40]);
        }
        state_rows(k, x);
        if (k == N - 2) up(fabs(u[1] + u[3] + P.mb * P.g));
        step_forward(P, k, kt, im, Ib, x, u, xn);
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            x[i] = xn[i];
            Zb[20 * (k + 1) + i] = xn[i];
        }
    }
    {   // termcost(obj[N], x_N) and the terminal rows
        const double* __restrict__ rec = costg + 41 * (N - 1);
        double a = (0.5 * (rec[0] * x[0])) * x[0], bb = rec[20] * x[0];
#pragma unroll
        for (int i = 1; i < 15; ++i) {
            a = a + (0.5 * (rec[i] * x[i])) * x[i];
            bb = bb + rec[20 + i] * x[i];
        }
        f = f + ((a + bb) + rec[40]);
        state_rows(N - 1, x);
#pragma unroll
        for (int i = 0; i < 14; ++i) up(fabs(x[i] - x0g[15 + i]));
    }
    if (poison != poison) viol = poison;
    if (info) {
        info[16 * (int64_t)b + 2] = f;
        info[16 * (int64_t)b + 3] = viol;
    }
}

}  // namespace

hipError_t launch_al_ilqr(const BatchParams& p, const SolveParams& s, double* Z, double* info, double* scratch, hipStream_t stream) {
    const size_t lds = ilqr_lds_bytes(p.N);
    // Two register budgets: a batch that fills the chip more than once over runs two waves per SIMD (256 registers, a
    // few spills outside the sweep) -- one wave keeps a SIMD's issue slots about 45 % busy, the second one fills them
    // (profiles/r02_solve_occupancy.txt); a batch of at most one wave per SIMD gets the whole register file.
    bool two_per_simd = p.B > 4 * kCUs && 5 * lds <= kLdsPerCU;  // more than 4 waves per CU must fit the LDS
#ifdef QLN_TUNING
    // tuning build only: QLN_ILQR_OCC=1|2 forces the register budget for A/B runs (bench/solve_sweep.py)
    if (const char* e = getenv("QLN_ILQR_OCC")) two_per_simd = atoi(e) == 2;
#endif
    auto go = [&](auto kern) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(xcd_grid(p.B)), dim3(kWave), lds, stream, p, s, Z, info, scratch);
        return hipGetLastError();
    };
    if (hipError_t e2 = two_per_simd ? go(k_al_ilqr<2>) : go(k_al_ilqr<1>); e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(k_exact_rollout, dim3((p.B + kWave - 1) / kWave), dim3(kWave), 0, stream, p, s, Z, info);
    return hipGetLastError();
}

}  // namespace qln
