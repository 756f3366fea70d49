// qln_kernel_common.h -- device helpers shared by the gfx950 kernels (qln_kernels.hip: the evaluator;
// qln_solver_kernels.hip: Jacobian products and the batched Gauss-Newton step).  Internal.
#pragma once

#include "qln_device.h"

namespace qln {
namespace {

constexpr int kWave = 64;
constexpr int kBlk = 300;  // 15 x 20 doubles per step block

// Workgroups are observed to be dealt round-robin over the 8 XCDs (blocks b and b+8 share one;
// MI355X_MICROARCH.md "Workgroup dispatch").  Giving each XCD a CONTIGUOUS range of problems makes
// every XCD's L2/TLB see one sequential write front instead of every 8th 100-KB region: a pure
// fill of the evaluator's store shape goes from ~5.8 to ~6.8 TB/s with this map alone
// (profiles/r01_store_ceiling.txt).  Placement is a speed matter only: any dispatch order computes
// the same results.
__device__ __forceinline__ int xcd_contiguous_index(int block, int n) {
    const int per_xcd = (n + 7) >> 3;
    return (block & 7) * per_xcd + (block >> 3);  // may be >= n for the last blocks: caller checks
}
__host__ inline unsigned xcd_grid(int n) { return 8u * (unsigned)((n + 7) / 8); }

// Every workgroup of the hot kernel is ONE wavefront, so cross-lane hand-offs through LDS need no
// s_barrier: the LDS executes a wave's DS instructions in issue order.  What is needed is that the
// compiler keeps that order; a wavefront-scope fence plus the (instruction-less) wave barrier do
// that.  Unlike __syncthreads() this does not drain vmcnt, so the global store stream of one
// sub-tile keeps flowing while the next one is assembled.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// Value path (shared by the evaluator and by the solver's rollout, which must round identically): literal restatement of contact{1,2,3}_dynamics (src/planar_quadruped.jl:36-185).
// f1free/f2free select the mode: mode 1 = foot 2 free, mode 2 = foot 1 free, mode 3 = none.
// ---------------------------------------------------------------------------------------------
struct StepConst {
    double abx, aby;    // body acceleration
    double a1x, a1y;    // foot 1 acceleration (0 if pinned)
    double a2x, a2y;    // foot 2 acceleration (0 if pinned)
};

__device__ __forceinline__ void dynamics(const double (&s)[14], const double (&u)[5], const StepConst& k, bool f1free,
                                         bool f2free, double Ib, double (&f)[14]) {
    const double F1x = u[0], F1y = u[1], F2x = u[2], F2y = u[3];
    // tauF = -F1x*(p1[2]-pb[2]) + F1y*(p1[1]-pb[1]) - F2x*(p2[2]-pb[2]) + F2y*(p2[1]-pb[1])
    const double tauF = -F1x * (s[4] - s[1]) + F1y * (s[3] - s[0]) - F2x * (s[6] - s[1]) + F2y * (s[5] - s[0]);
    f[0] = s[7];
    f[1] = s[8];
    f[2] = s[9];
    f[3] = f1free ? s[10] : 0.0;
    f[4] = f1free ? s[11] : 0.0;
    f[5] = f2free ? s[12] : 0.0;
    f[6] = f2free ? s[13] : 0.0;
    f[7] = k.abx;
    f[8] = k.aby;
    f[9] = tauF / Ib;
    f[10] = k.a1x;
    f[11] = k.a1y;
    f[12] = k.a2x;
    f[13] = k.a2y;
}

// contact*_dynamics_rk4 (src/planar_quadruped.jl:189-221)
__device__ __forceinline__ void rk4_step(const double (&x)[15], const double (&u)[5], const StepConst& k, bool f1free,
                                         bool f2free, double Ib, double (&xn)[15]) {
    const double h = u[4];
    const double hh = 0.5 * h;
    double s0[14], s[14], f1[14], f2[14], f3[14], f4[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) s0[i] = x[i];
    dynamics(s0, u, k, f1free, f2free, Ib, f1);
#pragma unroll
    for (int i = 0; i < 14; ++i) s[i] = s0[i] + hh * f1[i];
    dynamics(s, u, k, f1free, f2free, Ib, f2);
#pragma unroll
    for (int i = 0; i < 14; ++i) s[i] = s0[i] + hh * f2[i];
    dynamics(s, u, k, f1free, f2free, Ib, f3);
#pragma unroll
    for (int i = 0; i < 14; ++i) s[i] = s0[i] + h * f3[i];
    dynamics(s, u, k, f1free, f2free, Ib, f4);
    const double h6 = h / 6.0;
#pragma unroll
    for (int i = 0; i < 14; ++i) xn[i] = s0[i] + h6 * (((f1[i] + 2 * f2[i]) + 2 * f3[i]) + f4[i]);
    xn[14] = x[14] + u[4];
}

// Base quantities of a step block (closed form in qln_kernels.hip's header) from the knot's state x[0..13], forces
// F1x..F2y, step h, the mode flags f1free / f2free / jump and the model constants g, mb, mf, Ib in scope.  Defines
// everything QLN_STEP_ENTRIES() refers to.  Column 19 (hc[]), the eight force-column entries of the theta/omega rows
// and a dozen scalars are values; the 30 (weight x force) products of the theta/omega rows are formed where an
// entry is used, from wAt / wBt / wAw (one multiply each).
#define QLN_STEP_BASE()                                                                              \
    const double m1 = f1free ? 1.0 : 0.0, m2 = f2free ? 1.0 : 0.0; \
    const double keep = jump ? 0.0 : 1.0; \
    const double km1 = keep * m1, km2 = keep * m2; \
    const double abx = (F1x + F2x) / mb, aby = (F1y + F2y) / mb + g; \
    const double a1x = m1 * (-F1x / mf), a1y = m1 * (-F1y / mf + g); \
    const double a2x = m2 * (-F2x / mf), a2y = m2 * (-F2y / mf + g); \
    const double h2 = h * h, h3 = h2 * h, h4 = h2 * h2; \
    const double iIb = 1.0 / Ib; \
    const double Aw = h * iIb; \
    const double At = 0.5 * h2 * iIb; \
    const double Bt = h3 * iIb * (1.0 / 6.0); \
    const double Ct = h4 * iIb * (1.0 / 24.0); \
    const double sFx = F1x + F2x, sFy = F1y + F2y; \
    const double r1x = x[3] - x[0], r1y = x[4] - x[1], r2x = x[5] - x[0], r2y = x[6] - x[1]; \
    const double w1x = m1 * x[10] - x[7], w1y = m1 * x[11] - x[8]; \
    const double w2x = m2 * x[12] - x[7], w2y = m2 * x[13] - x[8]; \
    const double tau0 = r1x * F1y - r1y * F1x + r2x * F2y - r2y * F2x; \
    const double tauv = w1x * F1y - w1y * F1x + w2x * F2y - w2y * F2x; \
    const double ga1 = g * (1.0 - m1), ga2 = g * (1.0 - m2); \
    const double taua = ga1 * F1x + ga2 * F2x; \
    const double hmb = h / mb, h2mb = 0.5 * h2 / mb, hmf = h / mf, h2mf = 0.5 * h2 / mf; \
    double hc[15]; \
    hc[0] = x[7] + h * abx; \
    hc[1] = x[8] + h * aby; \
    hc[2] = x[9] + (Aw * tau0 + At * tauv + Bt * taua); \
    hc[3] = m1 * (x[10] + h * a1x); \
    hc[4] = km1 * (x[11] + h * a1y); \
    hc[5] = m2 * (x[12] + h * a2x); \
    hc[6] = km2 * (x[13] + h * a2y); \
    hc[7] = abx; \
    hc[8] = aby; \
    hc[9] = iIb * (tau0 + h * tauv + 0.5 * h2 * taua); \
    hc[10] = keep * a1x; \
    hc[11] = keep * a1y; \
    hc[12] = keep * a2x; \
    hc[13] = keep * a2y; \
    hc[14] = keep; \
    const double t15 = -At * r1y - Bt * w1y + Ct * ga1, t16 = At * r1x + Bt * w1x; \
    const double t17 = -At * r2y - Bt * w2y + Ct * ga2, t18 = At * r2x + Bt * w2x; \
    const double o15 = -Aw * r1y - At * w1y + Bt * ga1, o16 = Aw * r1x + At * w1x; \
    const double o17 = -Aw * r2y - At * w2y + Bt * ga2, o18 = Aw * r2x + At * w2x; \
    const double mF1x = m1 * F1x, mF1y = m1 * F1y, mF2x = m2 * F2x, mF2y = m2 * F2y; \
    const double s_m1h = m1 * h, s_m1h2 = -m1 * h2mf, s_k1h = km1 * h, s_k1h2 = -km1 * h2mf; \
    const double s_m2h = m2 * h, s_m2h2 = -m2 * h2mf, s_k2h = km2 * h, s_k2h2 = -km2 * h2mf; \
    const double s_k1f = -km1 * hmf, s_k2f = -km2 * hmf; \
    double wAt = At, wBt = Bt, wAw = Aw;

// The 85 entries of the union pattern of a step block, as JW(row, col, value) statements over the
// base quantities of the Jacobian phase (closed form in the file header): column 19 (d/dh, dense), row 2
// (theta) and row 9 (omega), rows 0-1 (body position), rows 3-6 (foot positions; the y rows are masked at
// the jump), rows 7-8 (body velocity), rows 10-13 (foot velocities) and row 14 (clock), masked at the jump
// (quirk Q1).  The statements are in column-major order of (row, col) -- the order of the values inside a block
// in both formats; the structural format's emission relies on it.
#define QLN_STEP_ENTRIES()                                                                                        \
    /* columns 0-6: positions */                                                                                  \
    JW(0, 0, 1.0); JW(2, 0, -wAt * sFy); JW(9, 0, -wAw * sFy);                                                    \
    JW(1, 1, 1.0); JW(2, 1, wAt * sFx); JW(9, 1, wAw * sFx);                                                      \
    JW(2, 2, 1.0);                                                                                                \
    JW(2, 3, wAt * F1y); JW(3, 3, 1.0); JW(9, 3, wAw * F1y);                                                      \
    JW(2, 4, -wAt * F1x); JW(4, 4, keep); JW(9, 4, -wAw * F1x);                                                   \
    JW(2, 5, wAt * F2y); JW(5, 5, 1.0); JW(9, 5, wAw * F2y);                                                      \
    JW(2, 6, -wAt * F2x); JW(6, 6, keep); JW(9, 6, -wAw * F2x);                                                   \
    /* columns 7-14: velocities and the clock */                                                                  \
    JW(0, 7, h); JW(2, 7, -wBt * sFy); JW(7, 7, 1.0); JW(9, 7, -wAt * sFy);                                       \
    JW(1, 8, h); JW(2, 8, wBt * sFx); JW(8, 8, 1.0); JW(9, 8, wAt * sFx);                                         \
    JW(2, 9, h); JW(9, 9, 1.0);                                                                                   \
    JW(2, 10, wBt * mF1y); JW(3, 10, s_m1h); JW(9, 10, wAt * mF1y); JW(10, 10, keep);                             \
    JW(2, 11, -wBt * mF1x); JW(4, 11, s_k1h); JW(9, 11, -wAt * mF1x); JW(11, 11, keep);                           \
    JW(2, 12, wBt * mF2y); JW(5, 12, s_m2h); JW(9, 12, wAt * mF2y); JW(12, 12, keep);                             \
    JW(2, 13, -wBt * mF2x); JW(6, 13, s_k2h); JW(9, 13, -wAt * mF2x); JW(13, 13, keep);                           \
    JW(14, 14, keep);                                                                                             \
    /* columns 15-18: forces */                                                                                   \
    JW(0, 15, h2mb); JW(2, 15, t15); JW(3, 15, s_m1h2); JW(7, 15, hmb); JW(9, 15, o15); JW(10, 15, s_k1f);        \
    JW(1, 16, h2mb); JW(2, 16, t16); JW(4, 16, s_k1h2); JW(8, 16, hmb); JW(9, 16, o16); JW(11, 16, s_k1f);        \
    JW(0, 17, h2mb); JW(2, 17, t17); JW(5, 17, s_m2h2); JW(7, 17, hmb); JW(9, 17, o17); JW(12, 17, s_k2f);        \
    JW(1, 18, h2mb); JW(2, 18, t18); JW(6, 18, s_k2h2); JW(8, 18, hmb); JW(9, 18, o18); JW(13, 18, s_k2f);        \
    /* column 19: the step length h */                                                                            \
    JW(0, 19, hc[0]); JW(1, 19, hc[1]); JW(2, 19, hc[2]); JW(3, 19, hc[3]); JW(4, 19, hc[4]);                     \
    JW(5, 19, hc[5]); JW(6, 19, hc[6]); JW(7, 19, hc[7]); JW(8, 19, hc[8]); JW(9, 19, hc[9]);                     \
    JW(10, 19, hc[10]); JW(11, 19, hc[11]); JW(12, 19, hc[12]); JW(13, 19, hc[13]); JW(14, 19, hc[14])

}  // namespace
}  // namespace qln
