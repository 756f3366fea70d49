// qln_solver_kernels.hip -- gfx950 kernels for the caller side of the evaluator: products with
// the constraint Jacobian of jac_c! (src/constraints.jl:212-291) that never materialise it.
//
//   k_constraint_jvp   y = J(Z) v        v in the layout of Z, y in the layout of c
//   k_constraint_vjp   g = J(Z)^T lam    lam in the layout of c, g in the layout of Z
//
// A stored Jacobian costs 2400 B (dense block) or ~570 B (structural format) of HBM traffic per knot every time it
// is applied; its 85 possible non-zeros follow from the knot's 20 inputs in ~150 flops (closed form in
// qln_kernels.hip's header).  So a product re-derives the block in registers: a J v or J^T lam of the whole batch
// reads Z and one vector and writes one vector -- ~18 KB per N=40 problem instead of 106 KB.
//
// Mapping as in the evaluator: one 64-lane wavefront (= one workgroup) per problem, lane = knot, chunks of 63
// dynamics knots so that lane nk is free for the knot behind the chunk (the terminal knot x_N in the last one).
#include "qln_kernel_common.h"

// Nothing here has to round like the reference (the parity tests hold these kernels to 1e-8), so a*b+c may fuse.
#pragma clang fp contract(fast)

namespace qln {
namespace {

constexpr int kPC = 63;                   // dynamics knots per chunk
constexpr int kPZ = 20 * kPC + 15;        // doubles of Z (or of a vector in Z's layout) a chunk touches

struct ProblemView {
    int N, kt, im;
    int o_dyn, o_ci, o_co, o_fc, o_bp;    // 0-based offsets of the constraint groups (cinds, src/nlp.jl:48-63)
    bool init1;
};

__device__ __forceinline__ ProblemView view_of(const BatchParams& P, const ProblemDesc& pd) {
    ProblemView v;
    v.N = P.N;
    v.kt = pd.k_trans;
    v.im = pd.init_mode;
    v.o_dyn = 29;
    v.o_ci = v.o_dyn + 15 * (P.N - 1);
    v.o_co = v.o_ci + P.N;
    v.o_fc = v.o_co + (P.N - v.kt + 1);
    v.o_bp = v.o_fc + 1;
    v.init1 = (v.im == 1);
    return v;
}

// d(clearance_k)/d(theta_k), src/constraints.jl:269-273 (theta == 0 takes the + branch, quirk Q3)
__device__ __forceinline__ double clearance_dtheta(double th, double lb) {
    const double cth = cos(th);
    return (th > 0) ? (-lb / 2 * cth) : (lb / 2 * cth);
}

// Coalesced copy of n <= kPZ doubles global -> LDS by one wave, every load in flight before the first wait (a plain
// copy loop is one memory round trip per 64 doubles).  Indices past n are clamped, not predicated; dst has room for
// kStageIters * 64 doubles.
constexpr int kStageIters = (kPZ + kWave - 1) / kWave;
constexpr int kStageRoom = kStageIters * kWave;

__device__ __forceinline__ void stage_load(double (&reg)[kStageIters], const double* __restrict__ src, int n, int lane) {
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) reg[it] = src[min(it * kWave + lane, n - 1)];
}
__device__ __forceinline__ void stage_store(double* dst, const double (&reg)[kStageIters], int lane) {
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) dst[it * kWave + lane] = reg[it];
}

// ---------------------------------------------------------------------------------------------
// y = J(Z) v
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_constraint_jvp(BatchParams P, const double* __restrict__ Z,
                                                         const double* __restrict__ V, double* __restrict__ Y) {
    // s_y (the chunk's dynamics rows, transposed for the coalesced store) lives in s_z's bytes: once every lane has its
    // knot's state in registers the staged slice is dead -- 21.5 KB instead of 29 KB of LDS, 7 waves per CU instead of 5
    __shared__ double s_z[kStageRoom], s_v[kStageRoom];
    static_assert(kStageRoom >= kPC * 15 + 1, "the product rows fit in the slice they alias");
    double* const s_y = s_z;
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;  // wave-uniform
    const ProblemDesc pd = P.desc[b];
    const ProblemView pv = view_of(P, pd);
    const int N = pv.N, kt = pv.kt, im = pv.im;
    const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;
    const double* __restrict__ Vb = V + (int64_t)b * P.z_stride;
    double* __restrict__ Yb = Y + pd.c_off;
    const double g = P.g, mb = P.mb, mf = P.mf, lb = P.lb;
    const double Ib = mb * (lb * lb) / 12;

    for (int kc0 = 0; kc0 < N - 1; kc0 += kPC) {
        const int nk = min(kPC, N - 1 - kc0);
        const int nz = 20 * nk + 15;
        const bool first_chunk = (kc0 == 0), last_chunk = (kc0 + nk == N - 1);
        {
            double zr[kStageIters], vr[kStageIters];
            stage_load(zr, Zb + 20 * kc0, nz, lane);
            stage_load(vr, Vb + 20 * kc0, nz, lane);
            wave_lds_sync();  // the previous chunk's readers are done
            stage_store(s_z, zr, lane);
            stage_store(s_v, vr, lane);
            wave_lds_sync();
        }
        // rows of I(15) on x_1 (src/constraints.jl:228), I(15)[1:14,:] on x_N (:229), final control (:259-260)
        if (first_chunk && lane < 15) Yb[lane] = s_v[lane];
        if (last_chunk) {
            if (lane >= 15 && lane < 29) Yb[lane] = s_v[20 * nk + (lane - 15)];
            if (lane == 29) Yb[pv.o_fc] = s_v[20 * (nk - 1) + 16] + s_v[20 * (nk - 1) + 18];
        }
        const bool valid = lane < nk;
        const bool own = valid || (last_chunk && lane == nk);  // lane nk of the last chunk holds x_N
        const int kk = kc0 + lane, K = kk + 1;
        const double* zk = s_z + 20 * (own ? lane : 0);
        const double* vk = s_v + 20 * (own ? lane : 0);
        {
            // contact rows (:235-256) and clearance rows (:263-274), one per knot
            const double dth = clearance_dtheta(zk[2], lb);
            if (own) {
                Yb[pv.o_ci + kk] = pv.init1 ? vk[4] : vk[6];
                if (K >= kt) Yb[pv.o_co + (K - kt)] = pv.init1 ? vk[6] : vk[4];
                Yb[pv.o_bp + kk] = vk[1] + dth * vk[2];
            }
        }
        if (valid) {
            // dynamics rows: D[ci, [x_k; u_k]] = J_k (jump-masked at k_trans-1), D[ci, x_{k+1}] = -I (:186-200)
            double x[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) x[i] = zk[i];
            const double F1x = zk[15], F1y = zk[16], F2x = zk[17], F2y = zk[18], h = zk[19];
            const int mode = (K <= kt - 1) ? im : 3;
            const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
            QLN_STEP_BASE();
            double vin[20], y[15];
#pragma unroll
            for (int i = 0; i < 20; ++i) vin[i] = vk[i];
#pragma unroll
            for (int i = 0; i < 15; ++i) y[i] = 0.0;
#define JW(row, col, val) y[row] += (val) * vin[col]
            QLN_STEP_ENTRIES();
#undef JW
#pragma unroll
            for (int i = 0; i < 15; ++i) y[i] -= vk[20 + i];
            wave_lds_sync();  // every lane is done with the staged slice of Z: its bytes now take the product rows
#pragma unroll
            for (int i = 0; i < 15; ++i) s_y[lane * 15 + i] = y[i];
        } else {
            wave_lds_sync();
        }
        wave_lds_sync();
        for (int i = lane; i < 15 * nk; i += kWave) Yb[pv.o_dyn + 15 * kc0 + i] = s_y[i];
    }
}

// ---------------------------------------------------------------------------------------------
// g = J(Z)^T lam
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_constraint_vjp(BatchParams P, const double* __restrict__ Z,
                                                         const double* __restrict__ L, double* __restrict__ G) {
    // s_l: multipliers of the dynamics rows of knots kc0-1 .. kc0+nk-1 (15 each; knot -1 = zeros)
    // s_g (the chunk's slice of the result, for the coalesced store) lives in s_z's bytes, see k_constraint_jvp:
    // 18.4 KB instead of 28.7 KB of LDS, 8 waves per CU instead of 5
    // s_b: the 29 multipliers of the initial-state and terminal rows (one lane adds fifteen of them to x_1, one fourteen to x_N)
    __shared__ double s_z[kStageRoom], s_l[15 * (kPC + 1)], s_b[30];
    static_assert(kStageRoom >= 20 * (kPC + 1), "the result slice fits in the slice of Z it aliases");
    double* const s_g = s_z;
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;  // wave-uniform
    const ProblemDesc pd = P.desc[b];
    const ProblemView pv = view_of(P, pd);
    const int N = pv.N, kt = pv.kt, im = pv.im;
    const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;
    const double* __restrict__ Lb = L + pd.c_off;
    double* __restrict__ Gb = G + (int64_t)b * P.z_stride;
    const double g = P.g, mb = P.mb, mf = P.mf, lb = P.lb;
    const double Ib = mb * (lb * lb) / 12;

    for (int kc0 = 0; kc0 < N - 1; kc0 += kPC) {
        const int nk = min(kPC, N - 1 - kc0);
        const int nz = 20 * nk + 15;
        const bool last_chunk = (kc0 + nk == N - 1);
        double l_ci, l_co, l_bp, l_fc;
        {
            constexpr int kLamIters = (15 * (kPC + 1) + kWave - 1) / kWave;
            double zr[kStageIters], lr[kLamIters];
            stage_load(zr, Zb + 20 * kc0, nz, lane);
            const int nl = 15 * (nk + 1);
#pragma unroll
            for (int it = 0; it < kLamIters; ++it) {
                const int i = min(it * kWave + lane, nl - 1);
                const int j = 15 * (kc0 - 1) + i;  // index into the dynamics rows; knot -1 has no multipliers
                lr[it] = Lb[pv.o_dyn + max(j, 0)];
                if (j < 0) lr[it] = 0.0;
            }
            // Every multiplier a lane adds behind the products -- contact, final-control and clearance rows of its knot, the
            // initial-state / terminal rows -- is requested HERE, with the slices, unconditionally (clamped indices): requested
            // where it is used, inside the lane's branches, each was a memory round trip of its own at the end of the wave's life.
            const int kq = min(kc0 + lane, N - 1);
            l_ci = Lb[pv.o_ci + kq];
            l_co = Lb[pv.o_co + min(max(kq + 1 - kt, 0), N - kt)];
            l_bp = Lb[pv.o_bp + kq];
            l_fc = Lb[pv.o_fc];
            const double l_b = Lb[min(lane, 28)];
            wave_lds_sync();  // the previous chunk's readers are done
            stage_store(s_z, zr, lane);
#pragma unroll
            for (int it = 0; it < kLamIters; ++it)
                if (it * kWave + lane < 15 * (kPC + 1)) s_l[it * kWave + lane] = lr[it];
            if (lane < 29) s_b[lane] = l_b;
            wave_lds_sync();
        }
        const bool valid = lane < nk;
        const bool own = valid || (last_chunk && lane == nk);
        const int kk = kc0 + lane, K = kk + 1;
        const double* zk = s_z + 20 * (own ? lane : 0);
        double gk[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) gk[i] = 0.0;
        if (valid) {
            double x[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) x[i] = zk[i];
            const double F1x = zk[15], F1y = zk[16], F2x = zk[17], F2y = zk[18], h = zk[19];
            const int mode = (K <= kt - 1) ? im : 3;
            const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
            QLN_STEP_BASE();
            double lam[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) lam[i] = s_l[15 * (lane + 1) + i];
#define JW(row, col, val) gk[col] += (val) * lam[row]
            QLN_STEP_ENTRIES();
#undef JW
        }
        {
            const double dth = clearance_dtheta(zk[2], lb);
            if (own) {
                // -I of the previous dynamics knot (zeros for the first knot)
#pragma unroll
                for (int i = 0; i < 15; ++i) gk[i] -= s_l[15 * lane + i];
                if (kk == 0) {
#pragma unroll
                    for (int i = 0; i < 15; ++i) gk[i] += s_b[i];        // I(15) on x_1
                }
                if (kk == N - 1) {
#pragma unroll
                    for (int i = 0; i < 14; ++i) gk[i] += s_b[15 + i];   // I(15)[1:14,:] on x_N
                }
                const double l_cok = (K >= kt) ? l_co : 0.0;
                gk[4] += pv.init1 ? l_ci : l_cok;
                gk[6] += pv.init1 ? l_cok : l_ci;
                if (kk == N - 2) {
                    gk[16] += l_fc;
                    gk[18] += l_fc;
                }
                gk[1] += l_bp;
                gk[2] += dth * l_bp;
            }
            wave_lds_sync();  // every lane is done with the staged slice of Z: its bytes now take the result
            if (own) {
#pragma unroll
                for (int i = 0; i < 20; ++i) s_g[20 * lane + i] = gk[i];
            }
        }
        wave_lds_sync();
        const int ng = 20 * nk + (last_chunk ? 15 : 0);
        for (int i = lane; i < ng; i += kWave) Gb[20 * kc0 + i] = s_g[i];
    }
}


// ---------------------------------------------------------------------------------------------
// Batched Gauss-Newton step on the constraint violation: for every problem
//     dZ = argmin || A dZ + rho ||_2   of minimum norm,
// where rho_i = c_i on the equality rows and min(c_i, 0) on the clearance rows (the bounds of src/nlp.jl:66-69),
// and A = jac_c(Z) with the rows of satisfied clearance constraints removed.  Solved by CGLS (conjugate gradients on
// the normal equations, started at 0, so the iterates stay in range(A') and converge to the minimum-norm solution
// even though A is rank deficient).  One wavefront owns one problem and keeps EVERYTHING in LDS -- Z, the five CGLS
// vectors and the clearance mask -- so an iteration touches no global memory at all; A is never formed: both products
// re-derive each step block from Z in registers.  HBM traffic of a whole step: read Z and c, write dZ.
// Optional: a diagonal column scaling D (variables of very different magnitude: forces ~1e2, time steps ~1e-2; a zero
// holds a variable fixed, e.g. one sitting on a bound) and a per-problem trust radius on ||D^-1 dZ|| (Steihaug-Toint
// truncation), which is what an outer trust-region loop needs.
// LDS per problem: (5 n_nlp + 2 m_nlp + N) doubles = 43 KB at N = 40, 87 KB at N = 80; N <= 149 fits the 160 KB of a CU.
// ---------------------------------------------------------------------------------------------
struct ModelConst {
    double g, mb, mf, lb, Ib;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// y = A v, all operands in LDS (z, v: layout of Z; y: layout of c; mask[k] = 1 if clearance row k is active)
// The step block of a knot does not change during a Gauss-Newton step, so when every lane owns at most one knot
// (N <= 64) its 85 entries are computed once and kept in registers (170 VGPRs; the kernel runs one wave per SIMD
// anyway, LDS being what limits residency): a product is then 85 fused multiply-adds per knot.
struct KnotJac {
    double e[kStepUnion];  // entries of the step block, column-major over the union pattern
    double dth;            // d(clearance)/d(theta)
};

__device__ __forceinline__ void knot_jacobian(const ProblemView& pv, const ModelConst& M, const double* z, int lane,
                                              KnotJac& J) {
    const int N = pv.N, kt = pv.kt, im = pv.im;
    const double g = M.g, mb = M.mb, mf = M.mf, lb = M.lb, Ib = M.Ib;
    const int kk = lane, K = kk + 1;
    const double* zk = z + 20 * (kk < N ? kk : 0);
    J.dth = clearance_dtheta(zk[2], lb);
    double x[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) x[i] = zk[i];
    const double F1x = zk[15], F1y = zk[16], F2x = zk[17], F2y = zk[18], h = zk[19];  // (unused garbage for kk >= N-1)
    const int mode = (K <= kt - 1) ? im : 3;
    const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
    QLN_STEP_BASE();
#define JW(row, col, val)                                \
    {                                                    \
        constexpr int pos_ = step_union_pos(row, col);   \
        J.e[pos_] = (val);                               \
    }
    QLN_STEP_ENTRIES();
#undef JW
}

// (A D) v with the lane's block in registers (N <= 64)
__device__ __forceinline__ void lds_jvp_cached(const ProblemView& pv, const KnotJac& J, const double* v, const double* dsc,
                                               const double* mask, double* y, int lane) {
    const int N = pv.N, kt = pv.kt;
    if (lane < 15) y[lane] = dsc[lane] * v[lane];
    if (lane >= 15 && lane < 29) y[lane] = dsc[20 * (N - 1) + (lane - 15)] * v[20 * (N - 1) + (lane - 15)];
    if (lane == 29)
        y[pv.o_fc] = dsc[20 * (N - 2) + 16] * v[20 * (N - 2) + 16] + dsc[20 * (N - 2) + 18] * v[20 * (N - 2) + 18];
    const int kk = lane, K = kk + 1;
    const bool own = kk < N, valid = kk < N - 1;
    const double* vk = v + 20 * (own ? kk : 0);
    const double* dk = dsc + 20 * (own ? kk : 0);
    if (own) {
        const double v4 = dk[4] * vk[4], v6 = dk[6] * vk[6];
        y[pv.o_ci + kk] = pv.init1 ? v4 : v6;
        if (K >= kt) y[pv.o_co + (K - kt)] = pv.init1 ? v6 : v4;
        y[pv.o_bp + kk] = mask[kk] * (dk[1] * vk[1] + J.dth * (dk[2] * vk[2]));
    }
    if (valid) {
        double vin[20], acc[15];
#pragma unroll
        for (int i = 0; i < 20; ++i) vin[i] = dk[i] * vk[i];
#pragma unroll
        for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#define JW(row, col, val)                                \
    {                                                    \
        constexpr int pos_ = step_union_pos(row, col);   \
        acc[row] += J.e[pos_] * vin[col];                \
    }
        QLN_STEP_ENTRIES();
#undef JW
#pragma unroll
        for (int i = 0; i < 15; ++i) y[pv.o_dyn + 15 * kk + i] = acc[i] - dk[20 + i] * vk[20 + i];
    }
}

// (A D)' lam with the lane's block in registers (N <= 64)
__device__ __forceinline__ void lds_vjp_cached(const ProblemView& pv, const KnotJac& J, const double* lam, const double* dsc,
                                               const double* mask, double* gz, int lane) {
    const int N = pv.N, kt = pv.kt;
    const int kk = lane, K = kk + 1;
    const bool own = kk < N, valid = kk < N - 1;
    double gk[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) gk[i] = 0.0;
    if (valid) {
        double l[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) l[i] = lam[pv.o_dyn + 15 * kk + i];
#define JW(row, col, val)                                \
    {                                                    \
        constexpr int pos_ = step_union_pos(row, col);   \
        gk[col] += J.e[pos_] * l[row];                   \
    }
        QLN_STEP_ENTRIES();
#undef JW
    }
    if (own) {
        if (kk >= 1) {
#pragma unroll
            for (int i = 0; i < 15; ++i) gk[i] -= lam[pv.o_dyn + 15 * (kk - 1) + i];
        } else {
#pragma unroll
            for (int i = 0; i < 15; ++i) gk[i] += lam[i];
        }
        if (kk == N - 1) {
#pragma unroll
            for (int i = 0; i < 14; ++i) gk[i] += lam[15 + i];
        }
        const double l_ci = lam[pv.o_ci + kk];
        const double l_co = (K >= kt) ? lam[pv.o_co + (K - kt)] : 0.0;
        gk[4] += pv.init1 ? l_ci : l_co;
        gk[6] += pv.init1 ? l_co : l_ci;
        if (kk == N - 2) {
            const double l_fc = lam[pv.o_fc];
            gk[16] += l_fc;
            gk[18] += l_fc;
        }
        const double l_bp = mask[kk] * lam[pv.o_bp + kk];
        gk[1] += l_bp;
        gk[2] += J.dth * l_bp;
#pragma unroll
        for (int i = 0; i < 15; ++i) gz[20 * kk + i] = dsc[20 * kk + i] * gk[i];
        if (valid) {
#pragma unroll
            for (int i = 15; i < 20; ++i) gz[20 * kk + i] = dsc[20 * kk + i] * gk[i];
        }
    }
}

// (A D) v: the columns of A are scaled by dsc (0 = variable held fixed); every block re-derived from z (any N)
__device__ __forceinline__ void lds_jvp(const ProblemView& pv, const ModelConst& M, const double* z, const double* v,
                                        const double* dsc, const double* mask, double* y, int lane) {
    const int N = pv.N, kt = pv.kt, im = pv.im;
    const double g = M.g, mb = M.mb, mf = M.mf, lb = M.lb, Ib = M.Ib;
    if (lane < 15) y[lane] = dsc[lane] * v[lane];
    if (lane >= 15 && lane < 29) y[lane] = dsc[20 * (N - 1) + (lane - 15)] * v[20 * (N - 1) + (lane - 15)];
    if (lane == 29)
        y[pv.o_fc] = dsc[20 * (N - 2) + 16] * v[20 * (N - 2) + 16] + dsc[20 * (N - 2) + 18] * v[20 * (N - 2) + 18];
    for (int k0 = 0; k0 < N; k0 += kWave) {
        const int kk = k0 + lane, K = kk + 1;
        const bool own = kk < N, valid = kk < N - 1;
        const double* zk = z + 20 * (own ? kk : 0);
        const double* vk = v + 20 * (own ? kk : 0);
        const double* dk = dsc + 20 * (own ? kk : 0);
        const double dth = clearance_dtheta(zk[2], lb);
        if (own) {
            const double v4 = dk[4] * vk[4], v6 = dk[6] * vk[6];
            y[pv.o_ci + kk] = pv.init1 ? v4 : v6;
            if (K >= kt) y[pv.o_co + (K - kt)] = pv.init1 ? v6 : v4;
            y[pv.o_bp + kk] = mask[kk] * (dk[1] * vk[1] + dth * (dk[2] * vk[2]));
        }
        if (valid) {
            double x[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) x[i] = zk[i];
            const double F1x = zk[15], F1y = zk[16], F2x = zk[17], F2y = zk[18], h = zk[19];
            const int mode = (K <= kt - 1) ? im : 3;
            const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
            QLN_STEP_BASE();
            double vin[20], acc[15];
#pragma unroll
            for (int i = 0; i < 20; ++i) vin[i] = dk[i] * vk[i];
#pragma unroll
            for (int i = 0; i < 15; ++i) acc[i] = 0.0;
#define JW(row, col, val) acc[row] += (val) * vin[col]
            QLN_STEP_ENTRIES();
#undef JW
#pragma unroll
            for (int i = 0; i < 15; ++i) y[pv.o_dyn + 15 * kk + i] = acc[i] - dk[20 + i] * vk[20 + i];
        }
    }
}

// gz = A' lam, all operands in LDS
// (A D)' lam
__device__ __forceinline__ void lds_vjp(const ProblemView& pv, const ModelConst& M, const double* z, const double* lam,
                                        const double* dsc, const double* mask, double* gz, int lane) {
    const int N = pv.N, kt = pv.kt, im = pv.im;
    const double g = M.g, mb = M.mb, mf = M.mf, lb = M.lb, Ib = M.Ib;
    for (int k0 = 0; k0 < N; k0 += kWave) {
        const int kk = k0 + lane, K = kk + 1;
        const bool own = kk < N, valid = kk < N - 1;
        const double* zk = z + 20 * (own ? kk : 0);
        double gk[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) gk[i] = 0.0;
        if (valid) {
            double x[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) x[i] = zk[i];
            const double F1x = zk[15], F1y = zk[16], F2x = zk[17], F2y = zk[18], h = zk[19];
            const int mode = (K <= kt - 1) ? im : 3;
            const bool jump = (K == kt - 1), f1free = (mode == 2), f2free = (mode == 1);
            QLN_STEP_BASE();
            double l[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) l[i] = lam[pv.o_dyn + 15 * kk + i];
#define JW(row, col, val) gk[col] += (val) * l[row]
            QLN_STEP_ENTRIES();
#undef JW
        }
        const double dth = clearance_dtheta(zk[2], lb);
        if (own) {
            if (kk >= 1) {
#pragma unroll
                for (int i = 0; i < 15; ++i) gk[i] -= lam[pv.o_dyn + 15 * (kk - 1) + i];
            } else {
#pragma unroll
                for (int i = 0; i < 15; ++i) gk[i] += lam[i];
            }
            if (kk == N - 1) {
#pragma unroll
                for (int i = 0; i < 14; ++i) gk[i] += lam[15 + i];
            }
            const double l_ci = lam[pv.o_ci + kk];
            const double l_co = (K >= kt) ? lam[pv.o_co + (K - kt)] : 0.0;
            gk[4] += pv.init1 ? l_ci : l_co;
            gk[6] += pv.init1 ? l_co : l_ci;
            if (kk == N - 2) {
                const double l_fc = lam[pv.o_fc];
                gk[16] += l_fc;
                gk[18] += l_fc;
            }
            const double l_bp = mask[kk] * lam[pv.o_bp + kk];
            gk[1] += l_bp;
            gk[2] += dth * l_bp;
#pragma unroll
            for (int i = 0; i < 15; ++i) gz[20 * kk + i] = dsc[20 * kk + i] * gk[i];
            if (valid) {
#pragma unroll
                for (int i = 15; i < 20; ++i) gz[20 * kk + i] = dsc[20 * kk + i] * gk[i];
            }
        }
    }
}

// info[b][8] = {iterations, ||(AD)' rho||^2, ||(AD)'(A dZ + rho)||^2 at exit, ||A dZ + rho||^2 at exit, ||rho||^2,
//              1 if the step was cut at the trust radius, ||D^-1 dZ||, 0}
template <bool CACHED>
__global__ __launch_bounds__(kWave, 1) void k_gauss_newton_step(BatchParams P, const double* __restrict__ Z,
                                                                const double* __restrict__ C, double* __restrict__ DZ,
                                                                int max_iters, double rel_tol,
                                                                const double* __restrict__ radius,
                                                                const double* __restrict__ col_scale,
                                                                double* __restrict__ info) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;  // wave-uniform
    const ProblemDesc pd = P.desc[b];
    const ProblemView pv = view_of(P, pd);
    const int N = pv.N, n = 20 * N - 5, m = 18 * N - pv.kt + 16;
    ModelConst M;
    M.g = P.g, M.mb = P.mb, M.mf = P.mf, M.lb = P.lb;
    M.Ib = P.mb * (P.lb * P.lb) / 12;
    double* z = lds;        // [n]  decision vector
    double* x = z + n;      // [n]  CGLS solution, in scaled variables: dZ = D x
    double* p = x + n;      // [n]  search direction
    double* s = p + n;      // [n]  (A D)' r
    double* dsc = s + n;    // [n]  column scaling D
    double* r = dsc + n;    // [m]  residual -(A D x + rho)
    double* q = r + m;      // [m]  A D p
    double* mask = q + m;   // [N]  1 = clearance row active (violated)
    const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;
    const double* __restrict__ Cb = C + pd.c_off;
    // eight 512-byte loads in flight per round trip (clamped indices, predicated LDS writes)
    constexpr int kU = 8;
    for (int i0 = 0; i0 < n; i0 += kU * kWave) {
        double zt[kU], dt[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = min(i0 + u * kWave + lane, n - 1);
            zt[u] = Zb[i];
            dt[u] = col_scale ? col_scale[i] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * kWave + lane;
            if (i < n) {
                z[i] = zt[u];
                dsc[i] = dt[u];
                x[i] = 0.0;
            }
        }
    }
    double phi0 = 0.0;
    for (int i0 = 0; i0 < m; i0 += kU * kWave) {
        double ct[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) ct[u] = Cb[min(i0 + u * kWave + lane, m - 1)];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * kWave + lane;
            if (i < m) {
                const double ci = ct[u];
                const bool ineq = i >= pv.o_bp;
                const double rho = (ineq && !(ci < 0)) ? 0.0 : ci;  // a NaN in c propagates
                r[i] = -rho;
                phi0 += rho * rho;
                if (ineq) mask[i - pv.o_bp] = (ci < 0 || ci != ci) ? 1.0 : 0.0;
            }
        }
    }
    phi0 = wave_sum(phi0);
    wave_lds_sync();
    KnotJac J;
    if constexpr (CACHED) knot_jacobian(pv, M, z, lane, J);
#define QLN_APPLY_A(v_, y_)                                               \
    do {                                                                  \
        if constexpr (CACHED) lds_jvp_cached(pv, J, v_, dsc, mask, y_, lane); \
        else lds_jvp(pv, M, z, v_, dsc, mask, y_, lane);                  \
    } while (0)
#define QLN_APPLY_AT(l_, g_)                                              \
    do {                                                                  \
        if constexpr (CACHED) lds_vjp_cached(pv, J, l_, dsc, mask, g_, lane); \
        else lds_vjp(pv, M, z, l_, dsc, mask, g_, lane);                  \
    } while (0)
    QLN_APPLY_AT(r, s);
    wave_lds_sync();
    double gamma = 0.0;
    for (int i = lane; i < n; i += kWave) {
        const double si = s[i];
        p[i] = si;
        gamma += si * si;
    }
    gamma = wave_sum(gamma);
    const double gamma0 = gamma;
    const double delta2 = radius ? radius[b] * radius[b] : -1.0;  // trust radius on ||x|| = ||D^-1 dZ||; none if null
    double xx = 0.0;  // ||x||^2
    double hit = 0.0;
    int it = 0;
    while (it < max_iters && gamma > rel_tol * rel_tol * gamma0) {  // wave-uniform; false for NaN
        wave_lds_sync();
        QLN_APPLY_A(p, q);
        wave_lds_sync();
        double qq = 0.0;
        for (int i = lane; i < m; i += kWave) qq += q[i] * q[i];
        qq = wave_sum(qq);
        if (!(qq > 0.0)) break;
        double alpha = gamma / qq;
        if (radius) {
            // Steihaug-Toint: CGLS iterates grow in norm, so the first one to leave the ball is cut at its boundary
            double xp = 0.0, pp = 0.0;
            for (int i = lane; i < n; i += kWave) {
                xp += x[i] * p[i];
                pp += p[i] * p[i];
            }
            xp = wave_sum(xp);
            pp = wave_sum(pp);
            const double xn2 = xx + 2.0 * alpha * xp + alpha * alpha * pp;
            if (xn2 >= delta2) {
                alpha = (-xp + sqrt(xp * xp + pp * (delta2 - xx))) / pp;
                hit = 1.0;
            }
            xx = xx + 2.0 * alpha * xp + alpha * alpha * pp;
        }
        for (int i = lane; i < n; i += kWave) x[i] += alpha * p[i];
        for (int i = lane; i < m; i += kWave) r[i] -= alpha * q[i];
        wave_lds_sync();
        QLN_APPLY_AT(r, s);
        wave_lds_sync();
        double gnew = 0.0;
        for (int i = lane; i < n; i += kWave) gnew += s[i] * s[i];
        gnew = wave_sum(gnew);
        const double beta = gnew / gamma;
        for (int i = lane; i < n; i += kWave) p[i] = s[i] + beta * p[i];
        gamma = gnew;
        ++it;
        if (hit != 0.0) break;
    }
    wave_lds_sync();
    double rr = 0.0, xn = 0.0;
    for (int i = lane; i < m; i += kWave) rr += r[i] * r[i];
    for (int i = lane; i < n; i += kWave) xn += x[i] * x[i];
    rr = wave_sum(rr);
    xn = wave_sum(xn);
    double* __restrict__ Db = DZ + (int64_t)b * P.z_stride;
    for (int i = lane; i < n; i += kWave) Db[i] = dsc[i] * x[i];
#undef QLN_APPLY_A
#undef QLN_APPLY_AT
    if (info && lane == 0) {
        double* o = info + 8 * (int64_t)b;
        o[0] = (double)it;
        o[1] = gamma0;
        o[2] = gamma;
        o[3] = rr;
        o[4] = phi0;
        o[5] = hit;
        o[6] = sqrt(xn);
        o[7] = 0.0;
    }
}

}  // namespace

hipError_t launch_constraint_jvp(const BatchParams& p, const double* Z, const double* v, double* y, hipStream_t stream) {
    hipLaunchKernelGGL(k_constraint_jvp, dim3(xcd_grid(p.B)), dim3(kWave), 0, stream, p, Z, v, y);
    return hipGetLastError();
}

hipError_t launch_constraint_vjp(const BatchParams& p, const double* Z, const double* lam, double* g, hipStream_t stream) {
    hipLaunchKernelGGL(k_constraint_vjp, dim3(xcd_grid(p.B)), dim3(kWave), 0, stream, p, Z, lam, g);
    return hipGetLastError();
}

// LDS bytes one problem of the Gauss-Newton step needs (largest m_nlp is at k_trans = 1)
size_t gauss_newton_lds_bytes(int32_t N) { return sizeof(double) * (size_t)(5 * (20 * N - 5) + 2 * (18 * N + 15) + N); }

hipError_t launch_gauss_newton_step(const BatchParams& p, const double* Z, const double* c, double* dZ, int max_iters,
                                    double rel_tol, const double* radius, const double* col_scale, double* info,
                                    hipStream_t stream) {
    const size_t lds = gauss_newton_lds_bytes(p.N);
    auto go = [&](auto kernel) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3(xcd_grid(p.B)), dim3(kWave), lds, stream, p, Z, c, dZ, max_iters, rel_tol, radius,
                           col_scale, info);
        return hipGetLastError();
    };
    // one knot per lane: the step blocks stay in registers for the whole step
    return (p.N <= kWave) ? go(k_gauss_newton_step<true>) : go(k_gauss_newton_step<false>);
}

}  // namespace qln
