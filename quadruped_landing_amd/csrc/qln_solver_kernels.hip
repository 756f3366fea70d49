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

// coalesced copy of n doubles global -> LDS by one wave
__device__ __forceinline__ void stage(double* dst, const double* __restrict__ src, int n, int lane) {
    for (int i = lane; i < n; i += kWave) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// y = J(Z) v
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_constraint_jvp(BatchParams P, const double* __restrict__ Z,
                                                         const double* __restrict__ V, double* __restrict__ Y) {
    __shared__ double s_z[kPZ + 1], s_v[kPZ + 1], s_y[kPC * 15 + 1];
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
        wave_lds_sync();
        stage(s_z, Zb + 20 * kc0, nz, lane);
        stage(s_v, Vb + 20 * kc0, nz, lane);
        wave_lds_sync();
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
            for (int i = 0; i < 15; ++i) s_y[lane * 15 + i] = y[i] - vk[20 + i];
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
    __shared__ double s_z[kPZ + 1], s_l[15 * (kPC + 1)], s_g[20 * (kPC + 1)];
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
        wave_lds_sync();
        stage(s_z, Zb + 20 * kc0, nz, lane);
        for (int i = lane; i < 15 * (nk + 1); i += kWave) {
            const int j = 15 * (kc0 - 1) + i;  // index into the dynamics rows
            s_l[i] = (j >= 0) ? Lb[pv.o_dyn + j] : 0.0;
        }
        wave_lds_sync();
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
                    for (int i = 0; i < 15; ++i) gk[i] += Lb[i];        // I(15) on x_1
                }
                if (kk == N - 1) {
#pragma unroll
                    for (int i = 0; i < 14; ++i) gk[i] += Lb[15 + i];   // I(15)[1:14,:] on x_N
                }
                const double l_ci = Lb[pv.o_ci + kk];
                const double l_co = (K >= kt) ? Lb[pv.o_co + (K - kt)] : 0.0;
                gk[4] += pv.init1 ? l_ci : l_co;
                gk[6] += pv.init1 ? l_co : l_ci;
                if (kk == N - 2) {
                    const double l_fc = Lb[pv.o_fc];
                    gk[16] += l_fc;
                    gk[18] += l_fc;
                }
                const double l_bp = Lb[pv.o_bp + kk];
                gk[1] += l_bp;
                gk[2] += dth * l_bp;
#pragma unroll
                for (int i = 0; i < 20; ++i) s_g[20 * lane + i] = gk[i];
            }
        }
        wave_lds_sync();
        const int ng = 20 * nk + (last_chunk ? 15 : 0);
        for (int i = lane; i < ng; i += kWave) Gb[20 * kc0 + i] = s_g[i];
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

}  // namespace qln
