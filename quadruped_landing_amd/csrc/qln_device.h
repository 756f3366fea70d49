// qln_device.h -- launch interface between the C-ABI host layer (qln_api.cpp) and
// the gfx950 kernels (qln_kernels.hip).  Internal; the public boundary is
// include/qln_evaluator.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qln_evaluator.h"

namespace qln {

// Per-problem descriptor, one 32-byte record per problem so that a wave fetches everything it needs to
// start with ONE scalar load (k_trans / init_mode of HybridNLP, src/nlp.jl:17-19, and the output offsets).
struct alignas(32) ProblemDesc {
    int32_t k_trans;    // 1-based start index of mode 3
    int32_t init_mode;  // 1 or 2
    int64_t c_off;      // offset of the problem's constraint vector in c
    int64_t j_off;      // offset of the problem's Jacobian values in vals (even)
    int64_t reserved;
};

// Device-resident description of a batch (mirrors HybridNLP, src/nlp.jl:13-33, per problem).
struct BatchParams {
    int32_t B;
    int32_t N;
    double g, mb, mf, lb;      // PlanarQuadruped, src/planar_quadruped.jl:11-20
    const ProblemDesc* desc;   // [B]
    const double* bnd;         // [B][30]: x0 (15) then xf (15) of each problem
    const double* cost;        // [cost_batch][N][41]
    int32_t cost_batch;
    int64_t z_stride;
    int32_t jac_format;        // QLN_JAC_FORMAT_*: layout of the step-block section of vals
    int32_t kt_max;            // largest k_trans of the batch (host-side sizing of the structural format's LDS tile)
};

// ---------------------------------------------------------------------------------------------
// Structural non-zeros of the 15x20 step Jacobian d(x+)/d[x;u] (contact*_jacobian,
// src/planar_quadruped.jl:225-248, times the jump mask of :262-263 at the transition knot).
// A knot falls into one of five categories; the pattern of each is fixed:
//   0  contact mode 1 (foot 1 pinned, foot 2 free)   71 entries
//   1  contact mode 2 (foot 2 pinned, foot 1 free)   71
//   2  mode 3 (both feet pinned)                      57
//   3  mode 1 followed by the jump map                56   (rows 5, 7, 11-15 masked; quirk Q1)
//   4  mode 2 followed by the jump map                56
// In QLN_JAC_FORMAT_STRUCTURAL the values of a block are stored in column-major order of its pattern.
// ---------------------------------------------------------------------------------------------
constexpr int kStepCategories = 5;

__host__ __device__ constexpr bool step_entry_present(int cat, int row, int col) {
    const bool f1 = (cat == 1 || cat == 4);  // foot 1 free
    const bool f2 = (cat == 0 || cat == 3);  // foot 2 free
    const bool keep = cat < 3;               // not masked by the jump
    switch (row) {
        case 0: return col == 0 || col == 7 || col == 15 || col == 17 || col == 19;
        case 1: return col == 1 || col == 8 || col == 16 || col == 18 || col == 19;
        case 2:   // theta: every position, velocity, force and h
        case 9:   // omega: the same without theta itself
            if (col == 14) return false;
            if (col == 2) return row == 2;
            if (col == 10 || col == 11) return f1;
            if (col == 12 || col == 13) return f2;
            return true;
        case 3: return col == 3 || (f1 && (col == 10 || col == 15 || col == 19));
        case 4: return keep && (col == 4 || (f1 && (col == 11 || col == 16 || col == 19)));
        case 5: return col == 5 || (f2 && (col == 12 || col == 17 || col == 19));
        case 6: return keep && (col == 6 || (f2 && (col == 13 || col == 18 || col == 19)));
        case 7: return col == 7 || col == 15 || col == 17 || col == 19;
        case 8: return col == 8 || col == 16 || col == 18 || col == 19;
        case 10: return keep && (col == 10 || (f1 && (col == 15 || col == 19)));
        case 11: return keep && (col == 11 || (f1 && (col == 16 || col == 19)));
        case 12: return keep && (col == 12 || (f2 && (col == 17 || col == 19)));
        case 13: return keep && (col == 13 || (f2 && (col == 18 || col == 19)));
        case 14: return keep && (col == 14 || col == 19);
        default: return false;
    }
}

// position of (row, col) inside the category's value list (column-major over the pattern)
__host__ __device__ constexpr int step_entry_pos(int cat, int row, int col) {
    int n = 0;
    for (int c = 0; c <= col; ++c)
        for (int r = 0; r < 15; ++r) {
            if (c == col && r == row) return n;
            if (step_entry_present(cat, r, c)) ++n;
        }
    return n;
}

__host__ __device__ constexpr int step_nnz(int cat) { return step_entry_pos(cat, 15, 19); }

// union of the five patterns (85 entries) and the position of an entry inside it (column-major)
constexpr int kStepUnion = 85;
__host__ __device__ constexpr bool step_union_present(int row, int col) {
    return step_entry_present(0, row, col) || step_entry_present(1, row, col);
}
__host__ __device__ constexpr int step_union_pos(int row, int col) {
    int n = 0;
    for (int c = 0; c <= col; ++c)
        for (int r = 0; r < 15; ++r) {
            if (c == col && r == row) return n;
            if (step_union_present(r, c)) ++n;
        }
    return n;
}
static_assert(step_union_pos(15, 19) == kStepUnion, "the union of the patterns has 85 entries");
static_assert(step_nnz(0) == 71 && step_nnz(1) == 71 && step_nnz(2) == 57 && step_nnz(3) == 56 && step_nnz(4) == 56,
              "structural non-zero counts of SURVEY.md 8.0");

// Category of dynamics knot K (1-based, 1..N-1) under the mode schedule of src/constraints.jl:23-37.
__host__ __device__ constexpr int step_category(int K, int k_trans, int init_mode) {
    return (K == k_trans - 1) ? (init_mode == 1 ? 3 : 4) : (K < k_trans - 1) ? (init_mode == 1 ? 0 : 1) : 2;
}

// Offset (doubles) of 0-based knot k's block inside the step-block section of a problem, structural format:
// contact knots first (71 each), then the jump knot (56), then mode 3 (57 each).  step_block_offset(N-1) is the
// section's length.
__host__ __device__ constexpr int step_block_offset(int k, int N, int k_trans) {
    const int nc = (k_trans - 2 < 0) ? 0 : (k_trans - 2 > N - 1 ? N - 1 : k_trans - 2);  // knots before the jump
    const int nj = (k_trans >= 2 && k_trans <= N) ? 1 : 0;
    return (k <= nc) ? 71 * k : 71 * nc + 56 * nj + 57 * (k - nc - nj);
}

// internal launch flag (bit 0 is QLN_JAC_WRITE_CONSTANTS): prefer latency over throughput for a small batch
constexpr uint32_t kLaunchSplit = 2u;
// L2 prefetch for a later workgroup of the XCD (k_constraint_jacobian): distance in problems in bits 8..28 of the flags (0 = off),
// what is prefetched in bits 29..31 (default: the slice of Z only)
constexpr uint32_t kDensePrefetchAhead = 64;
constexpr uint32_t kPrefetchNoZ = 1u << 29, kPrefetchBnd = 1u << 30, kPrefetchDesc = 1u << 31;

// Fused eval_c! + jac_c! over problems [b_begin, b_begin + nb).  c or vals may be null.
hipError_t launch_constraint_jacobian(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c,
                                      double* vals, uint32_t flags, hipStream_t stream);
hipError_t launch_eval_all(const BatchParams& p, const double* Z, double* f, double* grad, double* c, double* vals, uint32_t flags,
                           hipStream_t stream);
hipError_t launch_objective_and_constraint(const BatchParams& p, const double* Z, double* f, double* c, hipStream_t stream);
hipError_t launch_kinematic_rows(const BatchParams& p, const double* Z, double* d, double* jac_vals, hipStream_t stream);
hipError_t launch_friction_rows(const BatchParams& p, const double* Z, double mu, double* d, double* jac_vals, hipStream_t stream);
hipError_t launch_jacobian_constants(const BatchParams& p, double* vals, hipStream_t stream);
hipError_t launch_objective(const BatchParams& p, const double* Z, double* f, hipStream_t stream);
hipError_t launch_objective_gradient(const BatchParams& p, const double* Z, double* grad, hipStream_t stream);
hipError_t launch_initial_guess(const BatchParams& p, double* Z, hipStream_t stream);
hipError_t launch_constraint_violation(const BatchParams& p, const double* c, double* viol, hipStream_t stream);
// y = J(Z) v and g = J(Z)^T lam with the constraint Jacobian re-derived in registers (qln_solver_kernels.hip)
hipError_t launch_constraint_jvp(const BatchParams& p, const double* Z, const double* v, double* y, hipStream_t stream);
hipError_t launch_constraint_vjp(const BatchParams& p, const double* Z, const double* lam, double* g, hipStream_t stream);
// batched Gauss-Newton step on the constraint violation, CGLS per problem in LDS (qln_solver_kernels.hip)
size_t gauss_newton_lds_bytes(int32_t N);
hipError_t launch_gauss_newton_step(const BatchParams& p, const double* Z, const double* c, double* dZ, int max_iters,
                                    double rel_tol, const double* radius, const double* col_scale, double* info,
                                    hipStream_t stream);
// device-side generator of the synthetic workload (qln_sampler_kernels.hip)
struct DropStateSampler {
    unsigned long long state_hi, state_lo, inc_hi, inc_lo;  // numpy.random.PCG64(seed).state
    long long stream_offset;                                // draws of the stream consumed before this call
    double x0_template[15];
    double lo[4], range[4];                                 // theta0 [deg], y2_0, drop height H, omega0: low and high - low
    double deg2rad, two_g;
};
hipError_t launch_sample_drop_states(const BatchParams& p, const DropStateSampler& s, double* bnd, hipStream_t stream);
hipError_t launch_bounded_integers(const DropStateSampler& s, uint32_t range, int32_t low, int64_t count, int32_t* out,
                                   unsigned long long* rejected, hipStream_t stream);
hipError_t launch_perturb_point(const BatchParams& p, const DropStateSampler& s, double* Z, double sigma, double h_lo, double h_hi,
                                int redraw_h, hipStream_t stream);
// batched augmented-Lagrangian iLQR solve of the reference NLP (qln_ilqr_kernels.hip)
struct SolveParams {
    int32_t max_outer, max_inner;
    double tol, inner_tol;
    double rho0, rho_factor, rho_max;
    double mu0, mu_min, mu_max;
    double h_lo, h_hi, th_lo, th_hi;
    double h_prox;  // proximal weight on the step lengths in Quu: without the d(h l)/dh term (quirk Q2) the objective does
                    // not see h at all, the h_k are then fixed by the constraints alone and wander along flat directions
    int32_t q6, exact_h;
    int32_t rescue_outer;  // extra multiplier updates with accurate inner solves for problems the schedule did not finish
};
size_t ilqr_lds_bytes(int32_t N);
size_t ilqr_scratch_doubles(int32_t B, int32_t N);
hipError_t launch_al_ilqr(const BatchParams& p, const SolveParams& s, double* Z, double* info, double* scratch,
                          hipStream_t stream);
hipError_t launch_lqr_cost(const BatchParams& p, const double* qrqf, double dt, double* cost, int cost_batch,
                           hipStream_t stream);

}  // namespace qln
