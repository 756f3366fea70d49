// qln_device.h -- launch interface between the C-ABI host layer (qln_api.cpp) and
// the gfx950 kernels (qln_kernels.hip).  Internal; the public boundary is
// include/qln_evaluator.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
};

// Fused eval_c! + jac_c! over problems [b_begin, b_begin + nb).  c or vals may be null.
hipError_t launch_constraint_jacobian(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c,
                                      double* vals, uint32_t flags, hipStream_t stream);
hipError_t launch_jacobian_constants(const BatchParams& p, double* vals, hipStream_t stream);
hipError_t launch_objective(const BatchParams& p, const double* Z, double* f, hipStream_t stream);
hipError_t launch_objective_gradient(const BatchParams& p, const double* Z, double* grad, hipStream_t stream);
hipError_t launch_initial_guess(const BatchParams& p, double* Z, hipStream_t stream);
hipError_t launch_constraint_violation(const BatchParams& p, const double* c, double* viol, hipStream_t stream);
hipError_t launch_lqr_cost(const BatchParams& p, const double* qrqf, double dt, double* cost, int cost_batch,
                           hipStream_t stream);

}  // namespace qln
