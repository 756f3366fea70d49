// qln_device.h -- launch interface between the C-ABI host layer (qln_api.cpp) and
// the gfx950 kernels (qln_kernels.hip).  Internal; the public boundary is
// include/qln_evaluator.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qln {

// Device-resident description of a batch (mirrors HybridNLP, src/nlp.jl:13-33, per problem).
struct BatchParams {
    int32_t B;
    int32_t N;
    double g, mb, mf, lb;      // PlanarQuadruped, src/planar_quadruped.jl:11-20
    const int32_t* k_trans;    // [B] 1-based start index of mode 3
    const int32_t* init_mode;  // [B]
    const double* x0;          // [B][15]
    const double* xf;          // [B][15]
    const double* cost;        // [cost_batch][N][41]
    int32_t cost_batch;
    int64_t z_stride;
    const int64_t* c_off;      // [B]
    const int64_t* j_off;      // [B], even
};

// Fused eval_c! + jac_c! over problems [b_begin, b_begin + nb).  c or vals may be null.
hipError_t launch_constraint_jacobian(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c,
                                      double* vals, uint32_t flags, hipStream_t stream);
hipError_t launch_jacobian_constants(const BatchParams& p, double* vals, hipStream_t stream);
hipError_t launch_objective(const BatchParams& p, const double* Z, double* f, hipStream_t stream);
hipError_t launch_objective_gradient(const BatchParams& p, const double* Z, double* grad, hipStream_t stream);
hipError_t launch_initial_guess(const BatchParams& p, double* Z, hipStream_t stream);

}  // namespace qln
