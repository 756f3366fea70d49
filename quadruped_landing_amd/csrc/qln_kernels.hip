// qln_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the batched landing-NLP evaluator.
//
// Hot path: k_constraint_jacobian -- eval_c! (src/constraints.jl:145-158) and jac_c!
// (src/constraints.jl:212-291) of every knot of every problem in one launch.
//
// Mapping.  One 64-lane wavefront (= one workgroup) per landing problem; lane l owns dynamics knot
// kc0 + l.  The problem's slice of the decision vector is read with coalesced loads into LDS, each
// lane picks its (x_k, u_k, x_{k+1}) out of LDS, integrates the RK4 step in registers (literal
// reference operation order, no FMA contraction, so the residual rounds like the reference) and
// forms the 85 structurally non-zero entries of the 15x20 step Jacobian in closed form.  The dense
// 300-double blocks the reference's Jacobian is made of are then assembled T knots at a time in an
// LDS tile whose structural zeros are written once, and streamed to HBM as full 16-byte-per-lane,
// 1-KiB-per-instruction stores.  The kernel is HBM-write bound (2400 B of Jacobian per knot against
// ~0.4 kflop), so everything is organised around that store stream.
//
// Closed form of the step ("Jacobian phase"): with zero-order-hold forces every
// acceleration except the body's angular one is constant over the step, so RK4 reproduces
//   p+ = p + h v + h^2/2 a,  v+ = v + h a            (body, and the free foot; pinned foot: identity)
//   w+ = w  + (h tau0 + h^2/2 tauv + h^3/6 taua)/Ib
//   th+ = th + h w + (h^2/2 tau0 + h^3/6 tauv + h^4/24 taua)/Ib
// where tau0/tauv/taua are the force moment evaluated on relative positions / velocities /
// accelerations.  Differentiating these gives every entry of ForwardDiff.jacobian of the RK4 step
// (src/planar_quadruped.jl:225-248) up to rounding.
#include "qln_kernel_common.h"

#include <algorithm>
#include <type_traits>

#if defined(QLN_TUNING) || defined(QLN_PREFETCH_KNOB)
#include <cstdlib>
#endif

namespace qln {

#ifdef QLN_DIAG
__device__ unsigned long long* g_stamps = nullptr;
#endif
#ifdef QLN_TUNING
// tuning build only (QLN_FLOOR=1, profiles/r03_structural_floor.txt): the launch's memory shape on the very same buffers with
// (nearly) no arithmetic -- the RK4 step, the trigonometry and the structural pattern walk are skipped, every load, store and
// drain stays.  What is left between that and the real launch is arithmetic the launch does not hide.
__device__ int g_floor_mode = 0;
#define QLN_FLOOR_MODE (g_floor_mode != 0)
#else
#define QLN_FLOOR_MODE false
#endif

namespace {

// Diagnostic build only (-DQLN_DIAG, never the shipped library): per-wave s_memtime stamps at phase
// boundaries, written to a buffer of their own (cdna_hip_programming.md section 7, In-kernel stamps).
#ifdef QLN_DIAG
#define QLN_STAMP(i)                                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (g_stamps && threadIdx.x == 0) g_stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define QLN_STAMP(i) do { } while (0)
#endif

// Copy consecutive 1-KiB wave pieces (16 B per lane each) LDS -> global with all the batch's
// ds_read_b128 in flight before the first store, so a batch pays the LDS latency once.  hipcc
// re-interleaves any C++ formulation into read/wait/store through one register quad (or parks the
// batch in scratch), so the reads and their single wait are inline asm; per cdna_hip_programming.md
// section 5.7 the asm counts and waits for its own loads and no output is consumed before that wait.
typedef double v2f64 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lds_offset(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));  // low 32 bits of a generic LDS address
}

// The block stream is written once and not read again by this launch: non-temporal stores when the output is larger
// than the caches (-1.5 % on a region-placed buffer, profiles/r01_nt_stores.txt); for small batches, whose output
// stays in L2 / the Infinity Cache, plain stores are the faster ones (B = 1024: 24 against 29 us).
template <bool STREAM>
__device__ __forceinline__ void block_store(double2* dst, v2f64 v) {
    if constexpr (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<v2f64*>(dst));
    else *reinterpret_cast<v2f64*>(dst) = v;
}
template <bool STREAM>
__device__ __forceinline__ void block_store(double2* dst, double2 v) {
    v2f64 t = {v.x, v.y};
    block_store<STREAM>(dst, t);
}

template <int STEP, bool STREAM>
__device__ __forceinline__ void drain6(uint32_t lds, double2* dst) {
    v2f64 r0, r1, r2, r3, r4, r5;
    asm volatile(
        "ds_read_b128 %0, %6\n\t"
        "ds_read_b128 %1, %6 offset:1024\n\t"
        "ds_read_b128 %2, %6 offset:2048\n\t"
        "ds_read_b128 %3, %6 offset:3072\n\t"
        "ds_read_b128 %4, %6 offset:4096\n\t"
        "ds_read_b128 %5, %6 offset:5120\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
        : "v"(lds)
        : "memory");
    block_store<STREAM>(dst + 0 * STEP, r0);
    block_store<STREAM>(dst + 1 * STEP, r1);
    block_store<STREAM>(dst + 2 * STEP, r2);
    block_store<STREAM>(dst + 3 * STEP, r3);
    block_store<STREAM>(dst + 4 * STEP, r4);
    block_store<STREAM>(dst + 5 * STEP, r5);
}

template <bool STREAM>
__device__ __forceinline__ void drain1(uint32_t lds, double2* dst) {
    v2f64 r0;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(lds) : "memory");
    block_store<STREAM>(dst, r0);
}

// REM full pieces starting at (lds, dst), both already offset by the lane
template <int REM, int STEP, bool STREAM>
__device__ __forceinline__ void drain_full(uint32_t lds, double2* dst) {
    if constexpr (REM >= 6) {
        drain6<STEP, STREAM>(lds, dst);
        drain_full<REM - 6, STEP, STREAM>(lds + 6 * 1024, dst + 6 * STEP);
    } else if constexpr (REM > 0) {
        drain1<STREAM>(lds, dst);
        drain_full<REM - 1, STEP, STREAM>(lds + 1024, dst + STEP);
    }
}

// A contiguous run of doubles LDS -> global: s_j[lbeg .. lbeg + len) -> gfirst[0 .. len), with gfirst's address (in
// doubles) of the same parity as lbeg, so that 16-byte pieces are aligned on both sides.  A leading / trailing half piece
// goes out as one 8-byte store; the complete pieces as 16 B per lane, 1 KiB per wave instruction, six LDS reads in flight.
template <bool STREAM>
__device__ __forceinline__ void drain_run(double* s_j, int lbeg, int len, double* __restrict__ gfirst, int lane) {
    const int lend = lbeg + len;
    double* gbase = gfirst - lbeg;  // gbase[i] <-> s_j[i]
    if (lane == 0) {
        if (lbeg & 1) gbase[lbeg] = s_j[lbeg];
        if ((lend & 1) && len > 0) gbase[lend - 1] = s_j[lend - 1];
    }
    const int first = (lbeg + 1) >> 1;
    const int np = (lend >> 1) - first;  // complete pieces
    double2* dst = reinterpret_cast<double2*>(gbase) + first + lane;
    const double2* src = reinterpret_cast<const double2*>(s_j) + first + lane;
    int it = 0;
#pragma unroll 1
    for (; (it + 6) * kWave <= np; it += 6) drain6<kWave, STREAM>(lds_offset(src + it * kWave), dst + it * kWave);
#pragma unroll 1
    for (; it * kWave + lane < np; ++it) block_store<STREAM>(dst + it * kWave, src[it * kWave]);
}

// ---------------------------------------------------------------------------------------------
// Objective (src/costs.jl:6-16).  One knot's term with the reference's operation order: the left-to-right sums of
// 0.5 x'Qx, q'x, 0.5 u'Ru, r'u over a knot's entries (src/quadratic_cost.jl:44-52; Q, R diagonal), then
// hk * ((((a + b) + c) + d) + const) -- so that the value rounds like eval_f.  z: the knot's 20 (terminal: 15) entries
// of Z in LDS; rec: its 41-double cost record in global memory (all 41 loads in flight before the first use).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double objective_term(const double* z, const double* __restrict__ rec, bool stage) {
    double D[20], d[20], zz[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        D[i] = rec[i];
        d[i] = rec[20 + i];
    }
    const double c40 = rec[40];
#pragma unroll
    for (int i = 0; i < 20; ++i) zz[i] = z[(i < 15 || stage) ? i : 14];
    double a = (0.5 * (D[0] * zz[0])) * zz[0], bb = d[0] * zz[0];
#pragma unroll
    for (int i = 1; i < 15; ++i) {
        a = a + (0.5 * (D[i] * zz[i])) * zz[i];
        bb = bb + d[i] * zz[i];
    }
    double cc = (0.5 * (D[15] * zz[15])) * zz[15], dd = d[15] * zz[15];
#pragma unroll
    for (int i = 16; i < 20; ++i) {
        cc = cc + (0.5 * (D[i] * zz[i])) * zz[i];
        dd = dd + d[i] * zz[i];
    }
    return stage ? zz[19] * ((((a + bb) + cc) + dd) + c40) : (a + bb) + c40;
}

// the same with the record already in registers (D = rec[0..19], d = rec[20..39], c40 = rec[40]): qln_eval_all requests a
// knot's record with the slice of Z, a memory round trip before it is needed
__device__ __forceinline__ double objective_term_regs(const double* z, const double (&D)[20], const double (&d)[20], double c40, bool stage) {
    double zz[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) zz[i] = z[(i < 15 || stage) ? i : 14];
    double a = (0.5 * (D[0] * zz[0])) * zz[0], bb = d[0] * zz[0];
#pragma unroll
    for (int i = 1; i < 15; ++i) {
        a = a + (0.5 * (D[i] * zz[i])) * zz[i];
        bb = bb + d[i] * zz[i];
    }
    double cc = (0.5 * (D[15] * zz[15])) * zz[15], dd = d[15] * zz[15];
#pragma unroll
    for (int i = 16; i < 20; ++i) {
        cc = cc + (0.5 * (D[i] * zz[i])) * zz[i];
        dd = dd + d[i] * zz[i];
    }
    return stage ? zz[19] * ((((a + bb) + cc) + dd) + c40) : (a + bb) + c40;
}

// J <- J + term_0 + term_1 + ... in knot order (src/costs.jl:9-15): the wave's terms go through LDS, every lane adds
// them in the same order (lanes past the last knot hold 0.0, and J + 0.0 == J).
__device__ __forceinline__ double add_terms_in_order(double J, double term, double* s_term, int lane) {
    wave_lds_sync();
    s_term[lane] = term;
    wave_lds_sync();
    // sixteen at a time: the loads of a batch are in flight together, and 32 registers hold them instead of 128
#pragma unroll
    for (int g = 0; g < kWave; g += 16) {
        double t[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = s_term[g + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) J += t[i];
    }
    return J;
}

// The same sum without LDS, for the one kernel the LDS pipe bounds (k_objective_shared; in the fused kernel and in
// k_objective the three VALU instructions per term cost more than the broadcast read they replace -- measured, round 3):
// term i is read out of lane i (two v_readlane_b32 into a scalar pair, the add's operand), eight at a time behind a
// wave-uniform test -- J + term_0 + ... + term_{n-1} in knot order, n <= 64; lanes past n hold 0.0.
__device__ __forceinline__ double lane_value(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double add_lane_terms_in_order(double J, double term, int n) {
#pragma unroll
    for (int g = 0; g < kWave; g += 8) {
        if (g < n) {  // wave-uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) J += lane_value(term, g + i);
        }
    }
    return J;
}

// ---------------------------------------------------------------------------------------------
// Fused constraint + Jacobian kernel.
// ---------------------------------------------------------------------------------------------
// T  = knots assembled per LDS tile (tile = T*2400 B, the only LDS the kernel uses)
// KC = knots per chunk = lanes that integrate a knot at a time (<= 64); the chunk's Z slice and
//      residual stage alias the tile, so KC*35+16 doubles must fit in T*300
// W  = waves per SIMD the register budget is sized for (LDS admits 160 KiB / tile per CU)
// NNZ = structural format (QLN_JAC_FORMAT_STRUCTURAL): T is unused, the tile holds the chunk's KC compact blocks
// SPLIT = small batches: one workgroup per CHUNK of KC knots instead of per problem, so that a batch with fewer
//         problems than the chip has SIMDs still fills it (the launch is then as long as one chunk, not one problem)
// STREAM = non-temporal stores for the outputs (large batches)
// WITH_F = the objective and its gradient as well (qln_eval_all): eval_f and grad_f! out of the same staged slice of Z
template <int T, int KC, int W, bool WITH_C, bool WITH_J, bool NNZ = false, bool SPLIT = false, bool STREAM = true,
          bool WITH_F = false>
__global__ __launch_bounds__(kWave, W) void k_constraint_jacobian(BatchParams P, int32_t b_begin, int32_t nb,
                                                              const double* __restrict__ Z, double* __restrict__ C,
                                                              double* __restrict__ V, uint32_t flags,
                                                              double* __restrict__ F = nullptr, double* __restrict__ G = nullptr) {
    static_assert(!WITH_F || (!SPLIT && WITH_C), "the objective rides on the per-problem launch that also stages the value phase");
    // LDS: one tile of T dense step blocks.  Before the Jacobian phase of a chunk the same bytes
    // hold the staged Z slice (20*64+15 doubles at offset 0) and, behind it, the chunk's dynamics
    // residuals (64*15 doubles at offset kCStage).
    static_assert(KC <= kWave, "one lane per knot of a chunk");
    constexpr int kZSlice = KC * 20 + 15;
    constexpr int kCStage = (kZSlice + 1) & ~1;
    // structural format: up to 71 values per knot of the chunk, +1 so that the LDS image can start at the parity of
    // its global offset (16-byte pieces then line up on both sides)
    // Dense blocks: a static tile of T blocks.  Structural format: the tile is DYNAMIC LDS sized by the host for the batch
    // (nnz_lds_doubles below: the longest run a sub-tile of this batch can have, the staged slice and the residual stage) --
    // 71 values per knot is the worst case, a batch whose problems spend most knots in mode 3 (57) needs less, and what a
    // workgroup does not reserve another can: config 3's problems take 19.1 KB instead of 22.7 KB, 8 waves per CU instead of 7.
    constexpr int kTile = NNZ ? 2 : T * kBlk;
    static_assert(NNZ || kTile >= kCStage + KC * 15 + (WITH_F ? kWave : 0), "Z slice + residual stage (+ objective terms) must fit in the tile they alias");
    static_assert(NNZ || kTile >= ((kZSlice + kWave - 1) / kWave) * kWave, "unpredicated staging writes must fit in the tile");
    __shared__ double2 s_static2[kTile / 2];
    extern __shared__ double2 s_dyn2[];
    double2* const s_j2 = NNZ ? s_dyn2 : s_static2;
    double* const s_j = reinterpret_cast<double*>(s_j2);
    double* const s_z = s_j;
    double* const s_c = s_j + kCStage;

    const int lane = threadIdx.x;
    const int cpp = SPLIT ? (P.N - 2) / KC + 1 : 1;  // chunks (= workgroups) per problem
    const int vi = xcd_contiguous_index(blockIdx.x, nb * cpp);
    if (vi >= nb * cpp) return;  // wave-uniform
    const int bl = SPLIT ? vi / cpp : vi;
    const int b = b_begin + bl;
    QLN_STAMP(0);
    const int N = P.N;
    const int kc_begin = SPLIT ? (vi - bl * cpp) * KC : 0;      // the dynamics knots this workgroup owns
    const int kc_end = SPLIT ? min(kc_begin + KC, N - 1) : N - 1;
    const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;

    // The first chunk's slice of Z (and the boundary vectors for c1/c2) depends only on the kernel
    // arguments and the block index, so its loads are issued before anything else: the per-problem
    // descriptors (k_trans, init_mode, c_off, j_off) are then fetched in the shadow of this round
    // trip instead of in front of it.  Every load of a slice is in flight before the first wait;
    // indices past the slice are clamped, not predicated.
    constexpr int kStageIters = (kZSlice + kWave - 1) / kWave;
    double zr[kStageIters];
    double bnd = 0.0;
    {
        const int nz0 = 20 * min(KC, N - 1 - kc_begin) + 15;
#pragma unroll
        for (int it = 0; it < kStageIters; ++it) zr[it] = Zb[20 * kc_begin + min(it * kWave + lane, nz0 - 1)];
        if (WITH_C) {
            // x0 / xf for c1 / c2 (src/constraints.jl:149-150): unconditional and in bounds
            bnd = P.bnd[(int64_t)b * 30 + min(lane, 29)];  // x0[lane] for lane < 15, xf[lane - 15] after
        }
    }
    // WITH_F: what the objective and the gradient need of the cost table is requested with the slice -- the lane's knot's record
    // (eval_f, lane = knot) and the D_j, d_j of the lane's entries of the slice (grad_f!, lane = entry) -- so that they are in
    // registers when the value phase is over, not a memory round trip after it
    // WITH_F without the Jacobian = qln_eval_objective_and_constraint, the pair a line search asks for: no gradient, and the lane's
    // cost record is read where the objective term is formed, after the value phase -- this launch lives on its two waves per
    // SIMD (the RK4 step needs ~200 registers) and cannot hold a record in 80 of them from kernel entry as qln_eval_all does.
    // (Measured and not adopted: the chunk's records staged through LDS with coalesced loads, objective terms first -- the
    // slice's loop-carried staging registers then spill, 0.28-0.46 ms against 0.22.)
    constexpr bool kFullF = WITH_F && WITH_J;
    double oD[kFullF ? 20 : 1], od[kFullF ? 20 : 1], oc40 = 0.0, gD[kFullF ? kStageIters : 1], gd[kFullF ? kStageIters : 1];
    auto request_cost = [&](int kc0) {
        if constexpr (kFullF) {
            const double* __restrict__ cost = P.cost + (P.cost_batch > 1 ? (int64_t)b * P.N * 41 : 0);
            const int nk = min(KC, P.N - 1 - kc0);
            const bool last = (kc0 + nk == P.N - 1);
            const double* rec = cost + (int64_t)(kc0 + min(lane, last ? nk : nk - 1)) * 41;
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                oD[i] = rec[i];
                od[i] = rec[20 + i];
            }
            oc40 = rec[40];
            const int ng = 20 * nk + (last ? 15 : 0);
#pragma unroll
            for (int it = 0; it < kStageIters; ++it) {
                const int e = min(it * kWave + lane, ng - 1);
                const int kk = e / 20, j = e - 20 * kk;
                const double* ck = cost + (int64_t)(kc0 + kk) * 41;
                gD[it] = ck[j];
                gd[it] = ck[20 + j];
            }
        }
    };
    request_cost(kc_begin);
    __builtin_amdgcn_sched_barrier(0);

    const ProblemDesc pd = P.desc[b];  // one 32-byte scalar load
    const int kt = pd.k_trans;
    const int im = pd.init_mode;
    double* __restrict__ Cb = WITH_C ? C + pd.c_off : nullptr;
    double* __restrict__ Vb = WITH_J ? V + pd.j_off : nullptr;

    const double g = P.g, mb = P.mb, mf = P.mf, lb = P.lb;
    const double Ib = mb * (lb * lb) / 12;  // mb * lb^2 / 12, src/planar_quadruped.jl:41

    // offsets of the constraint groups inside c (0-based; cinds of src/nlp.jl:48-63)
    const int o_dyn = 29;
    const int o_ci = o_dyn + 15 * (N - 1);
    const int o_co = o_ci + N;
    const int o_fc = o_co + (N - kt + 1);
    const int o_bp = o_fc + 1;
    const bool init1 = (im == 1);  // foot 1 touches first: contact-init row is y1, contact-other is y2
    // every row of c goes through here.  (Measured and not adopted, profiles/r03_structural_floor.txt: the whole constraint
    // vector assembled in LDS behind the slice and written as ONE aligned 16-byte-per-lane stream instead of these short
    // 8-byte-per-lane stores -- no difference in the structural format's fused launch, 0.36 ms either way.)
    auto c_put = [&](int idx, double v) { Cb[idx] = v; };
    // length of the step-block section of vals
    const int dyn_blocks = NNZ ? step_block_offset(N - 1, N, kt) : kBlk * (N - 1);
    double J_obj = 0.0;  // WITH_F: eval_f accumulated over the chunks, in knot order
    // L2 prefetch for a LATER workgroup of this XCD (flags >> 8 = how many problems ahead on the XCD; 0 = off): one 4-byte load
    // per 64 bytes of that problem's slice, issued once this wave's own slice has arrived (loads return in order: a wait
    // for something requested after them would wait for them too) and never read -- their only "use" is a never-true test at
    // the very end of the kernel, where the hardware waits for every outstanding memory operation anyway (s_endpgm).  The
    // workgroup that is dispatched to the slot this or a neighbouring wave frees then finds its slice in the XCD's L2 instead
    // of paying an HBM round trip under a write-saturated memory system before it can start.  Four loads cover 16 KB of slice
    // (N <= 102; longer slices are prefetched in part), addresses clamped.
    int pf_acc[5] = {0, 0, 0, 0, 0};
    auto prefetch_later_workgroup = [&]() {
        if constexpr (!SPLIT) {
            const int ahead = (int)((flags >> 8) & 0x1fffffu);
            const int per_xcd = (nb + 7) >> 3;
            const int local = (int)(blockIdx.x >> 3) + ahead;
            const int vn = (int)(blockIdx.x & 7) * per_xcd + local;
            if (ahead > 0 && local < per_xcd && vn < nb) {  // wave-uniform
                if (!(flags & kPrefetchNoZ)) {
                    const char* Zn = reinterpret_cast<const char*>(Z + (int64_t)(b_begin + vn) * P.z_stride);
                    const int bytes = (20 * N - 5) * 8;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pf_acc[i] = *reinterpret_cast<const int*>(Zn + min(i * 4096 + lane * 64, bytes - 4));
                }
                if (flags & (kPrefetchBnd | kPrefetchDesc)) {  // that problem's boundary vectors (240 B) and / or descriptor (32 B)
                    const char* bn = reinterpret_cast<const char*>(P.bnd + (int64_t)(b_begin + vn) * 30);
                    const char* dn = reinterpret_cast<const char*>(P.desc + (b_begin + vn));
                    const bool both = (flags & kPrefetchBnd) && (flags & kPrefetchDesc);
                    const char* a = (flags & kPrefetchBnd) ? bn + min(lane * 64, 236) : dn;
                    if (both && lane >= 4) a = dn;
                    pf_acc[4] = *reinterpret_cast<const int*>(a);
                }
            }
        }
    };

    if (WITH_J && (flags & 1u) && kc_begin == 0) {
        // constant entries of jac_c! (src/constraints.jl:228-229, :200, :235-265)
        double* Vc = Vb + dyn_blocks + N;
        const int n_const = 435 + 15 * (N - 1) + 3 * N - kt + 3;
        for (int i = lane; i < n_const; i += kWave) {
            double v;
            if (i < 225) {
                v = (i % 15 == i / 15) ? 1.0 : 0.0;
            } else if (i < 435) {
                const int j = i - 225;
                v = (j % 14 == j / 14) ? 1.0 : 0.0;
            } else if (i < 435 + 15 * (N - 1)) {
                v = -1.0;
            } else {
                v = 1.0;
            }
            Vc[i] = v;
        }
    }

    // eval_f (src/costs.jl:6-16) of one chunk: lane = knot, the terminal knot x_N rides on lane nk of the last chunk (a second
    // pass when that chunk is full); terms are added in knot order.  The lane's record was requested with the slice.
    auto eval_objective_chunk = [&](int kc0, int nk, bool valid, bool last_chunk) {
        if constexpr (WITH_F) {
            const double* __restrict__ cost = P.cost + (P.cost_batch > 1 ? (int64_t)b * N * 41 : 0);
            const bool own = valid || (last_chunk && lane == nk);
            const int kk = own ? lane : 0;
            double term;
            if constexpr (kFullF) term = objective_term_regs(s_z + 20 * kk, oD, od, oc40, valid);
            else term = objective_term(s_z + 20 * kk, cost + (int64_t)(kc0 + kk) * 41, valid);
            J_obj = add_terms_in_order(J_obj, own ? term : 0.0, s_c + KC * 15, lane);
            if (last_chunk && nk == kWave) {  // wave-uniform
                const double tn = objective_term(s_z + 20 * nk, cost + (int64_t)(N - 1) * 41, false);
                J_obj = J_obj + tn;
            }
        }
    };

    for (int kc0 = kc_begin; kc0 < kc_end; kc0 += KC) {
        const int nk = min(KC, N - 1 - kc0);
        const int nz = 20 * nk + 15;
        const bool first_chunk = (kc0 == 0);
        const bool last_chunk = (kc0 + nk == N - 1);
        const bool valid = lane < nk;
        const int k = kc0 + lane;  // 0-based dynamics knot; K = k + 1 in the reference's numbering
        const int K = k + 1;
        // mode schedule, src/constraints.jl:23-37: K < k_trans-1 -> init mode; K == k_trans-1 ->
        // init mode then jump map; else mode 3.  mode 1 = foot 2 free, mode 2 = foot 1 free.
        const int mode = (K <= kt - 1) ? im : 3;
        const bool jump = (K == kt - 1);
        const bool f1free = (mode == 2);
        const bool f2free = (mode == 1);
        const double* zl = s_z + 20 * (valid ? lane : 0);

        // ---- stage the chunk's slice of Z through LDS (the first one was requested at kernel entry) ----
        {
            if (kc0 != kc_begin) {
                const double* __restrict__ zsrc = Zb + 20 * kc0;
#pragma unroll
                for (int it = 0; it < kStageIters; ++it) zr[it] = zsrc[min(it * kWave + lane, nz - 1)];
                request_cost(kc0);
            }
            wave_lds_sync();  // the previous chunk's drain reads precede this chunk's staging writes
            // (the tile has room for all kStageIters*64 doubles; what lies past the slice is never read)
#pragma unroll
            for (int it = 0; it < kStageIters; ++it) s_z[it * kWave + lane] = zr[it];
            wave_lds_sync();
            if (WITH_C) {
                // c1 = Z[x_1] - x0 (src/constraints.jl:149), c2 = Z[x_N][1:14] - xf[1:14] (:150),
                // c6 = F1y + F2y + mb*g of u_{N-1} (:154), all out of the staged slice
                if (first_chunk && lane < 15) c_put(lane, s_z[lane] - bnd);
                if (last_chunk) {
                    if (lane >= 15 && lane < 29) c_put(lane, s_z[20 * nk + (lane - 15)] - bnd);
                    if (lane == 29) c_put(o_fc, s_z[20 * (nk - 1) + 16] + s_z[20 * (nk - 1) + 18] + mb * g);
                }
            }
            if (kc0 == kc_begin) prefetch_later_workgroup();  // the wave's own first slice (and bnd) have arrived
        }

        QLN_STAMP(1);
        double cos_th = 0.0, cos_tn = 0.0;  // cos(theta) of the lane's knot / of x_N, value phase -> Jacobian phase
        // ============================== value phase (eval_c!) ==================================
        if (WITH_C) {
            double x[15], u[5], xnext[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) x[i] = zl[i];
#pragma unroll
            for (int i = 0; i < 5; ++i) u[i] = zl[15 + i];
#pragma unroll
            for (int i = 0; i < 15; ++i) xnext[i] = zl[20 + i];

            StepConst sc;
            sc.abx = (u[0] + u[2]) / mb;
            sc.aby = (u[1] + u[3]) / mb + g;
            sc.a1x = f1free ? (-u[0] / mf) : 0.0;
            sc.a1y = f1free ? (-u[1] / mf + g) : 0.0;
            sc.a2x = f2free ? (-u[2] / mf) : 0.0;
            sc.a2y = f2free ? (-u[3] / mf + g) : 0.0;
            double xn[15];
            if (!QLN_FLOOR_MODE) {
                rk4_step(x, u, sc, f1free, f2free, Ib, xn);
            } else {
#pragma unroll
                for (int i = 0; i < 15; ++i) xn[i] = x[i];
            }
            if (jump) {  // jump1_map / jump2_map, src/planar_quadruped.jl:250-260
                xn[4] = 0.0;
                xn[6] = 0.0;
                xn[10] = xn[11] = xn[12] = xn[13] = 0.0;
            }
            // per-knot scalar rows: contact (src/constraints.jl:48-91) and clearance (:98-113)
            if (valid) {
                c_put(o_ci + k, init1 ? x[4] : x[6]);
                if (K >= kt) c_put(o_co + (K - kt), init1 ? x[6] : x[4]);
                if (k == N - 2) {  // this lane also holds the terminal knot x_N
                    c_put(o_ci + k + 1, init1 ? xnext[4] : xnext[6]);
                    if (K + 1 >= kt) c_put(o_co + (K + 1 - kt), init1 ? xnext[6] : xnext[4]);
                }
                // dynamics residuals: 15 per knot, knot-major and contiguous in c
                // (src/constraints.jl:14-18); transposed through LDS so the store is coalesced
#pragma unroll
                for (int i = 0; i < 15; ++i) s_c[lane * 15 + i] = xn[i] - xnext[i];
            }
            // clearance rows (src/constraints.jl:98-113): one lane per knot of the slice, one sin() per
            // wave; the terminal knot rides on lane nk of the last chunk
            {
                const bool own = valid || (last_chunk && lane == nk);
                const double* zk = s_z + 20 * (own ? lane : 0);
                double sth;
                if (QLN_FLOOR_MODE) sth = zk[2], cos_th = 1.0;
                else if constexpr (WITH_J) sincos(zk[2], &sth, &cos_th);  // the Jacobian phase needs cos(theta_k) (one call)
                else sth = sin(zk[2]);
                const double cl = zk[1] - lb / 2 * fabs(sth);
                if (own) c_put(o_bp + kc0 + lane, cl);
                if (last_chunk && nk == kWave) {  // wave-uniform: a full last chunk has no lane left for x_N
                    const double* zn = s_z + 20 * nk;
                    double stn;
                    if constexpr (WITH_J) sincos(zn[2], &stn, &cos_tn);
                    else stn = sin(zn[2]);
                    const double cn = zn[1] - lb / 2 * fabs(stn);
                    if (lane == 0) c_put(o_bp + kc0 + nk, cn);
                }
            }
            wave_lds_sync();
            {
                double* __restrict__ dst = Cb + o_dyn + 15 * kc0;
                const int np = nk * 15;
                constexpr int kCIters = (KC * 15 + kWave - 1) / kWave;
                double cr[kCIters];
#pragma unroll
                for (int it = 0; it < kCIters; ++it) cr[it] = s_c[min(it * kWave + lane, KC * 15 - 1)];
#pragma unroll
                for (int it = 0; it < kCIters; ++it) {
                    const int i = it * kWave + lane;
                    if (i < np) {
                        if constexpr (STREAM) __builtin_nontemporal_store(cr[it], &dst[i]);  // streamed like the blocks (-1 %)
                        else dst[i] = cr[it];
                    }
                }
            }
        }

        // ============================== objective + gradient (eval_f, grad_f!) ==================
        if constexpr (WITH_F) {
            eval_objective_chunk(kc0, nk, valid, last_chunk);
            // grad_f! (src/costs.jl:23-34, no d(h l)/dh: quirk Q2): lane = entry of the staged slice, coalesced stores;
            // the slice's last 15 entries are x_{k+1} of the next chunk, or x_N -- the terminal knot -- in the last one
            if constexpr (kFullF) {
                double* __restrict__ Gb = G + (int64_t)b * P.z_stride + 20 * kc0;
                const int ng = 20 * nk + (last_chunk ? 15 : 0);
#pragma unroll
                for (int it = 0; it < kStageIters; ++it) {
                    const int e = min(it * kWave + lane, ng - 1);
                    const int kk = e / 20;
                    const double lin = gD[it] * s_z[e] + gd[it];
                    const double hk = s_z[min(20 * kk + 19, nz - 1)];
                    const double gv = (kk < nk) ? hk * lin : lin;
                    if (it * kWave + lane < ng) Gb[e] = gv;
                }
            }
        }

        QLN_STAMP(2);
        // ============================== Jacobian phase (jac_c!) ================================
        if (WITH_J) {
            // nothing of the value phase is kept: x, u are re-read from the staged slice, so the two
            // phases' register sets do not add up (a spill reload would cost a vmcnt(0) drain)
            wave_lds_sync();
            double x[14], F1x, F1y, F2x, F2y, h;
#pragma unroll
            for (int i = 0; i < 14; ++i) x[i] = zl[i];
            F1x = zl[15];
            F1y = zl[16];
            F2x = zl[17];
            F2y = zl[18];
            h = zl[19];
            // clearance d/dtheta (src/constraints.jl:269-273; theta == 0 takes the + branch): one lane
            // per knot of the slice, one cos() per wave; the terminal knot x_N rides on lane nk of the
            // last chunk
            {
                const bool own = valid || (last_chunk && lane == nk);
                const double th = s_z[20 * (own ? lane : 0) + 2];
                const double cth = WITH_C ? cos_th : cos(th);
                if (own) Vb[dyn_blocks + kc0 + lane] = (th > 0) ? (-lb / 2 * cth) : (lb / 2 * cth);
                if (last_chunk && nk == kWave) {  // wave-uniform: a full last chunk has no lane left for x_N
                    const double tn = s_z[20 * nk + 2];
                    const double ctn = WITH_C ? cos_tn : cos(tn);
                    if (lane == 0) Vb[dyn_blocks + kc0 + nk] = (tn > 0) ? (-lb / 2 * ctn) : (lb / 2 * ctn);
                }
            }
            wave_lds_sync();
            if constexpr (!NNZ) {
                // structural zeros of the tile: written here, never touched by the value writes below
                const double2 zero2 = make_double2(0.0, 0.0);
#pragma unroll
                for (int it = 0; it < (T * kBlk / 2 + kWave - 1) / kWave; ++it) {
                    const int i = it * kWave + lane;
                    if (i < T * kBlk / 2) s_j2[i] = zero2;
                }
            }
            // ---- base quantities of the step block's 85 non-zeros (closed form, see header) ----
            QLN_STEP_BASE();

            QLN_STAMP(3);
            if constexpr (NNZ) {
                // ---- structural format: the chunk's compact blocks are one contiguous run of vals ----
                // Every lane writes the values of its knot's pattern (71 / 56 / 57 of them) behind those of
                // the lane before it; lanes of different contact modes take different branches (at most
                // three per chunk: before, at and after the transition knot).
                // T = 0: the whole chunk is one LDS image; T > 0: sub-tiles of T knots, each emitted by its T lanes and
                // drained before the next (a smaller tile: more waves per CU, the emission's instructions issued once per sub-tile).
                constexpr int kSub = T > 0 ? T : KC;
#pragma unroll 1
                for (int t0 = 0; t0 < nk; t0 += kSub) {
                const int nkt = min(kSub, nk - t0);
                const int g0 = step_block_offset(kc0 + t0, N, kt);
                const int g1 = step_block_offset(kc0 + t0 + nkt, N, kt);
                const int p0 = g0 & 1;  // the LDS image starts at the parity of its global offset
                if (valid && lane >= t0 && lane < t0 + nkt && !QLN_FLOOR_MODE) {
                    // One pass over the pattern of the problem's contact mode (71 entries, column-major), the same
                    // instructions for every lane.  Lanes whose knot has a sparser pattern (mode 3: 57, transition
                    // knot: 56) pull their write pointer back by one slot after every entry their pattern lacks, so
                    // that such an entry lands where the lane's next real entry overwrites it (a wave's DS writes
                    // execute in order).  Only the transition knot's pattern ends early -- the masked rows 11-15 of
                    // column 19 -- and those five writes are predicated.
                    double* jr = s_j + p0 + (step_block_offset(k, N, kt) - g0);
                    const int decF = (mode == 3) ? 1 : 0, decJ = jump ? 1 : 0, decFJ = decF + decJ;
                    auto emit = [&](auto catc) {
                        constexpr int CB = decltype(catc)::value;  // contact category of the problem: 0 or 1
                        constexpr int CJ = CB + 3;                 // the same mode followed by the jump
#define JW(row, col, val)                                                                   \
    if constexpr (step_entry_present(CB, row, col)) {                                       \
        constexpr int pos_ = step_entry_pos(CB, row, col);                                  \
        constexpr bool inF_ = step_entry_present(2, row, col);                              \
        constexpr bool inJ_ = step_entry_present(CJ, row, col);                             \
        if constexpr (!inJ_ && step_entry_pos(CJ, row, col) == step_nnz(CJ)) {              \
            if (!jump) jr[pos_] = (val); /* nothing of the jump pattern follows */          \
        } else {                                                                            \
            jr[pos_] = (val);                                                               \
        }                                                                                   \
        if constexpr (!inF_ && !inJ_) jr -= decFJ;                                          \
        else if constexpr (!inF_) jr -= decF;                                               \
        else if constexpr (!inJ_) jr -= decJ;                                               \
    }
                        QLN_STEP_ENTRIES();
#undef JW
                    };
                    if (im == 1) emit(std::integral_constant<int, 0>{});
                    else emit(std::integral_constant<int, 1>{});
                }
                wave_lds_sync();
                QLN_STAMP(4);
                // LDS double p0 + i <-> vals[g0 + i]; j_off is even, so p0 = g0 & 1 lines the 16-byte pieces up on both sides
                drain_run<STREAM>(s_j, p0, g1 - g0, Vb + g0, lane);
                wave_lds_sync();
                QLN_STAMP(5);
                }  // sub-tiles
            } else {
            // ---- assemble T knots at a time in LDS and stream them out ---------------------------
            const int nt = (nk + T - 1) / T;
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                const int kb = kc0 + t * T;          // first knot of the sub-tile
                const int nkt = min(T, nk - t * T);  // knots in it
                if (valid && (lane / T) == t) {
                    double* jr = s_j + (lane - t * T) * kBlk;
#define JW(row, col, val) jr[(row) + 15 * (col)] = (val)
                    // opaque to loop-invariant code motion (see above)
                    asm volatile("" : "+v"(wAt), "+v"(wBt), "+v"(wAw));
                    QLN_STEP_ENTRIES();
#undef JW
                }
                wave_lds_sync();
                {
                    // nkt*300 contiguous doubles of the problem; 16 B per lane, 1 KiB per wave instruction.
                    constexpr int kStep = kWave;  // double2 between consecutive wave instructions
                    double2* dst = reinterpret_cast<double2*>(Vb + (int64_t)kBlk * kb) + lane;
                    const int np = nkt * (kBlk / 2);
                    constexpr int kPieces = T * kBlk / 2;   // 16-byte pieces in a full tile
                    constexpr int kFull = kPieces / kWave;  // unpredicated wave instructions
                    if (nkt == T) {
                        drain_full<kFull, kStep, STREAM>(lds_offset(s_j2 + lane), dst);
                        if (kPieces % kWave) {
                            if (kFull * kWave + lane < kPieces) block_store<STREAM>(dst + kFull * kStep, s_j2[kFull * kWave + lane]);
                        }
                    } else {
                        // last, partial sub-tile of a chunk: whole 6-instruction batches, then a predicated tail
                        int it = 0;
#pragma unroll 1
                        for (; (it + 6) * kWave <= np; it += 6) drain6<kStep, STREAM>(lds_offset(s_j2 + it * kWave + lane), dst + (int64_t)it * kStep);
#pragma unroll 1
                        for (; it * kWave + lane < np; ++it) block_store<STREAM>(dst + (int64_t)it * kStep, s_j2[it * kWave + lane]);
                    }
                }
                wave_lds_sync();
                QLN_STAMP(4 + min(t, 10));
            }
            }  // dense blocks
        }
    }
    if constexpr (WITH_F) {
        if (lane == 0) F[b] = J_obj;
    }
    if constexpr (!SPLIT) {
        // the prefetch loads' only reader (see above): never true (N >= 2), placed where the wave has nothing left to do
        if (N < 0 && (pf_acc[0] | pf_acc[1] | pf_acc[2] | pf_acc[3] | pf_acc[4]) == 0x5eedbeef) {
            if (WITH_C) Cb[0] = 0.0;
            else Vb[0] = 0.0;
        }
    }
    QLN_STAMP(15);
}

// constants only (qln_jacobian_init_constants)
__global__ __launch_bounds__(kWave) void k_jacobian_constants(BatchParams P, double* __restrict__ V) {
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;
    const int N = P.N;
    const int kt = P.desc[b].k_trans;
    const int dyn_blocks = (P.jac_format == QLN_JAC_FORMAT_STRUCTURAL) ? step_block_offset(N - 1, N, kt) : kBlk * (N - 1);
    double* Vc = V + P.desc[b].j_off + dyn_blocks + N;
    const int n_const = 435 + 15 * (N - 1) + 3 * N - kt + 3;
    for (int i = lane; i < n_const; i += kWave) {
        double v;
        if (i < 225) {
            v = (i % 15 == i / 15) ? 1.0 : 0.0;
        } else if (i < 435) {
            const int j = i - 225;
            v = (j % 14 == j / 14) ? 1.0 : 0.0;
        } else if (i < 435 + 15 * (N - 1)) {
            v = -1.0;
        } else {
            v = 1.0;
        }
        Vc[i] = v;
    }
}

// Objective (src/costs.jl:6-16): objective_term / add_terms_in_order are defined above the fused kernel, which uses them too.
// One wavefront per problem, 64 knots per pass: the pass's slice of Z is read with coalesced loads (all in flight before
// the first wait) into LDS -- the only LDS the kernel uses, 10 KB, so that a CU holds 14 waves -- and lane = knot
// forms its term straight from there.
__global__ __launch_bounds__(kWave) void k_objective(BatchParams P, const double* __restrict__ Z, double* __restrict__ F) {
    constexpr int kSlice = kWave * 21;  // a knot's 20 entries 21 doubles apart (LDS banks: lane = knot reads them)
    __shared__ double s_z[kSlice];
    __shared__ double s_term[kWave];
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;
    const int N = P.N;
    const int n_nlp = 20 * N - 5;
    const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;
    const double* __restrict__ cost = P.cost + (P.cost_batch > 1 ? (int64_t)b * N * 41 : 0);
    double J = 0.0;
    for (int k0 = 0; k0 < N; k0 += kWave) {
        const int nk = min(kWave, N - k0);
        const int ne = min(20 * nk, n_nlp - 20 * k0);  // entries of Z in this pass (x_N has no controls)
        double zr[20];
#pragma unroll
        for (int it = 0; it < 20; ++it) zr[it] = Zb[20 * k0 + min(it * kWave + lane, ne - 1)];  // clamped, not predicated
        wave_lds_sync();  // the previous pass's readers are done
#pragma unroll
        for (int it = 0; it < 20; ++it) s_z[it * kWave + lane + (it * kWave + lane) / 20] = zr[it];
        wave_lds_sync();
        const int kl = min(lane, nk - 1);
        const int k = k0 + kl;
        const double term = objective_term(s_z + 21 * kl, cost + (int64_t)k * 41, k < N - 1);
        J = add_terms_in_order(J, (lane < nk) ? term : 0.0, s_term, lane);
    }
    if (lane == 0) F[b] = J;
}

// The same for the common case of ONE cost table shared by the whole batch (cost_batch == 1) and N <= 64: persistent
// one-wave workgroups walk the batch.  lane = knot is the same knot in every problem, so its 41-double cost record is read
// ONCE into registers; a problem's slice of Z arrives as 16-byte pieces (the next problem's in flight while this one is
// summed) and goes through LDS only to turn entry-major into knot-major (a knot's 20 entries 21 doubles apart: lane = knot
// reads with a stride of 20 doubles would hit 8 of the 64 banks); the in-order sum of the knots' terms is a chain of
// adds whose operand is read out of the lane that holds it (v_readlane), not 64 LDS broadcasts per lane.  Round 2's
// version kept the table and the sum in LDS (552 LDS clocks per problem and CU against the 490 that 6.4 KB of Z cost at
// the HBM peak; this one issues 132), read Z in 8-byte pieces and had one slice in flight: 0.097-0.105 ms = 50-54 % of peak
// at config 3; this one 0.078-0.081 ms = 64-66 %, what a library reduction over the same bytes reaches
// (profiles/r03_objective_variants.txt: the sum in LDS, one or three slices in flight, three waves per SIMD are all slower).
// KI = 16-byte load instructions per slice (ceil((n_nlp - 1) / 128): 1 .. 10), DEPTH = slices in flight per wave,
// W = waves per SIMD the register budget is sized for; LANESUM = false is the tuning build's LDS-sum variant.
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));  // a problem's slice is only 8-byte aligned (n_nlp is odd)
template <int KI, bool LANESUM, int DEPTH, int W>
__global__ __launch_bounds__(kWave, W) void k_objective_shared(BatchParams P, const double* __restrict__ Z, double* __restrict__ F) {
    extern __shared__ double s_dyn[];  // [N][21], slack up to the KI * 128 entries the loads cover (+ 64 for the LDS sum)
    double* s_z = s_dyn;
    const int lane = threadIdx.x;
    const int N = P.N;
    const int n_nlp = 20 * N - 5;
    double* s_term = s_dyn + (KI * 2 * kWave * 21) / 20 + 2;
    const int npieces = (n_nlp - 1) / 2;  // n_nlp is odd: complete 16-byte pieces, then one double; KI = ceil(npieces / 64)
    const int kl = min(lane, N - 1);
    double D[20], d[20];
    {
        const double* __restrict__ rec = P.cost + 41 * kl;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            D[i] = rec[i];
            d[i] = rec[20 + i];
        }
    }
    const double c40 = P.cost[41 * kl + 40];
    const int stride = gridDim.x;
    // No branch between a request and its use: every load is issued with a clamped address (a wave past the end of the
    // batch re-reads the last problem, a lane past the end of the slice its last piece), so that the waits the compiler
    // places count loads (vmcnt(n): the OTHER slot's stay in flight) instead of draining the queue at a block boundary.
    double2_a8 zr[DEPTH][KI];
    double ztail[DEPTH];
    auto request = [&](int bb, int slot) {
        const double* __restrict__ Zb = Z + (int64_t)min(bb, P.B - 1) * P.z_stride;
#pragma unroll
        for (int it = 0; it < KI; ++it)
            zr[slot][it] = *reinterpret_cast<const double2_a8*>(Zb + 2 * min(it * kWave + lane, npieces - 1));
        ztail[slot] = Zb[n_nlp - 1];
    };
    auto consume = [&](int bb, int slot) {
        wave_lds_sync();  // the previous problem's readers are done
#pragma unroll
        for (int it = 0; it < KI; ++it) {  // unpredicated: a lane past the end of the slice writes into the slack behind it
            const int e = 2 * (it * kWave + lane);
            s_z[e + e / 20] = zr[slot][it].x;
            s_z[e + 1 + (e + 1) / 20] = zr[slot][it].y;
        }
        s_z[n_nlp - 1 + (n_nlp - 1) / 20] = ztail[slot];  // every lane, the same value: after the slack writes, which reach here
        wave_lds_sync();
        request(bb + DEPTH * stride, slot);  // into the registers just emptied
        const double term = objective_term_regs(s_z + 21 * kl, D, d, c40, kl < N - 1);
        double J;
        if constexpr (LANESUM) J = add_lane_terms_in_order(0.0, (lane < N) ? term : 0.0, N);
        else J = add_terms_in_order(0.0, (lane < N) ? term : 0.0, s_term, lane);
        if (lane == 0 && bb < P.B) F[bb] = J;
    };
    int b = blockIdx.x;
#pragma unroll
    for (int sl = 0; sl < DEPTH; ++sl) {
        // the prologue's requests in the order the loop issues them: the waits inside the loop are placed for both
        __builtin_amdgcn_sched_barrier(0);
        request(b + sl * stride, sl);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (; b < P.B; b += DEPTH * stride) {
#pragma unroll
        for (int sl = 0; sl < DEPTH; ++sl) consume(b + sl * stride, sl);
    }
}

// Objective gradient (src/costs.jl:23-34; no d(h*l)/dh term -- quirk Q2).  Flat over the entries of the whole batch
// (a problem's 20N-5 entries do not fill whole workgroups): 1024 consecutive entries per workgroup, four per thread,
// loads issued before the first store.
constexpr int kGradU = 4;  // entries per thread (8, and non-temporal stores, measured: no gain)
__global__ __launch_bounds__(256) void k_objective_gradient(BatchParams P, const double* __restrict__ Z,
                                                           double* __restrict__ G, int64_t total, int ntiles) {
    const int N = P.N;
    const int n_nlp = 20 * N - 5;
    const int tile = xcd_contiguous_index(blockIdx.x, ntiles);
    if (tile >= ntiles) return;
    constexpr int kU = kGradU;
    double z[kU], D[kU], d[kU], h[kU];
    int64_t at[kU];
    bool stage[kU];
    // (problem, entry) of the thread's first element by one division, of the next ones by stepping
    const int64_t e0 = (int64_t)tile * (kU * 256) + (int)threadIdx.x;
    const int b0 = (int)(e0 / n_nlp);
    const int i0 = (int)(e0 - (int64_t)b0 * n_nlp);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        int b = b0, i = i0 + u * 256;
        while (i >= n_nlp) {
            i -= n_nlp;
            ++b;
        }
        if (b >= P.B) {  // past the end of the batch: clamp to the last entry (the store below is predicated)
            b = P.B - 1;
            i = n_nlp - 1;
        }
        const int k = i / 20, j = i - 20 * k;
        const double* __restrict__ Zb = Z + (int64_t)b * P.z_stride;
        const double* ck = P.cost + (P.cost_batch > 1 ? (int64_t)b * N * 41 : 0) + (int64_t)k * 41;
        at[u] = (int64_t)b * P.z_stride + i;
        stage[u] = k < N - 1;
        z[u] = Zb[i];
        D[u] = ck[j];        // Q*x + q for j < 15, R*u + r for the controls: record slots [0,20) and [20,40)
        d[u] = ck[20 + j];
        h[u] = Zb[min(20 * k + 19, n_nlp - 1)];  // h_k; not used for the terminal knot
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
        const int64_t e = (int64_t)tile * (kU * 256) + u * 256 + (int)threadIdx.x;
        if (e < total) {
            const double lin = D[u] * z[u] + d[u];
            G[at[u]] = stage[u] ? h[u] * lin : lin;
        }
    }
}

// grad_f! for a batch that shares one cost table (cost_batch == 1), N <= 64 (src/costs.jl:26-31; no d(h l)/dh: quirk Q2):
// persistent one-wave workgroups, the design of k_objective_shared.  A problem's gradient has the layout of its Z, so a
// lane works on the SAME entries of every problem -- 16-byte piece it * 64 + lane of the slice, whose two entries lie in
// one knot (20 is even) -- and holds their D and d (Q*x + q / R*u + r: record slots [0, 20) and [20, 40)) in registers for
// the whole batch.  Only the step lengths h_k go through LDS (the terminal knot's "h" is a 1.0 written once: 1.0 * x == x).
// Nothing is predicated: a lane past the end of the slice works on the last piece again and stores the same two values to
// the same address, and a lane whose piece holds no h writes a spare LDS slot -- no block boundary between a request and
// its use, so the waits count memory operations instead of draining the queue.  Round 2's version (table in LDS, 8-byte
// pieces, 20 load iterations issued where 13 are needed): 0.164-0.174 ms = 60-64 % of peak at config 3; this one
// 0.150-0.153 ms = 68-70 % with one slice in flight per wave and 8 waves per CU -- two slices, or 12 / 16 waves per CU with
// smaller register budgets, change nothing beyond the scatter; three slices are slower (profiles/r03_gradient_variants.txt).
template <int KI, int DEPTH, int W>
__global__ __launch_bounds__(kWave, W) void k_objective_gradient_shared(BatchParams P, const double* __restrict__ Z,
                                                                        double* __restrict__ G) {
    __shared__ double s_h[kWave + 2];  // h_k, k < N - 1 <= 63; [N - 1] = 1.0; [65] = spare
    const int lane = threadIdx.x;
    const int N = P.N;
    const int n_nlp = 20 * N - 5;
    const int npieces = (n_nlp - 1) / 2;  // n_nlp is odd: complete 16-byte pieces, then one double (x_N[15]: its own knot's entry 14)
    double D0[KI], d0[KI], D1[KI], d1[KI];
    int kk[KI], hs[KI];
#pragma unroll
    for (int it = 0; it < KI; ++it) {
        const int e = 2 * min(it * kWave + lane, npieces - 1);
        const int k = e / 20, j = e - 20 * k;
        const double* __restrict__ rec = P.cost + 41 * k;
        D0[it] = rec[j], d0[it] = rec[20 + j];
        D1[it] = rec[j + 1], d1[it] = rec[21 + j];
        kk[it] = k;
        hs[it] = (j == 18) ? k : kWave + 1;  // the piece's second entry is u_k[5] = h_k
    }
    const double Dt = P.cost[41 * (N - 1) + 14], dt = P.cost[41 * (N - 1) + 34];
    if (lane == 0) s_h[N - 1] = 1.0;
    const int stride = gridDim.x;
    double2_a8 zr[DEPTH][KI];
    double ztail[DEPTH];
    auto request = [&](int bb, int slot) {
        const double* __restrict__ Zb = Z + (int64_t)min(bb, P.B - 1) * P.z_stride;
#pragma unroll
        for (int it = 0; it < KI; ++it)
            zr[slot][it] = *reinterpret_cast<const double2_a8*>(Zb + 2 * min(it * kWave + lane, npieces - 1));
        ztail[slot] = Zb[n_nlp - 1];
    };
    auto consume = [&](int bb, int slot) {
        wave_lds_sync();  // the previous problem's readers are done
#pragma unroll
        for (int it = 0; it < KI; ++it) s_h[hs[it]] = zr[slot][it].y;
        wave_lds_sync();
        double* __restrict__ Gb = G + (int64_t)min(bb, P.B - 1) * P.z_stride;
        double2_a8 out[KI];
#pragma unroll
        for (int it = 0; it < KI; ++it) {
            const double h = s_h[kk[it]];
            out[it].x = h * (D0[it] * zr[slot][it].x + d0[it]);
            out[it].y = h * (D1[it] * zr[slot][it].y + d1[it]);
        }
        const double gt = Dt * ztail[slot] + dt;
        request(bb + DEPTH * stride, slot);  // into the registers just emptied
#pragma unroll
        for (int it = 0; it < KI; ++it)
            __builtin_nontemporal_store(out[it], reinterpret_cast<double2_a8*>(Gb + 2 * min(it * kWave + lane, npieces - 1)));
        Gb[n_nlp - 1] = gt;  // every lane, the same value
    };
    int b = blockIdx.x;
#pragma unroll
    for (int sl = 0; sl < DEPTH; ++sl) {
        __builtin_amdgcn_sched_barrier(0);  // the prologue's requests in the order the loop issues them
        request(b + sl * stride, sl);
    }
    __builtin_amdgcn_sched_barrier(0);
    for (; b < P.B; b += DEPTH * stride) {
#pragma unroll
        for (int sl = 0; sl < DEPTH; ++sl) consume(b + sl * stride, sl);
    }
}

// Initial guess of the reference's notebook (src/main.ipynb:181-198) packed like packZ (src/nlp.jl:94-102),
// with U = Uref of reference_trajectory (src/ref_traj.jl:19-34): the step BEFORE the hot path, built where the
// evaluator will read it.  One thread per entry of Z; same operation order as the notebook, so the result is
// bit-identical to the host generator (quadruped_landing_amd/problem_gen.py).
__global__ __launch_bounds__(256) void k_initial_guess(BatchParams P, double* __restrict__ Z) {
    const int N = P.N;
    const int n_nlp = 20 * N - 5;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    const int i = blockIdx.y * blockDim.x + threadIdx.x;
    if (b >= P.B || i >= n_nlp) return;
    const int kt = P.desc[b].k_trans, im = P.desc[b].init_mode;
    const double* x0 = P.bnd + (int64_t)b * 30;
    const double* xf = x0 + 15;
    const int k = i / 20, j = i - 20 * k;  // 0-based knot, slot
    const int K = k + 1;                   // the notebook's 1-based k
    double v;
    if (j < 14) {
        // Xguess[k] = xinit + (xterm - xinit) / (k_trans - 1) * (k - 1) for k <= k_trans, else xterm[1:14]
        v = (K <= kt) ? x0[j] + (xf[j] - x0[j]) / (double)(kt - 1) * (double)(K - 1) : xf[j];
    } else if (j == 14) {
        // Xguess[k+1][end] = Xguess[k][end] + (k < k_trans ? 0.001 : 0.02), sequentially from Xguess[1][end]
        v = x0[14] + (xf[14] - x0[14]) / (double)(kt - 1) * 0.0;
        for (int q = 1; q < K; ++q) v = v + ((q < kt) ? 0.001 : 0.02);
    } else if (j == 19) {
        v = (K <= kt - 1) ? 0.001 : 0.02;  // Uref[5, :]
    } else {
        // Uref[2 or 4, 1:k_trans-1] = -mb*g ; Uref[2, k_trans:end] = Uref[4, k_trans:end] = -mb*g/2
        const int lead = (im == 1) ? 16 : 18, other = (im == 1) ? 18 : 16;
        const bool before = (K <= kt - 1);
        v = 0.0;
        if (j == lead) v = before ? (-P.mb * P.g) : (-P.mb * P.g / 2);
        if (j == other) v = before ? 0.0 : (-P.mb * P.g / 2);
    }
    Z[(int64_t)b * P.z_stride + i] = v;
}

// Constraint violation per problem, as Ipopt reports it for the reference's solve (src/main.ipynb:712): the largest
// violation of the bounds of src/nlp.jl:66-69 -- |c_i| over the equality rows, max(0, -c_i) over the clearance rows
// (lb = 0, ub = +Inf).  One wave per problem; NaN anywhere in the problem's c gives NaN.
__global__ __launch_bounds__(kWave) void k_constraint_violation(BatchParams P, const double* __restrict__ C,
                                                               double* __restrict__ viol) {
    const int lane = threadIdx.x;
    const int b = xcd_contiguous_index(blockIdx.x, P.B);
    if (b >= P.B) return;
    const ProblemDesc pd = P.desc[b];
    const int N = P.N;
    const int m = 18 * N - pd.k_trans + 16;
    const int m_eq = m - N;
    const double* __restrict__ cb = C + pd.c_off;
    double v = 0.0;
    bool bad = false;
    for (int i = lane; i < m; i += kWave) {
        const double x = cb[i];
        bad = bad || (x != x);
        v = fmax(v, (i < m_eq) ? fabs(x) : fmax(-x, 0.0));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v = fmax(v, __shfl_xor(v, off, kWave));
        const int other_bad = __shfl_xor((int)bad, off, kWave);  // unconditional: every lane must take part
        bad = bad | (other_bad != 0);
    }
    if (lane == 0) viol[b] = bad ? __builtin_nan("") : v;
}

// LQR cost records of the notebook's objective (src/main.ipynb:158-161): obj[k] = LQRCost(Q, R, Xref[k], Uref[k])
// for k < N, obj[N] = LQRCost(Qf, R*0, Xref[N], Uref[1]), with Xref/Uref of reference_trajectory
// (src/ref_traj.jl:6-39) and LQRCost of src/quadratic_cost.jl:33-42, built on the device: one thread per
// (problem, knot).  Same operation order as the host builder (quadruped_landing_amd/quadratic_cost.py), so
// the records are bit-identical.  cost_batch == 1 builds one table from problem 0's descriptors.
__global__ __launch_bounds__(256) void k_lqr_cost(BatchParams P, const double* __restrict__ QRQf /*15+5+15*/, double dt,
                                                 double* __restrict__ cost, int cost_batch) {
    const int N = P.N;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cost_batch * N) return;
    const int b = t / N, k = t - b * N;
    const int K = k + 1;
    const int kt = P.desc[b].k_trans, im = P.desc[b].init_mode;
    const double* xterm = P.bnd + (int64_t)b * 30 + 15;
    const bool last = (k == N - 1);
    const double* Qd = last ? QRQf + 20 : QRQf;
    // Xref[:, k] = xterm with the clock slot = range(0, dt*(N-1), length=N)[k]  (numpy.linspace on the host)
    double xr[15];
    for (int i = 0; i < 14; ++i) xr[i] = xterm[i];
    const double stop = dt * (double)(N - 1);
    xr[14] = (N > 1) ? (last ? stop : (double)k * (stop / (double)(N - 1))) : 0.0;
    // Uref[:, k] (the terminal record uses Uref[1] and R*0)
    double ur[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    {
        const int Ku = last ? 1 : K;
        const bool before = (Ku <= kt - 1);
        const int lead = (im == 1) ? 1 : 3, other = (im == 1) ? 3 : 1;
        ur[lead] = before ? (-P.mb * P.g) : (-P.mb * P.g / 2);
        ur[other] = before ? 0.0 : (-P.mb * P.g / 2);
        ur[4] = before ? 0.001 : 0.02;
    }
    double* out = cost + ((int64_t)b * N + k) * 41;
    double a = 0.0, bb = 0.0;
    for (int i = 0; i < 15; ++i) {
        out[i] = Qd[i];
        out[20 + i] = (-Qd[i]) * xr[i];                    // q = -Q * xf
        const double tq = (0.5 * (Qd[i] * xr[i])) * xr[i];
        a = (i == 0) ? tq : a + tq;
    }
    for (int i = 0; i < 5; ++i) {
        const double Ri = last ? QRQf[15 + i] * 0 : QRQf[15 + i];
        out[15 + i] = Ri;
        out[35 + i] = (-Ri) * ur[i];                       // r = -R * uf
        const double tr = (0.5 * (Ri * ur[i])) * ur[i];
        bb = (i == 0) ? tr : bb + tr;
    }
    out[40] = a + bb;                                      // c = 0.5*xf'Q*xf + 0.5*uf'R*uf
}

// Opt-in extension, NOT on the reference's path: the leg-length ("kinematic") rows the reference carries only as
// commented-out code (src/constraints.jl:115-138: d[2k-1] = norm(pb - p1), d[2k] = norm(pb - p2); bounds
// 0 <= d <= l1 + l2 + lb/2, src/nlp.jl:60,70).  Values as that source defines them; the Jacobian is the mathematically
// correct one, d/d(pb) = (pb - p_i)/|pb - p_i| and d/d(p_i) = -(pb - p_i)/|pb - p_i| in the slots of pb, p1, p2 (the
// commented Jacobian, :276-288, reads x[7:8] / x[9:10] -- y2, vbx / vby, omega -- which are not the feet).
// One thread per (problem, knot): 6 doubles in, 2 + 8 out.
__global__ __launch_bounds__(256) void k_kinematic_rows(BatchParams P, const double* __restrict__ Z, double* __restrict__ D,
                                                       double* __restrict__ JV) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)P.B * P.N) return;
    const int b = (int)(t / P.N), k = (int)(t - (int64_t)b * P.N);
    const double* x = Z + (int64_t)b * P.z_stride + 20 * k;
    const double xb = x[0], yb = x[1];
    const double d1x = xb - x[3], d1y = yb - x[4], d2x = xb - x[5], d2y = yb - x[6];
    const double n1 = sqrt(d1x * d1x + d1y * d1y), n2 = sqrt(d2x * d2x + d2y * d2y);
    D[2 * t] = n1;
    D[2 * t + 1] = n2;
    if (JV) {
        double* j = JV + 8 * t;
        j[0] = d1x / n1, j[1] = d1y / n1, j[2] = -d1x / n1, j[3] = -d1y / n1;
        j[4] = d2x / n2, j[5] = d2y / n2, j[6] = -d2x / n2, j[7] = -d2y / n2;
    }
}

// Opt-in friction pyramid |F_x| <= mu F_y of the feet that stand on the ground (qln_eval_friction_cone): one thread per
// (problem, dynamics knot).  Which feet stand is the mode schedule of the dynamics rows (src/constraints.jl:23-37).
__global__ __launch_bounds__(256) void k_friction_rows(BatchParams P, const double* __restrict__ Z, double mu,
                                                      double* __restrict__ D, double* __restrict__ JV) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = P.N - 1;
    if (t >= (int64_t)P.B * nk) return;
    const int b = (int)(t / nk), k = (int)(t - (int64_t)b * nk);
    const ProblemDesc pd = P.desc[b];
    const int mode = (k + 1 <= pd.k_trans - 1) ? pd.init_mode : 3;
    const bool on1 = mode != 2, on2 = mode != 1;  // mode 1: foot 2 in flight; mode 2: foot 1 in flight
    const double* u = Z + (int64_t)b * P.z_stride + 20 * k + 15;
    double* d = D + 4 * t;
    d[0] = on1 ? mu * u[1] - u[0] : 0.0;
    d[1] = on1 ? mu * u[1] + u[0] : 0.0;
    d[2] = on2 ? mu * u[3] - u[2] : 0.0;
    d[3] = on2 ? mu * u[3] + u[2] : 0.0;
    if (JV) {
        double* j = JV + 8 * t;
        j[0] = on1 ? -1.0 : 0.0, j[1] = on1 ? mu : 0.0, j[2] = on1 ? 1.0 : 0.0, j[3] = on1 ? mu : 0.0;
        j[4] = on2 ? -1.0 : 0.0, j[5] = on2 ? mu : 0.0, j[6] = on2 ? 1.0 : 0.0, j[7] = on2 ? mu : 0.0;
    }
}

// Dynamic LDS of the structural-format instantiations (bytes): the longest run of vals a sub-tile of `sub` knots can be
// in this batch -- a chunk's run grows with k_trans (71 values per knot before the transition, 57 after), so the batch's
// largest k_trans bounds it -- plus the parity slot, and never less than what aliases the tile: the staged slice of Z
// (written unpredicated in whole wave-rows), the residual stage and, for qln_eval_all, the objective terms.
inline size_t nnz_lds_bytes(const BatchParams& p, int KC, int sub, bool with_f) {
    const int N = p.N, kt = std::min(std::max(p.kt_max, 1), N + 1);
    int longest = 0;
    for (int kc0 = 0; kc0 < N - 1; kc0 += KC) {
        const int nk = std::min(KC, N - 1 - kc0);
        for (int t0 = 0; t0 < nk; t0 += sub) {
            const int nkt = std::min(sub, nk - t0);
            // x of the run's knots lie before the transition knot at the batch's largest k_trans (71 values each), and no
            // problem of the batch has more of them; every other knot has at most 57: 57 nkt + 14 x bounds every problem's run
            const int x = std::min(nkt, std::max(0, std::min(kt - 2, N - 1) - (kc0 + t0)));
            longest = std::max(longest, 57 * nkt + 14 * x);
        }
    }
    const int z_slice = KC * 20 + 15, c_stage = (z_slice + 1) & ~1;
    int need = std::max(c_stage + KC * 15 + (with_f ? kWave : 0), ((z_slice + kWave - 1) / kWave) * kWave);
    need = std::max(need, longest + 2);
    return (size_t)((need + 1) & ~1) * sizeof(double);
}

#ifdef QLN_TUNING
inline void apply_floor_mode() {
    static const int done = [] {
        const char* e = getenv("QLN_FLOOR");
        const int v = e ? atoi(e) : 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_floor_mode), &v, sizeof v);
        return 1;
    }();
    (void)done;
}
#endif

// bits 8.. of the kernel's flags: how many problems ahead on its XCD a workgroup prefetches into L2 (0 = off).  The knob
// builds (make tuning / prefetchknob) let QLN_PREFETCH_AHEAD override the caller's choice (bench/prefetch_ahead.py).
inline uint32_t prefetch_flags(uint32_t ahead_default, uint32_t what_default = 0u) {
#if defined(QLN_TUNING) || defined(QLN_PREFETCH_KNOB)
    static const int ahead = [] {
        const char* e = getenv("QLN_PREFETCH_AHEAD");
        return e ? atoi(e) : -1;
    }();
    static const int mask = [] {  // QLN_PREFETCH_MASK: 1 = slice of Z, 2 = boundary vectors, 4 = descriptor
        const char* e = getenv("QLN_PREFETCH_MASK");
        return e ? atoi(e) : -1;
    }();
    if (ahead >= 0) ahead_default = (uint32_t)ahead;
    if (mask >= 0) what_default = ((mask & 1) ? 0u : kPrefetchNoZ) | ((mask & 2) ? kPrefetchBnd : 0u) | ((mask & 4) ? kPrefetchDesc : 0u);
#endif
    return (ahead_default << 8) | what_default;
}

template <int T, int KC, int W, bool NNZ = false, bool SPLIT = false>
hipError_t launch_cj_t(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c, double* vals,
                       uint32_t flags, hipStream_t stream, uint32_t prefetch_ahead = 0) {
    dim3 grid(xcd_grid(SPLIT ? nb * ((p.N - 2) / KC + 1) : nb)), block(kWave);
#ifdef QLN_TUNING
    apply_floor_mode();
    // tuning build only: extra (unused) dynamic LDS per workgroup lowers the number of resident waves
    static const unsigned pad = [] {
        const char* e = getenv("QLN_PAD_LDS");
        return e ? (unsigned)atoi(e) : 0u;
    }();
#else
    constexpr unsigned pad = 0;
#endif
    if (!SPLIT) flags = (flags & 0xffu) | prefetch_flags(prefetch_ahead);
    // outputs larger than the caches are streamed (non-temporal stores); small ones stay cacheable
    const bool stream_out = (int64_t)nb * (p.N - 1) * (NNZ ? 71 : kBlk) * 8 > ((int64_t)512 << 20);
    auto go = [&](auto with_c, auto with_j, auto streamed) {
        constexpr bool WC = decltype(with_c)::value, WJ = decltype(with_j)::value, ST = decltype(streamed)::value;
        const unsigned lds = (NNZ ? (unsigned)nnz_lds_bytes(p, KC, T > 0 ? T : KC, false) : 0u) + ((WC && WJ) ? pad : 0u);
        hipLaunchKernelGGL((k_constraint_jacobian<T, KC, W, WC, WJ, NNZ, SPLIT, ST>), grid, block, lds, stream, p,
                           b_begin, nb, Z, c, vals, flags);
    };
    using yes = std::true_type;
    using no = std::false_type;
    if (c && vals) stream_out ? go(yes{}, yes{}, yes{}) : go(yes{}, yes{}, no{});
    else if (c) stream_out ? go(yes{}, no{}, yes{}) : go(yes{}, no{}, no{});
    else stream_out ? go(no{}, yes{}, yes{}) : go(no{}, yes{}, no{});
    return hipGetLastError();
}

// constraint-only launch of one instantiation (eval_c! alone: what a line search or Ipopt's eval_constraint callback asks for)
template <int T, int KC, int W>
hipError_t launch_c_only_t(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c, hipStream_t stream) {
    dim3 grid(xcd_grid(nb)), block(kWave);
    const bool stream_out = (int64_t)nb * (18 * p.N + 16) * 8 > ((int64_t)512 << 20);
    if (stream_out) hipLaunchKernelGGL((k_constraint_jacobian<T, KC, W, true, false, false, false, true>), grid, block, 0, stream, p, b_begin, nb, Z, c, nullptr, prefetch_flags(0));
    else hipLaunchKernelGGL((k_constraint_jacobian<T, KC, W, true, false, false, false, false>), grid, block, 0, stream, p, b_begin, nb, Z, c, nullptr, prefetch_flags(0));
    return hipGetLastError();
}

}  // namespace

#ifdef QLN_DIAG
extern "C" int qln_diag_set_stamps(void* dev_ptr) {
    unsigned long long* p = static_cast<unsigned long long*>(dev_ptr);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(qln::g_stamps), &p, sizeof(p));
}
#endif

hipError_t launch_constraint_jacobian(const BatchParams& p, int32_t b_begin, int32_t nb, const double* Z, double* c,
                                      double* vals, uint32_t flags, hipStream_t stream) {
    if (nb <= 0 || (!c && !vals)) return hipSuccess;
    // Latency-bound callers (the host-pointer MOI mode: one problem or a handful) ask for one workgroup per 16-knot
    // chunk instead of per problem: the launch is then as long as one chunk (profiles/r01_small_batch_split.txt: it
    // pays below ~256 problems only, so the batched entry points never ask for it).
    if (flags & kLaunchSplit) {
        flags &= ~kLaunchSplit;
        if (nb <= 256 && p.N > 17) {
            if (vals && p.jac_format == QLN_JAC_FORMAT_STRUCTURAL) return launch_cj_t<0, 16, 2, true, true>(p, b_begin, nb, Z, c, vals, flags, stream);
            return launch_cj_t<16, 16, 1, false, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        }
    }
    // Shipping configuration: T=12 (28.8 KB tile; 256 VGPRs + 17 AGPRs => one wave per SIMD, 4 waves per CU) when the Jacobian is written (below), 40- or 64-knot chunks
    // with two waves per SIMD for the constraint-only launch (profiles/r01_variants.txt, r03_dense_floor.txt, r03_c_only_variants.txt).
#ifdef QLN_TUNING
    // tuning build only (make tuning -> libqln_hip_tuning.so): QLN_VARIANT selects other instantiations for A/B runs
    static const int variant = [] {
        const char* e = getenv("QLN_VARIANT");
        return e ? atoi(e) : 0;
    }();
    switch (variant) {
        case 1: return launch_cj_t<8, 64, 2>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 2: return launch_cj_t<12, 64, 1>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 3: return launch_cj_t<16, 64, 1>(p, b_begin, nb, Z, c, vals, flags, stream);
        // the shipping tile with the prefetch, register budget of one / two waves per SIMD (4 / 5 waves per CU; two: 104 B of scratch)
        case 4: return launch_cj_t<12, 64, 1>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 64) ? kDensePrefetchAhead : 0u);
        case 5: return launch_cj_t<12, 64, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 64) ? kDensePrefetchAhead : 0u);
        case 6: return launch_cj_t<10, 64, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 64) ? kDensePrefetchAhead : 0u);
        // 40-knot chunks: 13 staging registers instead of 21 -- the fused instantiation then fits the two-waves-per-SIMD budget
        case 7: return launch_cj_t<12, 40, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 40) ? kDensePrefetchAhead : 0u);
        case 8: return launch_cj_t<8, 40, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 40) ? kDensePrefetchAhead : 0u);
        case 9: return launch_cj_t<16, 40, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 40) ? kDensePrefetchAhead : 0u);
        case 10: return launch_cj_t<10, 40, 2>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 40) ? kDensePrefetchAhead : 0u);
        default: break;
    }
    if (!vals) switch (variant) {  // constraint-only launch: chunk size / tile (= LDS) / register budget
        case 31: return launch_c_only_t<5, 40, 2>(p, b_begin, nb, Z, c, stream);
        case 32: return launch_c_only_t<5, 40, 3>(p, b_begin, nb, Z, c, stream);
        case 33: return launch_c_only_t<5, 40, 4>(p, b_begin, nb, Z, c, stream);
        case 34: return launch_c_only_t<8, 64, 3>(p, b_begin, nb, Z, c, stream);
        case 35: return launch_c_only_t<8, 64, 4>(p, b_begin, nb, Z, c, stream);
        case 36: return launch_c_only_t<4, 32, 3>(p, b_begin, nb, Z, c, stream);
        case 37: return launch_c_only_t<4, 32, 4>(p, b_begin, nb, Z, c, stream);
        default: break;
    }
    switch (variant) {  // small-batch launches: one workgroup per chunk
        case 21: if (p.jac_format == QLN_JAC_FORMAT_DENSE_BLOCKS) return launch_cj_t<16, 16, 1, false, true>(p, b_begin, nb, Z, c, vals, flags, stream); break;
        case 22: if (p.jac_format == QLN_JAC_FORMAT_DENSE_BLOCKS) return launch_cj_t<8, 8, 2, false, true>(p, b_begin, nb, Z, c, vals, flags, stream); break;
        case 23: if (p.jac_format == QLN_JAC_FORMAT_DENSE_BLOCKS) return launch_cj_t<10, 10, 2, false, true>(p, b_begin, nb, Z, c, vals, flags, stream); break;
        case 24: if (p.jac_format == QLN_JAC_FORMAT_STRUCTURAL && vals) return launch_cj_t<0, 16, 2, true, true>(p, b_begin, nb, Z, c, vals, flags, stream); break;
        case 25: if (p.jac_format == QLN_JAC_FORMAT_STRUCTURAL && vals) return launch_cj_t<0, 8, 2, true, true>(p, b_begin, nb, Z, c, vals, flags, stream); break;
        default: break;
    }
    if (vals && p.jac_format == QLN_JAC_FORMAT_STRUCTURAL) switch (variant) {
        case 11: return launch_cj_t<0, 32, 2, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 12: return launch_cj_t<0, 40, 2, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 13: return launch_cj_t<0, 32, 1, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 14: return launch_cj_t<0, 64, 1, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 15: return launch_cj_t<0, 64, 2, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 16: return launch_cj_t<20, 40, 3, true>(p, b_begin, nb, Z, c, vals, flags, stream);  // two sub-tiles, three waves per SIMD
        case 17: return launch_cj_t<20, 40, 2, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 18: return launch_cj_t<0, 40, 2, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        case 19: return launch_cj_t<14, 40, 3, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        default: break;
    }
#endif
    if (!vals) {
        // constraint-only launch (eval_c! alone: a line search's or Ipopt's eval_constraint call): no tile to fill, so the LDS holds
        // only the staged slice and the residual stage.  40-knot chunks (12 KB of LDS, 13 staging registers per lane) where
        // they make no more passes than 64-knot chunks would (N <= 41, 66 <= N <= 81, ...): 0.157 -> 0.149 ms at config 3
        // (profiles/r03_c_only_variants.txt; three or four waves per SIMD need <= 168 / 128 VGPRs and spill: 0.18 / 0.30 ms)
        const int knots = p.N - 1;
        if ((knots + 39) / 40 == (knots + 63) / 64) return launch_c_only_t<5, 40, 2>(p, b_begin, nb, Z, c, stream);
        return launch_c_only_t<8, 64, 2>(p, b_begin, nb, Z, c, stream);
    }
    // structural format: 40-knot chunks (dynamic LDS sized for the batch, 19.1 KB at config 3; 2 waves per SIMD; profiles/r01_structural_variants.txt)
    // Horizons that 64-knot chunks cover in fewer passes (N - 1 = 41 .. 64, 81 .. 128, ...) take those: N = 65 0.378 against 0.410 ms
    // for the same knot points as config 3 (bench/packing_probe.py, profiles/r03_packing_probe.txt)
    if (p.jac_format == QLN_JAC_FORMAT_STRUCTURAL) {
        const int knots = p.N - 1;
        if ((knots + 39) / 40 == (knots + 63) / 64) return launch_cj_t<0, 40, 1, true>(p, b_begin, nb, Z, c, vals, flags, stream);
        return launch_cj_t<0, 64, 1, true>(p, b_begin, nb, Z, c, vals, flags, stream);
    }
    // dense blocks: a 12-block tile (28.8 KB of LDS would admit 5 waves per CU; the fused instantiation with 64-knot chunks takes
    // 256 VGPRs + 17 AGPRs, i.e. one wave per SIMD = 4 waves per CU.  40-knot chunks fit the two-waves-per-SIMD budget without scratch
    // -- 5 / 6 / 8 waves per CU with T = 12 / 10 / 8 -- and are no faster: 1.044-1.047 / 1.076-1.078 / 1.070-1.072 ms against
    // 1.047 ms, same box: more concurrent write fronts cost, profiles/r03_dense_floor.txt).  All tile sizes sit on the launch's floor;
    // on region-placed buffers T = 12 is the fastest by 0.5-1 % (config 3: 1.060-1.063 against 1.069-1.074 ms for T = 16, config 4:
    // 2.155-2.162 against 2.168-2.174 ms; round 1 chose T = 16 on buffers lying in one region)
    // One-chunk problems (N <= 65) prefetch the slice of the workgroup 64 problems ahead on the XCD into L2 (kDensePrefetchAhead):
    // config 3 1.056-1.068 -> 1.031-1.040 ms for every distance from 16 to 1024, same box (profiles/r03_prefetch_ahead.txt);
    // two-chunk problems (config 4), the structural format and the constraint-only launch do not gain and stay without.
    return launch_cj_t<12, 64, 1>(p, b_begin, nb, Z, c, vals, flags, stream, (c && p.N - 1 <= 64) ? kDensePrefetchAhead : 0u);
}

// f, grad, c and the Jacobian values of the whole batch from ONE read of Z (qln_eval_all)
hipError_t launch_eval_all(const BatchParams& p, const double* Z, double* f, double* grad, double* c, double* vals, uint32_t flags,
                           hipStream_t stream) {
    const int nb = p.B;
    dim3 grid(xcd_grid(nb)), block(kWave);
    const bool structural = p.jac_format == QLN_JAC_FORMAT_STRUCTURAL;
    const bool stream_out = (int64_t)nb * (p.N - 1) * (structural ? 71 : kBlk) * 8 > ((int64_t)512 << 20);
    // dense one-chunk problems prefetch like the fused launch, and the later problem's boundary vectors and descriptor with the
    // slice: 1.18-1.20 -> 1.11-1.15 ms at config 3 (the slice alone: 1.16-1.19; for the fused launch WITHOUT the objective the two
    // extras cost what the slice gains -- profiles/r03_prefetch_ahead.txt)
    flags = (flags & QLN_JAC_WRITE_CONSTANTS) |
            ((!structural && p.N - 1 <= 64) ? prefetch_flags(kDensePrefetchAhead, kPrefetchBnd | kPrefetchDesc) : prefetch_flags(0));
    if (structural) {
        const unsigned lds = (unsigned)nnz_lds_bytes(p, 40, 40, true);
        if (stream_out) hipLaunchKernelGGL((k_constraint_jacobian<0, 40, 1, true, true, true, false, true, true>), grid, block, lds, stream, p, 0, nb, Z, c, vals, flags, f, grad);
        else hipLaunchKernelGGL((k_constraint_jacobian<0, 40, 1, true, true, true, false, false, true>), grid, block, lds, stream, p, 0, nb, Z, c, vals, flags, f, grad);
    } else {
        if (stream_out) hipLaunchKernelGGL((k_constraint_jacobian<16, 64, 1, true, true, false, false, true, true>), grid, block, 0, stream, p, 0, nb, Z, c, vals, flags, f, grad);
        else hipLaunchKernelGGL((k_constraint_jacobian<16, 64, 1, true, true, false, false, false, true>), grid, block, 0, stream, p, 0, nb, Z, c, vals, flags, f, grad);
    }
    return hipGetLastError();
}

// eval_f + eval_c! of the whole batch from ONE read of Z (qln_eval_objective_and_constraint): what a line search, or Ipopt's
// filter at a trial point, asks for -- objective and constraints, no derivatives
hipError_t launch_objective_and_constraint(const BatchParams& p, const double* Z, double* f, double* c, hipStream_t stream) {
    const int nb = p.B;
    dim3 grid(xcd_grid(nb)), block(kWave);
    const bool stream_out = (int64_t)nb * (18 * p.N + 16) * 8 > ((int64_t)512 << 20);
    const int knots = p.N - 1;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, block, 0, stream, p, 0, nb, Z, c, nullptr, 0u, f, nullptr); };
    if ((knots + 39) / 40 == (knots + 63) / 64) {  // chunk size as for the constraint-only launch
        if (stream_out) go(k_constraint_jacobian<5, 40, 2, true, false, false, false, true, true>);
        else go(k_constraint_jacobian<5, 40, 2, true, false, false, false, false, true>);
    } else {
        if (stream_out) go(k_constraint_jacobian<8, 64, 2, true, false, false, false, true, true>);
        else go(k_constraint_jacobian<8, 64, 2, true, false, false, false, false, true>);
    }
    return hipGetLastError();
}

hipError_t launch_kinematic_rows(const BatchParams& p, const double* Z, double* d, double* jac_vals, hipStream_t stream) {
    const int64_t n = (int64_t)p.B * p.N;
    hipLaunchKernelGGL(k_kinematic_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, Z, d, jac_vals);
    return hipGetLastError();
}

hipError_t launch_friction_rows(const BatchParams& p, const double* Z, double mu, double* d, double* jac_vals, hipStream_t stream) {
    const int64_t n = (int64_t)p.B * (p.N - 1);
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_friction_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, Z, mu, d, jac_vals);
    return hipGetLastError();
}

hipError_t launch_jacobian_constants(const BatchParams& p, double* vals, hipStream_t stream) {
    hipLaunchKernelGGL(k_jacobian_constants, dim3(xcd_grid(p.B)), dim3(kWave), 0, stream, p, vals);
    return hipGetLastError();
}

hipError_t launch_objective(const BatchParams& p, const double* Z, double* f, hipStream_t stream) {
    if (p.cost_batch == 1 && p.N <= kWave && p.B >= 4096) {
        // one shared cost table: persistent waves with lane = knot's cost record in registers, two per SIMD
        const int ki_ = ((20 * p.N - 6) / 2 + kWave - 1) / kWave;
        const size_t lds = (size_t)((ki_ * 2 * kWave * 21) / 20 + 2 + kWave) * sizeof(double);
        const int ki = ((20 * p.N - 6) / 2 + kWave - 1) / kWave;  // 16-byte load instructions per slice: 1 .. 10
        int per_cu = 8;
#ifdef QLN_TUNING
        static const int var = [] { const char* e = getenv("QLN_OBJ_VARIANT"); return e ? atoi(e) : 0; }();
        static const int pc = [] { const char* e = getenv("QLN_OBJ_PER_CU"); return e ? atoi(e) : 0; }();
        if (pc) per_cu = pc;
        const int g_ = std::min(256 * per_cu, p.B);
#define OBJ_LAUNCH(...) hipLaunchKernelGGL((k_objective_shared<__VA_ARGS__>), dim3(g_), dim3(kWave), lds, stream, p, Z, f); return hipGetLastError()
        if (ki == 7) {
            switch (var) {
                case 1: OBJ_LAUNCH(7, false, 2, 2);
                case 2: OBJ_LAUNCH(7, true, 1, 2);
                case 3: OBJ_LAUNCH(7, true, 1, 3);
                case 4: OBJ_LAUNCH(7, false, 1, 3);
                case 5: OBJ_LAUNCH(7, true, 3, 2);
                default: break;
            }
        }
#endif
        const int grid = std::min(256 * per_cu, p.B);
#define OBJ_CASE(K) case K: hipLaunchKernelGGL((k_objective_shared<K, true, 2, 2>), dim3(grid), dim3(kWave), lds, stream, p, Z, f); break
        switch (ki) {
            OBJ_CASE(1); OBJ_CASE(2); OBJ_CASE(3); OBJ_CASE(4); OBJ_CASE(5);
            OBJ_CASE(6); OBJ_CASE(7); OBJ_CASE(8); OBJ_CASE(9); OBJ_CASE(10);
            default: return hipErrorInvalidValue;
        }
#undef OBJ_CASE
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_objective, dim3(xcd_grid(p.B)), dim3(kWave), 0, stream, p, Z, f);
    return hipGetLastError();
}

hipError_t launch_constraint_violation(const BatchParams& p, const double* c, double* viol, hipStream_t stream) {
    hipLaunchKernelGGL(k_constraint_violation, dim3(xcd_grid(p.B)), dim3(kWave), 0, stream, p, c, viol);
    return hipGetLastError();
}

hipError_t launch_lqr_cost(const BatchParams& p, const double* qrqf, double dt, double* cost, int cost_batch,
                           hipStream_t stream) {
    const int n = cost_batch * p.N;
    hipLaunchKernelGGL(k_lqr_cost, dim3((n + 255) / 256), dim3(256), 0, stream, p, qrqf, dt, cost, cost_batch);
    return hipGetLastError();
}

hipError_t launch_initial_guess(const BatchParams& p, double* Z, hipStream_t stream) {
    const int n_nlp = 20 * p.N - 5;
    dim3 grid(xcd_grid(p.B), (n_nlp + 255) / 256);
    hipLaunchKernelGGL(k_initial_guess, grid, dim3(256), 0, stream, p, Z);
    return hipGetLastError();
}

hipError_t launch_objective_gradient(const BatchParams& p, const double* Z, double* grad, hipStream_t stream) {
    if (p.cost_batch == 1 && p.N <= kWave && p.B >= 4096) {
        const int ki = ((20 * p.N - 6) / 2 + kWave - 1) / kWave;  // 16-byte pieces per lane and slice: 1 .. 10
        int per_cu = 8;
#ifdef QLN_TUNING
        static const int var = [] { const char* e = getenv("QLN_GRAD_VARIANT"); return e ? atoi(e) : 0; }();
        static const int pc = [] { const char* e = getenv("QLN_GRAD_PER_CU"); return e ? atoi(e) : 0; }();
        if (pc) per_cu = pc;
        const int g_ = std::min(256 * per_cu, p.B);
#define GRAD_LAUNCH(...) hipLaunchKernelGGL((k_objective_gradient_shared<__VA_ARGS__>), dim3(g_), dim3(kWave), 0, stream, p, Z, grad); return hipGetLastError()
        if (ki == 7) {
            switch (var) {
                case 1: GRAD_LAUNCH(7, 2, 2);
                case 2: GRAD_LAUNCH(7, 2, 3);
                case 3: GRAD_LAUNCH(7, 1, 3);
                case 4: GRAD_LAUNCH(7, 3, 2);
                case 5: GRAD_LAUNCH(7, 1, 4);
                default: break;
            }
        }
#endif
        const int grid = std::min(256 * per_cu, p.B);
#define GRAD_CASE(K) case K: hipLaunchKernelGGL((k_objective_gradient_shared<K, 1, 2>), dim3(grid), dim3(kWave), 0, stream, p, Z, grad); break
        switch (ki) {
            GRAD_CASE(1); GRAD_CASE(2); GRAD_CASE(3); GRAD_CASE(4); GRAD_CASE(5);
            GRAD_CASE(6); GRAD_CASE(7); GRAD_CASE(8); GRAD_CASE(9); GRAD_CASE(10);
            default: return hipErrorInvalidValue;
        }
#undef GRAD_CASE
        return hipGetLastError();
    }
    const int64_t total = (int64_t)p.B * (20 * p.N - 5);
    const int ntiles = (int)((total + kGradU * 256 - 1) / (kGradU * 256));
    hipLaunchKernelGGL(k_objective_gradient, dim3(xcd_grid(ntiles)), dim3(256), 0, stream, p, Z, grad, total, ntiles);
    return hipGetLastError();
}

}  // namespace qln
