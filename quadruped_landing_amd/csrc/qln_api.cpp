// qln_api.cpp -- the C ABI of include/qln_evaluator.h on top of the gfx950 kernels.
//
// Owns: the handle (device copies of the problem descriptors, offset tables, lazily allocated
// staging for the host-pointer MOI mode).  Never owns or reallocates caller buffers, never throws
// across the boundary, never calls exit (SURVEY.md 8b "Errors"/"Ownership").
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/qln_evaluator.h"
#include "qln_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define QLN_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(QLN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Whether a placed buffer's virtual range is handed back (hipMemAddressFree) when the buffer is released.
// It is NOT: on this stack (ROCm 7.2 user space, the pool's host driver) a virtual address that has been unmapped
// with hipMemUnmap keeps serving GPU accesses through the translations of its previous mapping, with every call
// returning hipSuccess -- so an address that was ever unmapped must never be mapped again, neither in place nor
// after hipMemAddressFree + hipMemAddressReserve (which hands the same addresses out again).  Reproduced without
// this library or torch by bench/vmm_va_reuse.cpp (profiles/r02_vmm_va_reuse.txt):
//   reserve R; map chunks A; fill through R; hipMemUnmap(R); map chunks B at R; fill through R
//     -> every word of A (still alive, inspected through a second mapping) carries B's pattern;
//   ... hipMemUnmap(R); hipMemAddressFree(R); hipMemAddressReserve -> R again; map B; fill; read back
//     -> ~0.7 % of the words read back wrong, and A (alive) has been written into.
// That is what round 1 saw as "writes through stale translations"; it is not a missing synchronisation (the device was
// idle at every unmap in that sequence, and is here).  Keeping a released buffer's range reserved costs address space
// only (~70 GiB of 128 TiB per qln_vals_alloc_placed call); the physical memory is returned.
constexpr bool kReturnVirtualRange = false;
// ... so every placed allocation retires its virtual range for the life of the process.  The ranges are counted, reported
// (qln_vals_placed_address_space) and capped well inside the 128 TiB of user address space: a process that places buffers in
// a loop gets a clear refusal after ~900 default-size calls instead of a failing hipMemAddressReserve somewhere else.
constexpr uint64_t kVaCapBytes = (uint64_t)64 << 40;  // 64 TiB: half the address space stays for everything else
std::atomic<uint64_t> g_va_retired{0};

// sizes of src/nlp.jl:48-87
int32_t m_nlp_of(int32_t N, int32_t kt) { return 18 * N - kt + 16; }
int32_t nnz_dyn_of(int32_t N, int32_t kt, int32_t fmt) {
    return (fmt == QLN_JAC_FORMAT_STRUCTURAL ? qln::step_block_offset(N - 1, N, kt) : 300 * (N - 1)) + N;
}
int32_t nnz_of(int32_t N, int32_t kt, int32_t fmt) { return nnz_dyn_of(N, kt, fmt) + 435 + 15 * (N - 1) + 3 * N - kt + 3; }

void cinds_of(int32_t N, int32_t kt, int32_t out[14]) {
    int32_t e = 0;
    const int32_t len[7] = {15, 14, 15 * (N - 1), N, N - kt + 1, 1, N};
    for (int i = 0; i < 7; ++i) {
        out[2 * i] = e + 1;
        e += len[i];
        out[2 * i + 1] = e;
    }
}

}  // namespace

struct qln_handle {
    int device = 0;
    qln_model model{};
    hipStream_t stream = nullptr;
    qln::BatchParams p{};
    qln_dims dims{};
    std::vector<int32_t> k_trans, init_mode;
    std::vector<int64_t> c_off, j_off;
    // device storage owned by the handle
    qln::ProblemDesc* d_desc = nullptr;
    double* d_bnd = nullptr;
    double* d_cost = nullptr;
    // staging for host-pointer mode
    double* s_Z = nullptr;
    double* s_c = nullptr;
    double* s_vals = nullptr;
    double* s_f = nullptr;
    double* s_grad = nullptr;
    std::vector<double> h_vals_one;
    // zero-copy MOI mode (small batches): pinned host buffers mapped into the device's address space -- the kernels read
    // Z from and write their results to host memory directly, so a callback is one launch and one synchronisation
    struct Mapped {
        double* host = nullptr;
        double* dev = nullptr;
    };
    Mapped m_Z, m_c, m_vals, m_f, m_grad;
    bool zero_copy = false;
    // dense MOI scatter: per problem, where each value of the vals segment goes in the column-major matrix, and the
    // write-set's explicit zeros (built on first use)
    struct DenseMap {
        std::vector<int64_t> at, zeros;
    };
    std::vector<DenseMap> dense_map;
    // buffers handed out by qln_vals_alloc_placed
    struct Placed {
        char* va = nullptr;        // reserved virtual range
        size_t va_size = 0;        // bytes reserved at va
        size_t chunk = 0;
        size_t first = 0;          // first chunk still mapped
        size_t scanned = 0;        // chunks the placement scan looked at
        std::vector<hipMemGenericAllocationHandle_t> chunks;  // the mapped ones, in order
        double* vals = nullptr;
    };
    std::vector<Placed> placed;
    // scratch of qln_solve (step blocks of every knot), allocated on first use
    double* solve_scratch = nullptr;
    size_t solve_scratch_n = 0;
};

namespace {

template <class T>
int upload(T** dst, const T* src, size_t n) {
    QLN_HIP(hipMalloc(reinterpret_cast<void**>(dst), std::max<size_t>(n, 1) * sizeof(T)));
    if (n) QLN_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return QLN_OK;
}

int ensure(double** buf, int64_t n) {
    if (*buf) return QLN_OK;
    QLN_HIP(hipMalloc(reinterpret_cast<void**>(buf), std::max<int64_t>(n, 1) * sizeof(double)));
    // the padding between problems is never written by the kernels: the caller gets zeros there, not stale memory
    QLN_HIP(hipMemset(*buf, 0, std::max<int64_t>(n, 1) * sizeof(double)));
    return QLN_OK;
}

int ensure_mapped(qln_handle::Mapped* m, int64_t n) {
    if (m->host) return QLN_OK;
    void* hp = nullptr;
    QLN_HIP(hipHostMalloc(&hp, std::max<int64_t>(n, 1) * sizeof(double), hipHostMallocMapped));
    std::memset(hp, 0, std::max<int64_t>(n, 1) * sizeof(double));  // padding the kernels never write stays zero
    void* dp = nullptr;
    if (hipError_t e = hipHostGetDevicePointer(&dp, hp, 0); e != hipSuccess) {
        (void)hipHostFree(hp);
        return fail(QLN_ERR_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    m->host = static_cast<double*>(hp);
    m->dev = static_cast<double*>(dp);
    return QLN_OK;
}

int check_handle(const qln_handle* h) {
    if (!h) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    return QLN_OK;
}

int bind_device(const qln_handle* h) {
    QLN_HIP(hipSetDevice(h->device));
    return QLN_OK;
}

}  // namespace

extern "C" {

const char* qln_last_error(void) { return g_err.c_str(); }
int qln_set_last_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
const char* qln_version(void) { return "quadruped_landing_amd 0.1 (gfx950)"; }

// Argument checks and the layout of a batch (sizes, per-problem offsets): everything qln_create decides before it touches
// a device.  c_off / j_off: [B] or null.
static int layout_of(const qln_batch_desc* d, const char* who, bool need_states, qln_dims* D, int64_t* c_off, int64_t* j_off) {
    const std::string w = std::string(who) + ": ";
    if (d->B < 1) return fail(QLN_ERR_INVALID_ARGUMENT, w + "B must be >= 1");
    if (d->N < 2) return fail(QLN_ERR_INVALID_ARGUMENT, w + "N must be >= 2");
    if (d->N > 50000000 / 20) return fail(QLN_ERR_INVALID_ARGUMENT, w + "N too large for 32-bit indices");
    if (!d->k_trans || !d->init_mode || (need_states && (!d->x0 || !d->xf)))
        return fail(QLN_ERR_INVALID_ARGUMENT, w + "null descriptor array");
    if (d->cost_batch != 1 && d->cost_batch != d->B)
        return fail(QLN_ERR_INVALID_ARGUMENT, w + "cost_batch must be 1 or B");
    const int32_t n_nlp = 20 * d->N - 5;
    const int64_t z_stride = d->z_stride ? d->z_stride : n_nlp;
    if (z_stride < n_nlp) return fail(QLN_ERR_INVALID_ARGUMENT, w + "z_stride < n_nlp");
    const int64_t align = d->align ? d->align : 16;
    if (align < 1) return fail(QLN_ERR_INVALID_ARGUMENT, w + "align must be >= 1");
    const int32_t fmt = d->jac_format;
    if (fmt != QLN_JAC_FORMAT_DENSE_BLOCKS && fmt != QLN_JAC_FORMAT_STRUCTURAL)
        return fail(QLN_ERR_INVALID_ARGUMENT, w + "unknown jac_format");
    for (int32_t b = 0; b < d->B; ++b) {
        if (d->k_trans[b] < 1 || d->k_trans[b] > d->N + 1)
            return fail(QLN_ERR_INVALID_ARGUMENT, w + "k_trans out of range [1, N+1] at problem " + std::to_string(b));
        if (d->init_mode[b] != 1 && d->init_mode[b] != 2)
            return fail(QLN_ERR_INVALID_ARGUMENT, w + "init_mode must be 1 or 2 at problem " + std::to_string(b));
    }
    const int64_t jalign = (align % 2) ? align * 2 : align;  // keep j_off even: 16-byte stores
    int64_t co = 0, jo = 0;
    int32_t m_max = 0, nnz_max = 0, dyn_max = 0;
    for (int32_t b = 0; b < d->B; ++b) {
        const int32_t m = m_nlp_of(d->N, d->k_trans[b]), nz = nnz_of(d->N, d->k_trans[b], fmt);
        dyn_max = std::max(dyn_max, nnz_dyn_of(d->N, d->k_trans[b], fmt));
        co = round_up(co, align);
        jo = round_up(jo, jalign);
        if (c_off) c_off[b] = co;
        if (j_off) j_off[b] = jo;
        co += m;   // no padding behind the last problem: for B == 1 the totals are exactly m_nlp / nnz,
        jo += nz;  // so Ipopt-owned buffers of the reference's sizes can be passed as they are
        m_max = std::max(m_max, m);
        nnz_max = std::max(nnz_max, nz);
    }
    D->B = d->B;
    D->N = d->N;
    D->n_nlp = n_nlp;
    D->m_nlp_max = m_max;
    D->nnz_max = nnz_max;
    D->nnz_dynamic = dyn_max;
    D->z_stride = z_stride;
    D->z_total = z_stride * (int64_t)d->B;
    D->c_total = co;
    D->j_total = jo;
    return QLN_OK;
}

int qln_layout(const qln_batch_desc* d, qln_dims* dims, int64_t* c_off, int64_t* j_off) {
    if (!d || !dims) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_layout: null argument");
    return layout_of(d, "qln_layout", false, dims, c_off, j_off);
}

int qln_create(const qln_batch_desc* d, int device, qln_handle** out) {
    if (!d || !out) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_create: null argument");
    *out = nullptr;
    qln_dims lay{};
    std::vector<int64_t> c_off((size_t)std::max(d->B, 0)), j_off((size_t)std::max(d->B, 0));
    if (int rc = layout_of(d, "qln_create", true, &lay, c_off.data(), j_off.data())) return rc;
    const int64_t z_stride = lay.z_stride;
    const int32_t fmt = d->jac_format;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(QLN_ERR_NO_DEVICE, "qln_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_create: bad device ordinal");

    qln_handle* h = new (std::nothrow) qln_handle();
    if (!h) return fail(QLN_ERR_HIP, "qln_create: out of host memory");
    h->device = device;
    h->model = d->model;
    h->k_trans.assign(d->k_trans, d->k_trans + d->B);
    h->init_mode.assign(d->init_mode, d->init_mode + d->B);
    h->c_off = std::move(c_off);
    h->j_off = std::move(j_off);
    h->dims = lay;
    qln_dims& D = h->dims;

    int rc = QLN_OK;
    auto bail = [&](int code) {
        qln_destroy(h);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(QLN_ERR_HIP, "hipSetDevice failed"));
    {
        // one 32-byte descriptor record and one 30-double boundary record per problem
        std::vector<qln::ProblemDesc> desc((size_t)d->B);
        std::vector<double> bnd((size_t)d->B * 30);
        for (int32_t b = 0; b < d->B; ++b) {
            desc[b].k_trans = d->k_trans[b];
            desc[b].init_mode = d->init_mode[b];
            desc[b].c_off = h->c_off[b];
            desc[b].j_off = h->j_off[b];
            desc[b].reserved = 0;
            std::memcpy(&bnd[(size_t)b * 30], d->x0 + (size_t)b * 15, 15 * sizeof(double));
            std::memcpy(&bnd[(size_t)b * 30 + 15], d->xf + (size_t)b * 15, 15 * sizeof(double));
        }
        if ((rc = upload(&h->d_desc, desc.data(), desc.size()))) return bail(rc);
        if ((rc = upload(&h->d_bnd, bnd.data(), bnd.size()))) return bail(rc);
    }
    if (d->cost && (rc = upload(&h->d_cost, d->cost, (size_t)d->cost_batch * d->N * QLN_COST_STRIDE))) return bail(rc);

    qln::BatchParams& P = h->p;
    P.B = d->B;
    P.N = d->N;
    P.g = d->model.g;
    P.mb = d->model.mb;
    P.mf = d->model.mf;
    P.lb = d->model.lb;
    P.desc = h->d_desc;
    P.bnd = h->d_bnd;
    P.cost = h->d_cost;
    P.cost_batch = d->cost_batch;
    P.z_stride = z_stride;
    P.jac_format = fmt;
    P.kt_max = *std::max_element(d->k_trans, d->k_trans + d->B);
    // host-pointer (MOI) mode: up to 8 MB per callback the kernels work on mapped host memory directly (one launch, no
    // copies: 19-21 us against 32 us for the notebook's problem, profiles/r01_moi_latency.txt); larger batches are
    // staged through device memory
    h->zero_copy = (D.z_total + D.c_total + D.j_total) * (int64_t)sizeof(double) <= ((int64_t)8 << 20);
    *out = h;
    return QLN_OK;
}

// Gives a placed buffer back: unmap every chunk still mapped, release the physical handles, free the virtual range.
// The caller has synchronised the DEVICE (not just the handle's stream: the buffer was handed out as plain memory, so
// work on any stream -- torch's current one, a stream set later with qln_set_stream -- may have touched it).  Every
// return code is looked at; the first failure is reported through qln_last_error and the rest is still attempted.
static int release_placed(qln_handle::Placed& p) {
    hipError_t first = hipSuccess;
    const char* what = nullptr;
    auto note = [&](hipError_t e, const char* w) {
        if (e != hipSuccess && first == hipSuccess) {
            first = e;
            what = w;
        }
    };
    for (size_t i = 0; i < p.chunks.size(); ++i) {
        note(hipMemUnmap(p.va + (p.first + i) * p.chunk, p.chunk), "hipMemUnmap");
        note(hipMemRelease(p.chunks[i]), "hipMemRelease");
    }
    p.chunks.clear();
    if (p.va && kReturnVirtualRange) note(hipMemAddressFree(p.va, p.va_size), "hipMemAddressFree");
    p.va = nullptr;
    if (first != hipSuccess) return fail(QLN_ERR_HIP, std::string("releasing a placed buffer: ") + what + ": " + hipGetErrorString(first));
    return QLN_OK;
}

int qln_destroy(qln_handle* h) {
    if (!h) return QLN_OK;
    (void)hipSetDevice(h->device);
    int rc = QLN_OK;
    if (!h->placed.empty()) {
        // nothing on any stream may still be reading or writing memory about to be unmapped
        if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess)
            rc = fail(QLN_ERR_HIP, std::string("qln_destroy: hipDeviceSynchronize: ") + hipGetErrorString(e));
        for (auto& p : h->placed)
            if (int r = release_placed(p); r != QLN_OK && rc == QLN_OK) rc = r;
    }
    for (qln_handle::Mapped* m : {&h->m_Z, &h->m_c, &h->m_vals, &h->m_f, &h->m_grad})
        if (m->host) (void)hipHostFree(m->host);
    void* bufs[] = {h->d_desc, h->d_bnd, h->d_cost, h->s_Z, h->s_c, h->s_vals, h->s_f, h->s_grad, h->solve_scratch};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete h;
    return rc;
}

int qln_set_stream(qln_handle* h, void* hip_stream) {
    if (int rc = check_handle(h)) return rc;
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return QLN_OK;
}

int qln_synchronize(qln_handle* h) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(hipStreamSynchronize(h->stream));
    return QLN_OK;
}

int qln_get_dims(const qln_handle* h, qln_dims* out) {
    if (int rc = check_handle(h)) return rc;
    if (!out) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_get_dims: null out");
    *out = h->dims;
    return QLN_OK;
}

int qln_get_offsets(const qln_handle* h, int64_t* c_off, int64_t* j_off) {
    if (int rc = check_handle(h)) return rc;
    if (c_off) std::copy(h->c_off.begin(), h->c_off.end(), c_off);
    if (j_off) std::copy(h->j_off.begin(), h->j_off.end(), j_off);
    return QLN_OK;
}

static int check_problem(const qln_handle* h, int32_t b) {
    if (int rc = check_handle(h)) return rc;
    if (b < 0 || b >= h->dims.B) return fail(QLN_ERR_INVALID_ARGUMENT, "problem index out of range");
    return QLN_OK;
}

int qln_problem_dims(const qln_handle* h, int32_t b, int32_t* m_nlp, int32_t* nnz) {
    if (int rc = check_problem(h, b)) return rc;
    if (m_nlp) *m_nlp = m_nlp_of(h->dims.N, h->k_trans[b]);
    if (nnz) *nnz = nnz_of(h->dims.N, h->k_trans[b], h->p.jac_format);
    return QLN_OK;
}

int qln_problem_nnz_dynamic(const qln_handle* h, int32_t b, int32_t* nnz_dynamic) {
    if (int rc = check_problem(h, b)) return rc;
    if (!nnz_dynamic) return fail(QLN_ERR_INVALID_ARGUMENT, "null nnz_dynamic");
    *nnz_dynamic = nnz_dyn_of(h->dims.N, h->k_trans[b], h->p.jac_format);
    return QLN_OK;
}

int qln_constraint_index_ranges(const qln_handle* h, int32_t b, int32_t cinds[14]) {
    if (int rc = check_problem(h, b)) return rc;
    if (!cinds) return fail(QLN_ERR_INVALID_ARGUMENT, "null cinds");
    cinds_of(h->dims.N, h->k_trans[b], cinds);
    return QLN_OK;
}

int qln_constraint_bounds(const qln_handle* h, int32_t b, double* lb, double* ub) {
    // src/nlp.jl:66-69: lb = ub = 0 except ub[c_body_pos_inds] = Inf
    if (int rc = check_problem(h, b)) return rc;
    if (!lb || !ub) return fail(QLN_ERR_INVALID_ARGUMENT, "null bounds");
    int32_t ci[14];
    cinds_of(h->dims.N, h->k_trans[b], ci);
    for (int32_t i = 0; i < ci[13]; ++i) {
        lb[i] = 0.0;
        ub[i] = 0.0;
    }
    for (int32_t i = ci[12]; i <= ci[13]; ++i) ub[i - 1] = std::numeric_limits<double>::infinity();
    return QLN_OK;
}

int qln_jacobian_structure(const qln_handle* h, int32_t b, int32_t* rows, int32_t* cols) {
    if (int rc = check_problem(h, b)) return rc;
    if (!rows || !cols) return fail(QLN_ERR_INVALID_ARGUMENT, "null rows/cols");
    const int32_t N = h->dims.N, kt = h->k_trans[b], im = h->init_mode[b];
    int32_t ci[14];
    cinds_of(N, kt, ci);
    const int32_t r_init = ci[0] - 1, r_term = ci[2] - 1, r_dyn = ci[4] - 1, r_ci = ci[6] - 1, r_co = ci[8] - 1,
                  r_fc = ci[10] - 1, r_bp = ci[12] - 1;
    const int32_t y_init = (im == 1) ? 4 : 6, y_other = (im == 1) ? 6 : 4;
    int64_t e = 0;
    auto put = [&](int32_t r, int32_t c) {
        rows[e] = r;
        cols[e] = c;
        ++e;
    };
    const bool structural = (h->p.jac_format == QLN_JAC_FORMAT_STRUCTURAL);
    for (int32_t k = 0; k < N - 1; ++k) {  // D[ci, [xi[k]; ui[k]]], src/constraints.jl:186-198
        const int cat = qln::step_category(k + 1, kt, im);
        for (int32_t c = 0; c < 20; ++c)
            for (int32_t r = 0; r < 15; ++r)
                if (!structural || qln::step_entry_present(cat, r, c)) put(r_dyn + 15 * k + r, 20 * k + c);
    }
    for (int32_t k = 0; k < N; ++k) put(r_bp + k, 20 * k + 2);       // :269-273
    for (int32_t c = 0; c < 15; ++c)                                  // :228
        for (int32_t r = 0; r < 15; ++r) put(r_init + r, c);
    for (int32_t c = 0; c < 15; ++c)                                  // :229
        for (int32_t r = 0; r < 14; ++r) put(r_term + r, 20 * (N - 1) + c);
    for (int32_t k = 0; k < N - 1; ++k)                               // :200 (diagonal of -I(n))
        for (int32_t r = 0; r < 15; ++r) put(r_dyn + 15 * k + r, 20 * (k + 1) + r);
    for (int32_t k = 0; k < N; ++k) put(r_ci + k, 20 * k + y_init);   // :235-243
    for (int32_t K = kt; K <= N; ++K) put(r_co + (K - kt), 20 * (K - 1) + y_other);  // :246-256
    put(r_fc, 20 * (N - 2) + 16);                                     // :259
    put(r_fc, 20 * (N - 2) + 18);                                     // :260
    for (int32_t k = 0; k < N; ++k) put(r_bp + k, 20 * k + 1);        // :266
    return QLN_OK;
}

// ------------------------------------------------------------------ device-pointer mode

static int check_cost(const qln_handle* h) {
    if (!h->p.cost) return fail(QLN_ERR_INVALID_ARGUMENT, "no cost table: pass desc.cost or call qln_set_lqr_cost first");
    return QLN_OK;
}

int qln_set_lqr_cost(qln_handle* h, const double* Qdiag, const double* Rdiag, const double* Qfdiag, double dt,
                     int per_problem) {
    if (int rc = check_handle(h)) return rc;
    if (!Qdiag || !Rdiag || !Qfdiag) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_set_lqr_cost: null weights");
    if (int rc = bind_device(h)) return rc;
    const int32_t cb = per_problem ? h->dims.B : 1;
    double w[35];
    std::memcpy(w, Qdiag, 15 * sizeof(double));
    std::memcpy(w + 15, Rdiag, 5 * sizeof(double));
    std::memcpy(w + 20, Qfdiag, 15 * sizeof(double));
    double* d_w = nullptr;
    double* d_cost = nullptr;
    QLN_HIP(hipMalloc(reinterpret_cast<void**>(&d_w), sizeof(w)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cost), (size_t)cb * h->dims.N * QLN_COST_STRIDE * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, w, sizeof(w), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = qln::launch_lqr_cost(h->p, d_w, dt, d_cost, cb, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_w);
    if (e != hipSuccess) {
        if (d_cost) (void)hipFree(d_cost);
        return fail(QLN_ERR_HIP, std::string("qln_set_lqr_cost: ") + hipGetErrorString(e));
    }
    if (h->d_cost) (void)hipFree(h->d_cost);
    h->d_cost = d_cost;
    h->p.cost = d_cost;
    h->p.cost_batch = cb;
    return QLN_OK;
}

int qln_get_cost(qln_handle* h, double* cost_host, int32_t* cost_batch) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (cost_batch) *cost_batch = h->p.cost_batch;
    if (!cost_host) return QLN_OK;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(hipMemcpy(cost_host, h->p.cost, (size_t)h->p.cost_batch * h->dims.N * QLN_COST_STRIDE * sizeof(double),
                      hipMemcpyDeviceToHost));
    return QLN_OK;
}

int qln_eval_objective(qln_handle* h, const double* Z, double* f) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !f) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_objective: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_objective(h->p, Z, f, h->stream));
    return QLN_OK;
}

int qln_eval_objective_gradient(qln_handle* h, const double* Z, double* grad) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !grad) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_objective_gradient: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_objective_gradient(h->p, Z, grad, h->stream));
    return QLN_OK;
}

int qln_eval_constraint(qln_handle* h, const double* Z, double* c) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, nullptr, 0, h->stream));
    return QLN_OK;
}

static int check_vals(const double* vals) {
    if (!vals) return fail(QLN_ERR_INVALID_ARGUMENT, "null vals");
    if (reinterpret_cast<uintptr_t>(vals) % 16) return fail(QLN_ERR_INVALID_ARGUMENT, "vals must be 16-byte aligned");
    return QLN_OK;
}

int qln_eval_constraint_jacobian(qln_handle* h, const double* Z, double* vals, uint32_t flags) {
    if (int rc = check_handle(h)) return rc;
    if (!Z) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_jacobian: null Z");
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, nullptr, vals, flags & QLN_JAC_WRITE_CONSTANTS, h->stream));
    return QLN_OK;
}

int qln_eval_constraint_and_jacobian(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_and_jacobian: null pointer");
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, flags & QLN_JAC_WRITE_CONSTANTS, h->stream));
    return QLN_OK;
}

int qln_eval_all(qln_handle* h, const double* Z, double* f, double* grad, double* c, double* vals, uint32_t flags) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !f || !grad || !c) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_all: null pointer");
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_eval_all(h->p, Z, f, grad, c, vals, flags & QLN_JAC_WRITE_CONSTANTS, h->stream));
    return QLN_OK;
}

int qln_eval_objective_and_constraint(qln_handle* h, const double* Z, double* f, double* c) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !f || !c) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_objective_and_constraint: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_objective_and_constraint(h->p, Z, f, c, h->stream));
    return QLN_OK;
}

int qln_jacobian_init_constants(qln_handle* h, double* vals) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_jacobian_constants(h->p, vals, h->stream));
    return QLN_OK;
}

int qln_eval_kinematic_constraint(qln_handle* h, const double* Z, double* d, double* jac) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !d) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_kinematic_constraint: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_kinematic_rows(h->p, Z, d, jac, h->stream));
    return QLN_OK;
}

int qln_kinematic_bounds(const qln_handle* h, double* lower, double* upper) {
    if (int rc = check_handle(h)) return rc;
    if (!lower || !upper) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_kinematic_bounds: null pointer");
    *lower = 0.0;
    *upper = h->model.l1 + h->model.l2 + h->model.lb / 2;  // src/nlp.jl:70 (commented out there)
    return QLN_OK;
}

int qln_eval_friction_cone(qln_handle* h, const double* Z, double mu, double* d, double* jac) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !d) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_friction_cone: null pointer");
    if (!(mu > 0.0) || !std::isfinite(mu)) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_friction_cone: mu must be positive and finite");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_friction_rows(h->p, Z, mu, d, jac, h->stream));
    return QLN_OK;
}

int qln_constraint_violation(qln_handle* h, const double* c, double* viol) {
    if (int rc = check_handle(h)) return rc;
    if (!c || !viol) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_constraint_violation: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_violation(h->p, c, viol, h->stream));
    return QLN_OK;
}

int qln_eval_constraint_jvp(qln_handle* h, const double* Z, const double* v, double* y) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !v || !y) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_jvp: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_jvp(h->p, Z, v, y, h->stream));
    return QLN_OK;
}

int qln_eval_constraint_vjp(qln_handle* h, const double* Z, const double* lam, double* g) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !lam || !g) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_vjp: null pointer");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_constraint_vjp(h->p, Z, lam, g, h->stream));
    return QLN_OK;
}

int qln_gauss_newton_step(qln_handle* h, const double* Z, const double* c, double* dZ, int32_t max_iters, double rel_tol,
                          const double* radius, const double* col_scale, double* info) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c || !dZ) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_gauss_newton_step: null pointer");
    if (max_iters < 0 || !(rel_tol >= 0.0)) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_gauss_newton_step: bad max_iters / rel_tol");
    if (qln::gauss_newton_lds_bytes(h->dims.N) > 160 * 1024)
        return fail(QLN_ERR_UNSUPPORTED, "qln_gauss_newton_step: N = " + std::to_string(h->dims.N) +
                                             " does not fit one problem in the 160 KB of LDS of a CU (N <= 149)");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_gauss_newton_step(h->p, Z, c, dZ, max_iters, rel_tol, radius, col_scale, info, h->stream));
    return QLN_OK;
}

int qln_solve_default_options(qln_solve_options* o) {
    if (!o) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_solve_default_options: null");
    // Inexact inner solves: a few iLQR iterations per multiplier update, a gentle penalty growth.  Measured on 16 384
    // random N = 40 problems (profiles/r02_solve_sweep_options.txt): 60 inner iterations and rho x10 from 1 take a median
    // of 214 iterations; 8 inner iterations and rho x5 from 10 solve the same 16 384 with a median of 75 (max 144).
    o->max_outer = 80;
    o->max_inner = 6;
    o->tol_violation = 1e-6;
    o->inner_tol = 1e-7;
    o->rho0 = 3.0;
    o->rho_factor = 5.0;
    o->rho_max = 1e8;
    o->h_min = 0.001;   // src/moi.jl:59-60
    o->h_max = 0.02;
    o->theta_min = -M_PI / 2;  // src/moi.jl:55-56
    o->theta_max = M_PI / 2;
    o->q6_bounds = 1;
    o->exact_h_gradient = 0;
    o->h_prox = 1e4;
    o->rescue_outer = 20;
    return QLN_OK;
}

int qln_variable_bounds(int32_t N, const qln_solve_options* opt, double* x_l, double* x_u) {
    if (N < 2) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_variable_bounds: N must be >= 2");
    if (!x_l || !x_u) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_variable_bounds: null output");
    qln_solve_options o;
    qln_solve_default_options(&o);
    if (opt) o = *opt;
    const int64_t n_nlp = 20 * (int64_t)N - 5;
    for (int64_t i = 0; i < n_nlp; ++i) {
        x_l[i] = -HUGE_VAL;
        x_u[i] = HUGE_VAL;
    }
    for (int32_t k = 0; k < N; ++k) {  // 0-based knot; the reference's indices are 1-based, k = 1..N
        x_l[20 * k + 2] = o.theta_min;  // src/moi.jl:55-56
        x_u[20 * k + 2] = o.theta_max;
        if (k < N - 1) {
            x_l[20 * k + 19] = o.h_min;  // src/moi.jl:59-60
            x_u[20 * k + 19] = o.h_max;
            if (o.q6_bounds) {           // src/moi.jl:64-65: 22+20(k-1), 24+20(k-1) 1-based = yb_{k+1}, x1_{k+1} (quirk Q6)
                x_l[20 * k + 21] = 0.0;
                x_l[20 * k + 23] = 0.0;
            }
        }
    }
    return QLN_OK;
}

int qln_solve(qln_handle* h, double* Z, const qln_solve_options* opt, double* info) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_solve: null Z");
    qln_solve_options o;
    qln_solve_default_options(&o);
    if (opt) o = *opt;
    if (o.max_outer < 0 || o.max_inner < 1 || !(o.tol_violation > 0) || !(o.rho0 > 0) || !(o.rho_factor > 1) || !(o.rho_max >= o.rho0) ||
        !(o.h_min > 0) || !(o.h_max >= o.h_min) || !(o.theta_max > o.theta_min) || !(o.inner_tol >= 0) ||
        !(o.h_prox >= 0) || o.rescue_outer < 0)
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_solve: bad options");
    if (qln::ilqr_lds_bytes(h->dims.N) > 160 * 1024)
        return fail(QLN_ERR_UNSUPPORTED, "qln_solve: N = " + std::to_string(h->dims.N) + " does not fit one problem in the 160 KB of LDS of a CU");
    if (int rc = bind_device(h)) return rc;
    const size_t need = qln::ilqr_scratch_doubles(h->dims.B, h->dims.N);
    if (h->solve_scratch_n < need) {
        if (h->solve_scratch) {
            QLN_HIP(hipStreamSynchronize(h->stream));
            QLN_HIP(hipFree(h->solve_scratch));
            h->solve_scratch = nullptr;
            h->solve_scratch_n = 0;
        }
        QLN_HIP(hipMalloc(reinterpret_cast<void**>(&h->solve_scratch), need * sizeof(double)));
        h->solve_scratch_n = need;
    }
    qln::SolveParams sp;
    sp.max_outer = o.max_outer;
    sp.max_inner = o.max_inner;
    sp.tol = o.tol_violation;
    sp.inner_tol = o.inner_tol;
    sp.rho0 = o.rho0;
    sp.rho_factor = o.rho_factor;
    sp.rho_max = o.rho_max;
    sp.mu0 = 1e-6;
    sp.mu_min = 1e-8;
    sp.mu_max = 1e6;
    sp.h_lo = o.h_min;
    sp.h_hi = o.h_max;
    sp.th_lo = o.theta_min;
    sp.th_hi = o.theta_max;
    sp.h_prox = o.h_prox;
    sp.q6 = o.q6_bounds;
    sp.exact_h = o.exact_h_gradient;
    sp.rescue_outer = o.rescue_outer;
    QLN_HIP(qln::launch_al_ilqr(h->p, sp, Z, info, h->solve_scratch, h->stream));
    return QLN_OK;
}

static qln::DropStateSampler sampler_of(const qln_drop_state_sampler* s) {
    qln::DropStateSampler d;
    d.state_hi = s->pcg_state[0], d.state_lo = s->pcg_state[1];
    d.inc_hi = s->pcg_inc[0], d.inc_lo = s->pcg_inc[1];
    d.stream_offset = s->stream_offset;
    std::memcpy(d.x0_template, s->x0_template, sizeof d.x0_template);
    const double* r[4] = {s->theta_deg, s->y2, s->drop_height, s->omega};
    for (int a = 0; a < 4; ++a) {
        d.lo[a] = r[a][0];
        d.range[a] = r[a][1] - r[a][0];  // numpy: low + (high - low) * u
    }
    d.deg2rad = M_PI / 180.0;  // numpy.deg2rad: x * (pi / 180)
    d.two_g = s->two_g;
    return d;
}

int qln_sample_drop_states(qln_handle* h, const qln_drop_state_sampler* s) {
    if (int rc = check_handle(h)) return rc;
    if (!s) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_sample_drop_states: null sampler");
    if (s->stream_offset < 0) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_sample_drop_states: negative stream_offset");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_sample_drop_states(h->p, sampler_of(s), h->d_bnd, h->stream));
    return QLN_OK;
}

int qln_sample_bounded_integers(int device, const uint64_t pcg_state[2], const uint64_t pcg_inc[2], int64_t draw_offset, int32_t low,
                                int32_t high, int64_t count, int32_t* out, int64_t* rejected) {
    if (!pcg_state || !pcg_inc || !out || !rejected) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_sample_bounded_integers: null pointer");
    if (draw_offset < 0 || count < 0 || high <= low) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_sample_bounded_integers: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(QLN_ERR_NO_DEVICE, "qln_sample_bounded_integers: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_sample_bounded_integers: bad device ordinal");
    *rejected = 0;
    if (count == 0) return QLN_OK;
    if ((int64_t)high - low == 1) {  // one possible value: numpy fills it in without touching the stream
        for (int64_t i = 0; i < count; ++i) out[i] = low;
        return QLN_OK;
    }
    QLN_HIP(hipSetDevice(device));
    qln::DropStateSampler d{};
    d.state_hi = pcg_state[0], d.state_lo = pcg_state[1];
    d.inc_hi = pcg_inc[0], d.inc_lo = pcg_inc[1];
    d.stream_offset = draw_offset;
    int32_t* d_out = nullptr;
    unsigned long long* d_rej = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)count * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rej), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_rej, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = qln::launch_bounded_integers(d, (uint32_t)((int64_t)high - low), low, count, d_out, d_rej, nullptr);
    unsigned long long rej = 0;
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&rej, d_rej, sizeof rej, hipMemcpyDeviceToHost);
    if (d_out) (void)hipFree(d_out);
    if (d_rej) (void)hipFree(d_rej);
    if (e != hipSuccess) return fail(QLN_ERR_HIP, std::string("qln_sample_bounded_integers: ") + hipGetErrorString(e));
    *rejected = (int64_t)rej;
    return QLN_OK;
}

int qln_perturb_point(qln_handle* h, const qln_drop_state_sampler* s, double* Z, double sigma, double h_min, double h_max,
                      int redraw_h) {
    if (int rc = check_handle(h)) return rc;
    if (!s || !Z) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_perturb_point: null pointer");
    if (s->stream_offset < 0 || !(sigma >= 0) || !(h_max >= h_min)) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_perturb_point: bad argument");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_perturb_point(h->p, sampler_of(s), Z, sigma, h_min, h_max, redraw_h, h->stream));
    return QLN_OK;
}

int qln_get_boundary_states(qln_handle* h, double* x0, double* xf) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = bind_device(h)) return rc;
    std::vector<double> bnd((size_t)h->dims.B * 30);
    QLN_HIP(hipStreamSynchronize(h->stream));
    QLN_HIP(hipMemcpy(bnd.data(), h->d_bnd, bnd.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int32_t b = 0; b < h->dims.B; ++b) {
        if (x0) std::memcpy(x0 + (size_t)b * 15, &bnd[(size_t)b * 30], 15 * sizeof(double));
        if (xf) std::memcpy(xf + (size_t)b * 15, &bnd[(size_t)b * 30 + 15], 15 * sizeof(double));
    }
    return QLN_OK;
}

int qln_solve_host(qln_handle* h, double* Z, const qln_solve_options* opt, double* info) {
    if (int rc = check_handle(h)) return rc;
    if (!Z) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_solve_host: null Z");
    if (int rc = bind_device(h)) return rc;
    if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
    double* d_info = nullptr;
    const size_t ninfo = (size_t)h->dims.B * QLN_SOLVE_INFO_STRIDE;
    if (info) QLN_HIP(hipMalloc(reinterpret_cast<void**>(&d_info), ninfo * sizeof(double)));
    int rc = QLN_OK;
    hipError_t e = hipMemcpyAsync(h->s_Z, Z, h->dims.z_total * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) rc = qln_solve(h, h->s_Z, opt, d_info);
    if (e == hipSuccess && rc == QLN_OK) e = hipMemcpyAsync(Z, h->s_Z, h->dims.z_total * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && rc == QLN_OK && info) e = hipMemcpyAsync(info, d_info, ninfo * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && rc == QLN_OK) e = hipStreamSynchronize(h->stream);
    if (d_info) (void)hipFree(d_info);
    if (rc != QLN_OK) return rc;
    if (e != hipSuccess) return fail(QLN_ERR_HIP, std::string("qln_solve_host: ") + hipGetErrorString(e));
    return QLN_OK;
}

int qln_initial_guess(qln_handle* h, double* Z) {
    if (int rc = check_handle(h)) return rc;
    if (!Z) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_initial_guess: null Z");
    for (int32_t b = 0; b < h->dims.B; ++b)
        if (h->k_trans[b] < 2)
            return fail(QLN_ERR_UNSUPPORTED, "qln_initial_guess: k_trans < 2 at problem " + std::to_string(b) +
                                                 " (the notebook's rule divides by k_trans - 1)");
    if (int rc = bind_device(h)) return rc;
    QLN_HIP(qln::launch_initial_guess(h->p, Z, h->stream));
    return QLN_OK;
}

// ------------------------------------------------------------------ host-pointer (MOI) mode

int qln_eval_objective_host(qln_handle* h, const double* Z, double* f) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !f) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_objective_host: null pointer");
    if (int rc = bind_device(h)) return rc;
    if (h->zero_copy) {
        if (int rc = ensure_mapped(&h->m_Z, h->dims.z_total)) return rc;
        if (int rc = ensure_mapped(&h->m_f, h->dims.B)) return rc;
        std::memcpy(h->m_Z.host, Z, h->dims.z_total * sizeof(double));
        QLN_HIP(qln::launch_objective(h->p, h->m_Z.dev, h->m_f.dev, h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(f, h->m_f.host, h->dims.B * sizeof(double));
        return QLN_OK;
    }
    if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
    if (int rc = ensure(&h->s_f, h->dims.B)) return rc;
    QLN_HIP(hipMemcpyAsync(h->s_Z, Z, h->dims.z_total * sizeof(double), hipMemcpyHostToDevice, h->stream));
    QLN_HIP(qln::launch_objective(h->p, h->s_Z, h->s_f, h->stream));
    QLN_HIP(hipMemcpyAsync(f, h->s_f, h->dims.B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QLN_HIP(hipStreamSynchronize(h->stream));
    return QLN_OK;
}

int qln_eval_objective_gradient_host(qln_handle* h, const double* Z, double* grad) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = check_cost(h)) return rc;
    if (!Z || !grad) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_objective_gradient_host: null pointer");
    if (int rc = bind_device(h)) return rc;
    if (h->zero_copy) {
        if (int rc = ensure_mapped(&h->m_Z, h->dims.z_total)) return rc;
        if (int rc = ensure_mapped(&h->m_grad, h->dims.z_total)) return rc;
        std::memcpy(h->m_Z.host, Z, h->dims.z_total * sizeof(double));
        QLN_HIP(qln::launch_objective_gradient(h->p, h->m_Z.dev, h->m_grad.dev, h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(grad, h->m_grad.host, h->dims.z_total * sizeof(double));
        return QLN_OK;
    }
    if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
    if (int rc = ensure(&h->s_grad, h->dims.z_total)) return rc;
    QLN_HIP(hipMemcpyAsync(h->s_Z, Z, h->dims.z_total * sizeof(double), hipMemcpyHostToDevice, h->stream));
    QLN_HIP(hipMemsetAsync(h->s_grad, 0, h->dims.z_total * sizeof(double), h->stream));
    QLN_HIP(qln::launch_objective_gradient(h->p, h->s_Z, h->s_grad, h->stream));
    QLN_HIP(hipMemcpyAsync(grad, h->s_grad, h->dims.z_total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QLN_HIP(hipStreamSynchronize(h->stream));
    return QLN_OK;
}

int qln_eval_constraint_host(qln_handle* h, const double* Z, double* c) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_host: null pointer");
    if (int rc = bind_device(h)) return rc;
    if (h->zero_copy) {
        if (int rc = ensure_mapped(&h->m_Z, h->dims.z_total)) return rc;
        if (int rc = ensure_mapped(&h->m_c, h->dims.c_total)) return rc;
        std::memcpy(h->m_Z.host, Z, h->dims.z_total * sizeof(double));
        QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, h->m_Z.dev, h->m_c.dev, nullptr, qln::kLaunchSplit, h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(c, h->m_c.host, h->dims.c_total * sizeof(double));
        return QLN_OK;
    }
    if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
    if (int rc = ensure(&h->s_c, h->dims.c_total)) return rc;
    QLN_HIP(hipMemcpyAsync(h->s_Z, Z, h->dims.z_total * sizeof(double), hipMemcpyHostToDevice, h->stream));
    QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, h->s_Z, h->s_c, nullptr, 0, h->stream));
    QLN_HIP(hipMemcpyAsync(c, h->s_c, h->dims.c_total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QLN_HIP(hipStreamSynchronize(h->stream));
    return QLN_OK;
}

int qln_eval_constraint_jacobian_host(qln_handle* h, const double* Z, double* vals) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !vals) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_jacobian_host: null pointer");
    if (int rc = bind_device(h)) return rc;
    if (h->zero_copy) {
        if (int rc = ensure_mapped(&h->m_Z, h->dims.z_total)) return rc;
        if (int rc = ensure_mapped(&h->m_vals, h->dims.j_total)) return rc;
        std::memcpy(h->m_Z.host, Z, h->dims.z_total * sizeof(double));
        QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, h->m_Z.dev, nullptr, h->m_vals.dev,
                                                QLN_JAC_WRITE_CONSTANTS | qln::kLaunchSplit, h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        std::memcpy(vals, h->m_vals.host, h->dims.j_total * sizeof(double));
        return QLN_OK;
    }
    if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
    if (int rc = ensure(&h->s_vals, h->dims.j_total)) return rc;
    QLN_HIP(hipMemcpyAsync(h->s_Z, Z, h->dims.z_total * sizeof(double), hipMemcpyHostToDevice, h->stream));
    QLN_HIP(qln::launch_constraint_jacobian(h->p, 0, h->p.B, h->s_Z, nullptr, h->s_vals, QLN_JAC_WRITE_CONSTANTS,
                                            h->stream));
    QLN_HIP(hipMemcpyAsync(vals, h->s_vals, h->dims.j_total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    QLN_HIP(hipStreamSynchronize(h->stream));
    return QLN_OK;
}

int qln_eval_constraint_jacobian_dense_host(qln_handle* h, int32_t b, const double* Z, double* jac) {
    // MOI.eval_constraint_jacobian with use_sparse_jacobian=false (src/moi.jl:15-24): the values are
    // computed on the GPU; the host only scatters them into the caller's column-major matrix,
    // touching exactly the write-set of jac_c! (src/constraints.jl:219-274).
    if (int rc = check_problem(h, b)) return rc;
    if (!Z || !jac) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_eval_constraint_jacobian_dense_host: null pointer");
    if (int rc = bind_device(h)) return rc;
    const int32_t N = h->dims.N, kt = h->k_trans[b];
    const int32_t nnz = nnz_of(N, kt, h->p.jac_format);
    const int64_t m = m_nlp_of(N, kt);
    const double* v = nullptr;
    if (h->zero_copy) {
        if (int rc = ensure_mapped(&h->m_Z, h->dims.z_total)) return rc;
        if (int rc = ensure_mapped(&h->m_vals, h->dims.j_total)) return rc;
        std::memcpy(h->m_Z.host + (int64_t)b * h->dims.z_stride, Z, h->dims.n_nlp * sizeof(double));
        QLN_HIP(qln::launch_constraint_jacobian(h->p, b, 1, h->m_Z.dev, nullptr, h->m_vals.dev,
                                                QLN_JAC_WRITE_CONSTANTS | qln::kLaunchSplit, h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        v = h->m_vals.host + h->j_off[b];
    } else {
        if (int rc = ensure(&h->s_Z, h->dims.z_total)) return rc;
        if (int rc = ensure(&h->s_vals, h->dims.j_total)) return rc;
        h->h_vals_one.resize(nnz);
        QLN_HIP(hipMemcpyAsync(h->s_Z + (int64_t)b * h->dims.z_stride, Z, h->dims.n_nlp * sizeof(double), hipMemcpyHostToDevice,
                               h->stream));
        QLN_HIP(qln::launch_constraint_jacobian(h->p, b, 1, h->s_Z, nullptr, h->s_vals, QLN_JAC_WRITE_CONSTANTS, h->stream));
        QLN_HIP(hipMemcpyAsync(h->h_vals_one.data(), h->s_vals + h->j_off[b], nnz * sizeof(double), hipMemcpyDeviceToHost,
                               h->stream));
        QLN_HIP(hipStreamSynchronize(h->stream));
        v = h->h_vals_one.data();
    }
    if (h->dense_map.empty()) h->dense_map.resize((size_t)h->dims.B);
    qln_handle::DenseMap& dm = h->dense_map[(size_t)b];
    if (dm.at.empty()) {
        std::vector<int32_t> rows(nnz), cols(nnz);
        if (int rc = qln_jacobian_structure(h, b, rows.data(), cols.data())) return rc;
        dm.at.resize(nnz);
        for (int32_t e = 0; e < nnz; ++e) dm.at[e] = rows[e] + m * (int64_t)cols[e];
        // D[ci, xi[k+1]] .= -I(n) assigns the whole 15x15 block (explicit zeros off the diagonal)
        int32_t ci[14];
        cinds_of(N, kt, ci);
        for (int32_t k = 0; k < N - 1; ++k)
            for (int32_t c = 0; c < 15; ++c)
                for (int32_t r = 0; r < 15; ++r)
                    if (r != c) dm.zeros.push_back((ci[4] - 1 + 15 * k + r) + m * (int64_t)(20 * (k + 1) + c));
        if (h->p.jac_format == QLN_JAC_FORMAT_STRUCTURAL)
            // D[ci, [xi[k]; ui[k]]] .= J assigns the whole 15x20 block: the entries the structural format leaves out are 0
            for (int32_t k = 0; k < N - 1; ++k)
                for (int32_t c = 0; c < 20; ++c)
                    for (int32_t r = 0; r < 15; ++r) dm.zeros.push_back((ci[4] - 1 + 15 * k + r) + m * (int64_t)(20 * k + c));
    }
    for (int64_t i : dm.zeros) jac[i] = 0.0;
    for (int32_t e = 0; e < nnz; ++e) jac[dm.at[e]] = v[e];
    return QLN_OK;
}

// ------------------------------------------------------------------ placement-aware allocation

static int time_fused(qln_handle* h, const double* Z, double* c, double* vals, int warmup, int iters, float* best) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return fail(QLN_ERR_HIP, "hipEventCreate failed");
    }
    int rc = QLN_OK;
    *best = 1e30f;
    for (int i = 0; i < warmup + iters && rc == QLN_OK; ++i) {
        (void)hipEventRecord(e0, h->stream);
        if (qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, 0, h->stream) != hipSuccess) rc = fail(QLN_ERR_HIP, "launch failed");
        (void)hipEventRecord(e1, h->stream);
        if (rc == QLN_OK && hipEventSynchronize(e1) != hipSuccess) rc = fail(QLN_ERR_HIP, "hipEventSynchronize failed");
        float ms = 0.f;
        if (rc == QLN_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(QLN_ERR_HIP, "hipEventElapsedTime failed");
        if (rc == QLN_OK && i >= warmup) *best = std::min(*best, ms);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int qln_vals_alloc_placed(qln_handle* h, const double* Z, double* c, double** vals, float* ms_best) {
    return qln_vals_alloc_placed_budget(h, Z, c, (int64_t)64 << 30, vals, ms_best);
}

int qln_vals_placed_address_space(int64_t* retired_bytes, int64_t* cap_bytes) {
    if (retired_bytes) *retired_bytes = (int64_t)g_va_retired.load();
    if (cap_bytes) *cap_bytes = (int64_t)kVaCapBytes;
    return QLN_OK;
}

int qln_vals_alloc_placed_budget(qln_handle* h, const double* Z, double* c, int64_t transient_bytes, double** vals, float* ms_best) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !vals) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_vals_alloc_placed: null pointer");
    if (transient_bytes < 0) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_vals_alloc_placed: negative transient budget");
    *vals = nullptr;
    if (int rc = bind_device(h)) return rc;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = h->device;
    size_t gran = 0;
    QLN_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    // Physical chunks of 256 MiB: launch times are the same from 2 MiB to 1 GiB (profiles/r01_vmm_placement.txt), and a
    // finer chunk makes the window scan finer.  (Round 1 recorded "chunks of 2 GiB and more raise a GPU memory access
    // fault": that program had unmapped, freed and re-reserved the same range a dozen times before it got to that size,
    // i.e. it ran into the address-reuse defect described at kReturnVirtualRange.  On a fresh range 2-GiB chunks are
    // fine: bench/vmm_big_chunk.cpp, profiles/r02_vmm_big_chunk.txt.)
    const size_t chunk = round_up((int64_t)256 << 20, (int64_t)gran);
    const size_t bytes = (size_t)h->dims.j_total * 8;
    const size_t need = (bytes + chunk - 1) / chunk;                     // chunks under vals
    const size_t region = ((size_t)32 << 30) / chunk;                    // period of the speed classes, in chunks
    const bool small = bytes < ((size_t)1 << 30);                        // nothing to gain: one plain mapping
    size_t free_b = 0, total_b = 0;
    QLN_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = free_b / 10 * 9 / chunk;                       // chunks this call may hold at once
    if (budget < need + 1) return fail(QLN_ERR_HIP, "qln_vals_alloc_placed: not enough free device memory");

    // Everything this call holds until it has succeeded.  abandon() is the one way out on failure: synchronise the
    // device, unmap what is mapped, release every handle, hand the range back, free the scratch constraint buffer.
    std::vector<hipMemGenericAllocationHandle_t> slab;   // consecutive physical chunks (consecutive allocations are
    std::vector<char> mapped;                            // consecutive in device memory as far as launch times can tell)
    char* va = nullptr;
    size_t va_size = 0;
    double* scratch_c = nullptr;
    auto abandon = [&](int code) {
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i < slab.size(); ++i) {
            if (i < mapped.size() && mapped[i]) (void)hipMemUnmap(va + i * chunk, chunk);
            if (slab[i]) (void)hipMemRelease(slab[i]);
        }
        if (va && kReturnVirtualRange) (void)hipMemAddressFree(va, va_size);
        if (scratch_c) (void)hipFree(scratch_c);
        return code;
    };
    auto fail_hip = [&](const char* what, hipError_t e) {
        return abandon(fail(QLN_ERR_HIP, std::string("qln_vals_alloc_placed: ") + what + ": " + hipGetErrorString(e)));
    };

    // the timed launches write a constraint vector: the caller's c if one is given (it is overwritten), else a scratch
    if (!c) {
        if (hipError_t e = hipMalloc(reinterpret_cast<void**>(&scratch_c), (size_t)h->dims.c_total * 8); e != hipSuccess)
            return fail_hip("hipMalloc(scratch c)", e);
        c = scratch_c;
    }
    // two region lengths beyond the buffer: the slab then contains two boundaries, i.e. two chances of a clean
    // straddling window
    // (a caller with less memory to spare passes a smaller transient budget: one region length gives one boundary)
    (void)region;
    const size_t extra = ((size_t)transient_bytes + chunk - 1) / chunk;
    size_t nslab = small ? need : std::min(budget, need + extra + 1);
    if (g_va_retired.load() + (uint64_t)nslab * chunk > kVaCapBytes)
        return abandon(fail(QLN_ERR_UNSUPPORTED,
                            "qln_vals_alloc_placed: this process has retired " + std::to_string(g_va_retired.load() >> 30) +
                                " GiB of virtual address space in placed allocations (ranges are never returned: an unmapped address "
                                "must not be mapped again on this stack); the cap is " + std::to_string(kVaCapBytes >> 30) +
                                " GiB -- reuse placed buffers instead of re-allocating them, or use plain hipMalloc memory"));
    for (size_t i = 0; i < nslab; ++i) {
        hipMemGenericAllocationHandle_t hd;
        const hipError_t e = hipMemCreate(&hd, chunk, &prop, 0);
        if (e != hipSuccess) {
            if (slab.size() >= need + 1) break;  // less than planned, still usable
            return fail_hip("hipMemCreate", e);
        }
        slab.push_back(hd);
    }
    nslab = slab.size();
    mapped.assign(nslab, 0);
    va_size = nslab * chunk;
    {
        void* p = nullptr;
        if (hipError_t e = hipMemAddressReserve(&p, va_size, 0, nullptr, 0); e != hipSuccess) return fail_hip("hipMemAddressReserve", e);
        va = static_cast<char*>(p);
        if (!kReturnVirtualRange) g_va_retired.fetch_add((uint64_t)va_size);  // whatever happens next, this range is spent
    }
    for (size_t i = 0; i < nslab; ++i) {
        if (hipError_t e = hipMemMap(va + i * chunk, chunk, 0, slab[i], 0); e != hipSuccess) return fail_hip("hipMemMap", e);
        mapped[i] = 1;
    }
    {
        hipMemAccessDesc ad = {};
        ad.location = prop.location;
        ad.flags = hipMemAccessFlagsProtReadWrite;
        if (hipError_t e = hipMemSetAccess(va, va_size, &ad, 1); e != hipSuccess) return fail_hip("hipMemSetAccess", e);
    }

    // The slab as it lies, the fused launch timed on its windows: the best one straddles a region boundary (XCDs 0-3
    // write one region, XCDs 4-7 the next).  Giving every XCD's range a region of its own (ranges 16 GiB apart) was
    // tried as a second layout: 1.10 ms against 1.08-1.10 ms for the best window, so it is not built.
    size_t best_off = 0;  // chunks
    float best = 1e30f;
    {
        auto probe = [&](size_t off) {
            float ms = 0.f;
            const int rc = time_fused(h, Z, c, reinterpret_cast<double*>(va + off * chunk), 1, 2, &ms);
            if (rc == QLN_OK && ms < best) {
                best = ms;
                best_off = off;
            }
            return rc;
        };
        int rc = QLN_OK;
        if (small) {
            rc = probe(0);
        } else {
            for (size_t off = 0; off + need <= nslab && rc == QLN_OK; off += 4) rc = probe(off);  // 1-GiB steps
            const size_t centre = best_off;
            for (int k = -3; k <= 3 && rc == QLN_OK; ++k) {
                const int64_t off = (int64_t)centre + k;
                if (k != 0 && off >= 0 && (size_t)off + need <= nslab) rc = probe((size_t)off);
            }
        }
        if (rc != QLN_OK) return abandon(rc);
    }
    // Every chunk outside the window goes back to the driver; the window keeps its addresses.  time_fused waited for
    // each of its launches, but nothing is assumed: the device is idle before the first unmap.
    if (hipError_t e = hipDeviceSynchronize(); e != hipSuccess) return fail_hip("hipDeviceSynchronize", e);
    for (size_t i = 0; i < nslab; ++i) {
        if (i >= best_off && i < best_off + need) continue;
        if (hipError_t e = hipMemUnmap(va + i * chunk, chunk); e != hipSuccess) return fail_hip("hipMemUnmap", e);
        mapped[i] = 0;
    }
    for (size_t i = 0; i < nslab; ++i) {
        if (i >= best_off && i < best_off + need) continue;
        const hipError_t e = hipMemRelease(slab[i]);
        slab[i] = nullptr;  // released (or lost): abandon() must not release it again
        if (e != hipSuccess) return fail_hip("hipMemRelease", e);
    }
    if (scratch_c) {
        const hipError_t e = hipFree(scratch_c);
        scratch_c = nullptr;
        if (e != hipSuccess) return fail_hip("hipFree(scratch c)", e);
    }
    qln_handle::Placed P;
    P.chunk = chunk;
    P.va = va;
    P.va_size = va_size;
    P.first = best_off;
    P.chunks.assign(slab.begin() + (long)best_off, slab.begin() + (long)(best_off + need));
    P.vals = reinterpret_cast<double*>(va + best_off * chunk);
    P.scanned = nslab;
    h->placed.push_back(P);
    *vals = P.vals;
    if (ms_best) *ms_best = best;
    return QLN_OK;
}

int qln_vals_placed_info(const qln_handle* h, const double* vals, int64_t* chunk_bytes, int64_t* chunks_scanned,
                         int64_t* window_first_chunk) {
    if (int rc = check_handle(h)) return rc;
    for (const auto& p : h->placed)
        if (p.vals == vals) {
            if (chunk_bytes) *chunk_bytes = (int64_t)p.chunk;
            if (chunks_scanned) *chunks_scanned = (int64_t)p.scanned;
            if (window_first_chunk) *window_first_chunk = (int64_t)p.first;
            return QLN_OK;
        }
    return fail(QLN_ERR_INVALID_ARGUMENT, "qln_vals_placed_info: not a buffer of this handle");
}

int qln_vals_free_placed(qln_handle* h, double* vals) {
    if (int rc = check_handle(h)) return rc;
    if (int rc = bind_device(h)) return rc;
    for (size_t i = 0; i < h->placed.size(); ++i)
        if (h->placed[i].vals == vals) {
            // the whole device, not the handle's stream: the buffer was handed out as plain memory and any stream may
            // have work on it (torch's current stream; a stream set with qln_set_stream after the last launch)
            QLN_HIP(hipDeviceSynchronize());
            const int rc = release_placed(h->placed[i]);
            h->placed.erase(h->placed.begin() + (long)i);
            return rc;
        }
    return fail(QLN_ERR_INVALID_ARGUMENT, "qln_vals_free_placed: not a buffer of this handle");
}

// ------------------------------------------------------------------ measurement

int qln_time_constraint_and_jacobian(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags, int32_t warmup,
                                     int32_t iters, float* ms_each) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c || !ms_each || iters < 1 || warmup < 0)
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_time_constraint_and_jacobian: bad argument");
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    flags &= QLN_JAC_WRITE_CONSTANTS;
    std::vector<hipEvent_t> ev(2 * (size_t)iters, nullptr);
    int rc = QLN_OK;
    for (auto& e : ev)
        if (rc == QLN_OK && hipEventCreate(&e) != hipSuccess) rc = fail(QLN_ERR_HIP, "hipEventCreate failed");
    for (int32_t i = 0; i < warmup && rc == QLN_OK; ++i)
        if (qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, flags, h->stream) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "warmup launch failed");
    for (int32_t i = 0; i < iters && rc == QLN_OK; ++i) {
        (void)hipEventRecord(ev[2 * i], h->stream);
        if (qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, flags, h->stream) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "timed launch failed");
        (void)hipEventRecord(ev[2 * i + 1], h->stream);
    }
    if (rc == QLN_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(QLN_ERR_HIP, "stream synchronize failed");
    for (int32_t i = 0; i < iters && rc == QLN_OK; ++i)
        if (hipEventElapsedTime(&ms_each[i], ev[2 * i], ev[2 * i + 1]) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "hipEventElapsedTime failed");
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    return rc;
}

int qln_time_constraint_and_jacobian_total(qln_handle* h, const double* Z, double* c, double* vals, uint32_t flags, int32_t warmup,
                                           int32_t iters, float* ms_total) {
    if (int rc = check_handle(h)) return rc;
    if (!Z || !c || !ms_total || iters < 1 || warmup < 0)
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_time_constraint_and_jacobian_total: bad argument");
    if (int rc = check_vals(vals)) return rc;
    if (int rc = bind_device(h)) return rc;
    flags &= QLN_JAC_WRITE_CONSTANTS;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = QLN_OK;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(QLN_ERR_HIP, "hipEventCreate failed");
    for (int32_t i = 0; i < warmup && rc == QLN_OK; ++i)
        if (qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, flags, h->stream) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "warmup launch failed");
    if (rc == QLN_OK) (void)hipEventRecord(e0, h->stream);
    for (int32_t i = 0; i < iters && rc == QLN_OK; ++i)
        if (qln::launch_constraint_jacobian(h->p, 0, h->p.B, Z, c, vals, flags, h->stream) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "timed launch failed");
    if (rc == QLN_OK) (void)hipEventRecord(e1, h->stream);
    if (rc == QLN_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(QLN_ERR_HIP, "stream synchronize failed");
    if (rc == QLN_OK && hipEventElapsedTime(ms_total, e0, e1) != hipSuccess) rc = fail(QLN_ERR_HIP, "hipEventElapsedTime failed");
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
