// qln_multi.cpp -- include/qln_multi.h: batch sharding over the GPUs of one node and the single end-of-job RCCL
// gather, on top of the single-GPU C ABI (libqln_hip.so).  Built as libqln_multi.so so that the single-GPU library
// carries no RCCL dependency.
//
// There is no data-path collective: every device evaluates its contiguous range of problems with its own handle on
// its own stream.  The gather is n send/recv pairs inside one ncclGroupStart/End (per-rank counts: shards are ragged).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <cstring>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "../../include/qln_multi.h"

namespace {

int fail(int code, const std::string& msg) { return qln_set_last_error(code, msg.c_str()); }

#define QM_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(QLN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define QM_NCCL(expr)                                                                            \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) return fail(QLN_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define QM_OK(expr)                      \
    do {                                 \
        int rc_ = (expr);                \
        if (rc_ != QLN_OK) return rc_;   \
    } while (0)

}  // namespace

struct qln_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    long long* d_counts = nullptr;  // [world] gathered counts, then [1] this rank's count
    double* d_val = nullptr;        // scratch for qln_comm_max
};

struct qln_multi {
    struct Shard {
        int device = 0;
        int64_t lo = 0, hi = 0;
        qln_handle* h = nullptr;
        hipStream_t stream = nullptr;
        qln_dims dims{};
        double *Z = nullptr, *c = nullptr, *vals = nullptr, *f = nullptr, *viol = nullptr, *sinfo = nullptr;
        bool vals_placed = false;
        int64_t c_displ = 0;  // where the shard's constraint vector starts in the gathered one
        int64_t z_begin = 0;  // where the shard's rows start in the batch's host Z (doubles)
    };
    std::vector<Shard> shards;
    std::vector<ncclComm_t> comms;
    int64_t B = 0, z_stride = 0, c_total = 0;
    std::vector<int64_t> c_off;  // [B] offsets into the gathered constraint vector
    // gather buffers on the root device (allocated on first use)
    int root = -1;
    double *g_f = nullptr, *g_viol = nullptr, *g_c = nullptr;
    uint32_t gathered = 0;
    bool one_device = false;  // qln_multi_create_on_one_device: no RCCL clique, the gather's exchange is device copies
};

extern "C" {

int qln_shard_range(int64_t n, int rank, int world, int64_t* begin, int64_t* end) {
    if (world < 1 || rank < 0 || rank >= world || n < 0 || !begin || !end)
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_shard_range: bad argument");
    const int64_t base = n / world, rem = n % world;
    *begin = rank * base + std::min<int64_t>(rank, rem);
    *end = *begin + base + (rank < rem ? 1 : 0);
    return QLN_OK;
}

// ------------------------------------------------------------------------------------------ one process per GPU

int qln_comm_get_unique_id(void* id) {
    if (!id) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_get_unique_id: null id");
    static_assert(sizeof(ncclUniqueId) == QLN_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId u;
    QM_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return QLN_OK;
}

int qln_comm_init_rank(const void* id, int rank, int world, int device, qln_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_init_rank: bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(QLN_ERR_NO_DEVICE, "qln_comm_init_rank: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_init_rank: bad device ordinal");
    QM_HIP(hipSetDevice(device));
    qln_comm* c = new (std::nothrow) qln_comm();
    if (!c) return fail(QLN_ERR_HIP, "qln_comm_init_rank: out of host memory");
    c->rank = rank, c->world = world, c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    if (ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank); r != ncclSuccess) {
        delete c;
        return fail(QLN_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_counts), (size_t)(world + 1) * sizeof(long long)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_val), 2 * sizeof(double)) != hipSuccess) {
        qln_comm_destroy(c);
        return fail(QLN_ERR_HIP, "qln_comm_init_rank: hipMalloc failed");
    }
    *out = c;
    return QLN_OK;
}

int qln_comm_destroy(qln_comm* c) {
    if (!c) return QLN_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->d_val) (void)hipFree(c->d_val);
    delete c;
    return QLN_OK;
}

int qln_comm_rank(const qln_comm* c, int* rank, int* world) {
    if (!c) return fail(QLN_ERR_INVALID_ARGUMENT, "null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return QLN_OK;
}

int qln_comm_exchange_counts(qln_comm* c, int64_t count, int64_t* counts_out) {
    if (!c || !counts_out || count < 0) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_exchange_counts: bad argument");
    QM_HIP(hipSetDevice(c->device));
    const long long mine = count;
    std::vector<long long> counts((size_t)c->world);
    QM_HIP(hipMemcpy(c->d_counts + c->world, &mine, sizeof mine, hipMemcpyHostToDevice));
    QM_NCCL(ncclAllGather(c->d_counts + c->world, c->d_counts, 1, ncclInt64, c->comm, nullptr));
    QM_HIP(hipStreamSynchronize(nullptr));
    QM_HIP(hipMemcpy(counts.data(), c->d_counts, counts.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int r = 0; r < c->world; ++r) counts_out[r] = counts[(size_t)r];
    return QLN_OK;
}

int qln_comm_gather(qln_comm* c, const double* send, int64_t count, double* recv, const int64_t* counts, int root,
                    void* hip_stream) {
    if (!c) return fail(QLN_ERR_INVALID_ARGUMENT, "null comm");
    if (count < 0 || root < 0 || root >= c->world || (count > 0 && !send) || (c->rank == root && !counts))
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_gather: bad argument");
    if (c->rank == root) {
        if (counts[root] != count) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_gather: counts[root] != count");
        int64_t total = 0;
        for (int r = 0; r < c->world; ++r) total += counts[r];
        if (total > 0 && !recv) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_gather: null receive buffer on the root");
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    QM_HIP(hipSetDevice(c->device));
    // one send per rank, one receive per rank on the root, as one group: what ncclGather does inside, with per-rank
    // counts (shards are ragged: per-problem k_trans, batches that do not divide by the world size)
    QM_NCCL(ncclGroupStart());
    ncclResult_t res = ncclSuccess;
    if (count > 0) res = ncclSend(send, (size_t)count, ncclDouble, root, c->comm, s);
    if (c->rank == root) {
        int64_t displ = 0;
        for (int r = 0; r < c->world && res == ncclSuccess; ++r) {
            if (counts[r] > 0) res = ncclRecv(recv + displ, (size_t)counts[r], ncclDouble, r, c->comm, s);
            displ += counts[r];
        }
    }
    const ncclResult_t end = ncclGroupEnd();
    if (res != ncclSuccess) return fail(QLN_ERR_COMM, std::string("qln_comm_gather: ncclSend/ncclRecv: ") + ncclGetErrorString(res));
    if (end != ncclSuccess) return fail(QLN_ERR_COMM, std::string("qln_comm_gather: ncclGroupEnd: ") + ncclGetErrorString(end));
    return QLN_OK;
}

int qln_comm_max(qln_comm* c, double* value) {
    if (!c || !value) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_comm_max: null argument");
    QM_HIP(hipSetDevice(c->device));
    QM_HIP(hipMemcpy(c->d_val, value, sizeof(double), hipMemcpyHostToDevice));
    QM_NCCL(ncclAllReduce(c->d_val, c->d_val + 1, 1, ncclDouble, ncclMax, c->comm, nullptr));
    QM_HIP(hipStreamSynchronize(nullptr));
    QM_HIP(hipMemcpy(value, c->d_val + 1, sizeof(double), hipMemcpyDeviceToHost));
    return QLN_OK;
}

int qln_comm_barrier(qln_comm* c) {
    double v = 0.0;
    return qln_comm_max(c, &v);
}

// ------------------------------------------------------------------------------------------ one process, n devices

int qln_multi_destroy(qln_multi* m) {
    if (!m) return QLN_OK;
    int rc = QLN_OK;
    for (auto& s : m->shards) {
        (void)hipSetDevice(s.device);
        (void)hipDeviceSynchronize();
    }
    for (ncclComm_t c : m->comms)
        if (c) (void)ncclCommDestroy(c);
    if (m->root >= 0) {
        (void)hipSetDevice(m->shards[(size_t)m->root].device);
        for (double* p : {m->g_f, m->g_viol, m->g_c})
            if (p) (void)hipFree(p);
    }
    for (auto& s : m->shards) {
        (void)hipSetDevice(s.device);
        if (s.vals && !s.vals_placed) (void)hipFree(s.vals);  // placed buffers are released by qln_destroy
        for (double* p : {s.Z, s.c, s.f, s.viol, s.sinfo})
            if (p) (void)hipFree(p);
        if (s.h)
            if (int r = qln_destroy(s.h); r != QLN_OK && rc == QLN_OK) rc = r;
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete m;
    return rc;
}

// the descriptor of shard [lo, hi): the batch's host arrays cut at problem lo
static qln_batch_desc shard_desc(const qln_batch_desc& d, const qln_shard_plan& p) {
    qln_batch_desc sd = d;
    sd.B = (int32_t)(p.b_end - p.b_begin);
    sd.k_trans = d.k_trans + p.b_begin;
    sd.init_mode = d.init_mode + p.b_begin;
    sd.x0 = d.x0 ? d.x0 + p.b_begin * QLN_NX : nullptr;
    sd.xf = d.xf ? d.xf + p.b_begin * QLN_NX : nullptr;
    sd.cost = d.cost ? d.cost + p.cost_begin : nullptr;
    sd.cost_batch = p.cost_batch;
    return sd;
}

int qln_multi_plan(const qln_batch_desc* d, int n, qln_shard_plan* plan, int64_t* c_off, int64_t* c_total) {
    if (!d || !plan) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_plan: null argument");
    if (n < 1) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_plan: n_devices must be >= 1");
    if (d->B < n) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_plan: fewer problems than devices");
    if (!d->k_trans || !d->init_mode) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_plan: null descriptor array");
    if (d->cost_batch != 1 && d->cost_batch != d->B) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_plan: cost_batch must be 1 or B");
    int64_t displ = 0;
    std::vector<int64_t> local;
    for (int r = 0; r < n; ++r) {
        qln_shard_plan& p = plan[r];
        p = qln_shard_plan{};
        QM_OK(qln_shard_range(d->B, r, n, &p.b_begin, &p.b_end));
        const bool per_problem = d->cost_batch == d->B && d->B > 1;
        p.cost_begin = per_problem ? p.b_begin * (int64_t)d->N * QLN_COST_STRIDE : 0;
        p.cost_batch = per_problem ? (int32_t)(p.b_end - p.b_begin) : 1;
        const qln_batch_desc sd = shard_desc(*d, p);
        qln_dims dims{};
        local.assign((size_t)sd.B, 0);
        QM_OK(qln_layout(&sd, &dims, local.data(), nullptr));  // the offsets the shard's own handle will use
        p.z_begin = p.b_begin * dims.z_stride;
        p.c_displ = displ;
        p.z_total = dims.z_total, p.c_total = dims.c_total, p.j_total = dims.j_total;
        if (c_off)
            for (int64_t b = 0; b < sd.B; ++b) c_off[p.b_begin + b] = displ + local[(size_t)b];
        displ += dims.c_total;
    }
    if (c_total) *c_total = displ;
    return QLN_OK;
}

static int multi_create_impl(const qln_batch_desc* d, int n, const int* devices, bool one_device, qln_multi** out) {
    if (!d || !out) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: null argument");
    *out = nullptr;
    if (n < 1) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: n_devices must be >= 1");
    if (d->B < n) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: fewer problems than devices");
    if (!d->k_trans || !d->init_mode || !d->x0 || !d->xf) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: null descriptor array");
    if (d->cost_batch != 1 && d->cost_batch != d->B) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: cost_batch must be 1 or B");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(QLN_ERR_NO_DEVICE, "qln_multi_create: no HIP device visible (this library has no CPU fallback)");
    std::vector<int> devs((size_t)n);
    for (int r = 0; r < n; ++r) {
        devs[(size_t)r] = devices ? devices[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= ndev) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: bad device ordinal");
        for (int q = 0; q < r && !one_device; ++q)
            if (devs[(size_t)q] == devs[(size_t)r]) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: a device is listed twice");
    }
    qln_multi* m = new (std::nothrow) qln_multi();
    if (!m) return fail(QLN_ERR_HIP, "qln_multi_create: out of host memory");
    m->B = d->B;
    m->one_device = one_device;
    m->shards.resize((size_t)n);
    m->c_off.resize((size_t)d->B);
    auto bail = [&](int code) {
        const std::string keep = qln_last_error();  // qln_multi_destroy may overwrite it
        qln_multi_destroy(m);
        return fail(code, keep);
    };
    std::vector<qln_shard_plan> plan((size_t)n);
    if (int rc = qln_multi_plan(d, n, plan.data(), m->c_off.data(), &m->c_total)) return bail(rc);
    for (int r = 0; r < n; ++r) {
        auto& s = m->shards[(size_t)r];
        const qln_shard_plan& p = plan[(size_t)r];
        s.device = devs[(size_t)r];
        s.lo = p.b_begin, s.hi = p.b_end;
        s.c_displ = p.c_displ;
        s.z_begin = p.z_begin;
        const qln_batch_desc sd = shard_desc(*d, p);
        if (int rc = qln_create(&sd, s.device, &s.h)) return bail(rc);
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(QLN_ERR_HIP, "qln_multi_create: stream creation failed"));
        if (int rc = qln_set_stream(s.h, s.stream)) return bail(rc);
        if (int rc = qln_get_dims(s.h, &s.dims)) return bail(rc);
        if (s.dims.z_total != p.z_total || s.dims.c_total != p.c_total || s.dims.j_total != p.j_total)
            return bail(fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create: the shard handle's layout differs from the plan"));
        struct {
            double** p;
            int64_t n;
        } bufs[] = {{&s.Z, s.dims.z_total}, {&s.c, s.dims.c_total}, {&s.f, s.dims.B}, {&s.viol, s.dims.B}};
        for (auto& b : bufs) {
            if (hipMalloc(reinterpret_cast<void**>(b.p), (size_t)std::max<int64_t>(b.n, 1) * 8) != hipSuccess ||
                hipMemsetAsync(*b.p, 0, (size_t)std::max<int64_t>(b.n, 1) * 8, s.stream) != hipSuccess)
                return bail(fail(QLN_ERR_HIP, "qln_multi_create: device buffer allocation failed"));
        }
        m->z_stride = s.dims.z_stride;
    }
    m->comms.assign((size_t)n, nullptr);
    if (!one_device) {
        if (ncclResult_t r = ncclCommInitAll(m->comms.data(), n, devs.data()); r != ncclSuccess) {
            fail(QLN_ERR_COMM, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
            return bail(QLN_ERR_COMM);
        }
    }
    *out = m;
    return QLN_OK;
}

int qln_multi_create(const qln_batch_desc* d, int n, const int* devices, qln_multi** out) {
    return multi_create_impl(d, n, devices, false, out);
}

int qln_multi_create_on_one_device(const qln_batch_desc* d, int n_shards, int device, qln_multi** out) {
    if (n_shards < 1) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_create_on_one_device: n_shards must be >= 1");
    const std::vector<int> devs((size_t)n_shards, device);
    return multi_create_impl(d, n_shards, devs.data(), true, out);
}

int qln_multi_num_devices(const qln_multi* m, int* n) {
    if (!m || !n) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_num_devices: null argument");
    *n = (int)m->shards.size();
    return QLN_OK;
}

int qln_multi_shard(const qln_multi* m, int r, int* device, int64_t* b_begin, int64_t* b_end, qln_handle** handle, double** Z,
                    double** c, double** vals, double** f, double** viol) {
    if (!m || r < 0 || r >= (int)m->shards.size()) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_shard: bad shard index");
    const auto& s = m->shards[(size_t)r];
    if (device) *device = s.device;
    if (b_begin) *b_begin = s.lo;
    if (b_end) *b_end = s.hi;
    if (handle) *handle = s.h;
    if (Z) *Z = s.Z;
    if (c) *c = s.c;
    if (vals) *vals = s.vals;
    if (f) *f = s.f;
    if (viol) *viol = s.viol;
    return QLN_OK;
}

int qln_multi_get_offsets(const qln_multi* m, int64_t* c_off, int64_t* c_total) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    if (c_off) std::copy(m->c_off.begin(), m->c_off.end(), c_off);
    if (c_total) *c_total = m->c_total;
    return QLN_OK;
}

int qln_multi_set_Z(qln_multi* m, const double* Z_host) {
    if (!m || !Z_host) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_set_Z: null argument");
    for (auto& s : m->shards) {
        QM_HIP(hipSetDevice(s.device));
        QM_HIP(hipMemcpyAsync(s.Z, Z_host + s.z_begin, (size_t)s.dims.z_total * 8, hipMemcpyHostToDevice, s.stream));
    }
    // the copies read the caller's (pageable) buffer: it must not be freed or reused before they are done, and a C or
    // Julia caller cannot know when that is -- so the call waits (a one-off upload, never on a timed path)
    for (auto& s : m->shards) {
        QM_HIP(hipSetDevice(s.device));
        QM_HIP(hipStreamSynchronize(s.stream));
    }
    return QLN_OK;
}

int qln_multi_initial_guess(qln_multi* m) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) QM_OK(qln_initial_guess(s.h, s.Z));
    return QLN_OK;
}

int qln_multi_set_lqr_cost(qln_multi* m, const double* Q, const double* R, const double* Qf, double dt, int per_problem) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) QM_OK(qln_set_lqr_cost(s.h, Q, R, Qf, dt, per_problem));
    return QLN_OK;
}

int qln_multi_alloc_vals(qln_multi* m, int placed) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) {
        if (s.vals) continue;
        QM_HIP(hipSetDevice(s.device));
        if (placed) {
            QM_HIP(hipStreamSynchronize(s.stream));  // Z is in place before launches are timed on it
            QM_OK(qln_vals_alloc_placed(s.h, s.Z, nullptr, &s.vals, nullptr));
            s.vals_placed = true;
        } else {
            QM_HIP(hipMalloc(reinterpret_cast<void**>(&s.vals), (size_t)std::max<int64_t>(s.dims.j_total, 1) * 8));
            QM_HIP(hipMemsetAsync(s.vals, 0, (size_t)std::max<int64_t>(s.dims.j_total, 1) * 8, s.stream));
        }
        QM_OK(qln_jacobian_init_constants(s.h, s.vals));
    }
    return QLN_OK;
}

int qln_multi_eval_constraint_and_jacobian(qln_multi* m, int with_jacobian, uint32_t flags) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) {
        if (with_jacobian) {
            if (!s.vals) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_eval_constraint_and_jacobian: call qln_multi_alloc_vals first");
            QM_OK(qln_eval_constraint_and_jacobian(s.h, s.Z, s.c, s.vals, flags));
        } else {
            QM_OK(qln_eval_constraint(s.h, s.Z, s.c));
        }
    }
    return QLN_OK;
}

int qln_multi_eval_objective(qln_multi* m) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) QM_OK(qln_eval_objective(s.h, s.Z, s.f));
    return QLN_OK;
}

int qln_multi_constraint_violation(qln_multi* m) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) QM_OK(qln_constraint_violation(s.h, s.c, s.viol));
    return QLN_OK;
}

int qln_multi_solve(qln_multi* m, const qln_solve_options* opt) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) {
        if (!s.sinfo) {
            QM_HIP(hipSetDevice(s.device));
            QM_HIP(hipMalloc(reinterpret_cast<void**>(&s.sinfo), (size_t)s.dims.B * QLN_SOLVE_INFO_STRIDE * sizeof(double)));
        }
        QM_OK(qln_solve(s.h, s.Z, opt, s.sinfo));
    }
    return QLN_OK;
}

int qln_multi_solve_info(qln_multi* m, double* info_host) {
    if (!m || !info_host) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_solve_info: null argument");
    for (auto& s : m->shards) {
        if (!s.sinfo) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_solve_info: qln_multi_solve has not run");
        QM_HIP(hipSetDevice(s.device));
        QM_HIP(hipStreamSynchronize(s.stream));
        QM_HIP(hipMemcpy(info_host + s.lo * QLN_SOLVE_INFO_STRIDE, s.sinfo, (size_t)s.dims.B * QLN_SOLVE_INFO_STRIDE * sizeof(double),
                         hipMemcpyDeviceToHost));
    }
    return QLN_OK;
}

int qln_multi_synchronize(qln_multi* m) {
    if (!m) return fail(QLN_ERR_INVALID_ARGUMENT, "null handle");
    for (auto& s : m->shards) {
        QM_HIP(hipSetDevice(s.device));
        QM_HIP(hipStreamSynchronize(s.stream));
    }
    return QLN_OK;
}

int qln_multi_gather(qln_multi* m, uint32_t what, int root) {
    if (!m || root < 0 || root >= (int)m->shards.size()) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_gather: bad argument");
    if (m->root >= 0 && m->root != root) return fail(QLN_ERR_UNSUPPORTED, "qln_multi_gather: the root shard is fixed by the first gather");
    auto& rs = m->shards[(size_t)root];
    QM_HIP(hipSetDevice(rs.device));
    m->root = root;
    if ((what & QLN_GATHER_F) && !m->g_f) QM_HIP(hipMalloc(reinterpret_cast<void**>(&m->g_f), (size_t)m->B * 8));
    if ((what & QLN_GATHER_VIOL) && !m->g_viol) QM_HIP(hipMalloc(reinterpret_cast<void**>(&m->g_viol), (size_t)m->B * 8));
    if ((what & QLN_GATHER_C) && !m->g_c) QM_HIP(hipMalloc(reinterpret_cast<void**>(&m->g_c), (size_t)std::max<int64_t>(m->c_total, 1) * 8));
    if (m->one_device) {
        // n shards on ONE device (RCCL admits one rank per device): the exchange of every send / receive pair is a device
        // copy on the root's stream, ordered behind the sending shard's stream by an event -- the ordering the pair has
        // over RCCL (send after the shard's evaluations, receive on the root's stream), the same offsets and counts
        for (auto& s : m->shards) {
            const size_t nb = (size_t)(s.hi - s.lo);
            hipEvent_t ev = nullptr;
            QM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipError_t e = hipEventRecord(ev, s.stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(rs.stream, ev, 0);
            if (e == hipSuccess && (what & QLN_GATHER_F)) e = hipMemcpyAsync(m->g_f + s.lo, s.f, nb * 8, hipMemcpyDeviceToDevice, rs.stream);
            if (e == hipSuccess && (what & QLN_GATHER_VIOL)) e = hipMemcpyAsync(m->g_viol + s.lo, s.viol, nb * 8, hipMemcpyDeviceToDevice, rs.stream);
            if (e == hipSuccess && (what & QLN_GATHER_C) && s.dims.c_total > 0)
                e = hipMemcpyAsync(m->g_c + s.c_displ, s.c, (size_t)s.dims.c_total * 8, hipMemcpyDeviceToDevice, rs.stream);
            (void)hipEventDestroy(ev);  // released once the recorded work has completed
            if (e != hipSuccess) return fail(QLN_ERR_HIP, std::string("qln_multi_gather (one device): ") + hipGetErrorString(e));
        }
        m->gathered |= what;
        return QLN_OK;
    }
    // every shard sends on its own stream (ordered after its evaluations); the root receives on its stream
    ncclResult_t res = ncclSuccess;
    QM_NCCL(ncclGroupStart());
    const int n = (int)m->shards.size();
    for (int r = 0; r < n && res == ncclSuccess; ++r) {
        auto& s = m->shards[(size_t)r];
        const size_t nb = (size_t)(s.hi - s.lo);
        if ((what & QLN_GATHER_F) && res == ncclSuccess) res = ncclSend(s.f, nb, ncclDouble, root, m->comms[(size_t)r], s.stream);
        if ((what & QLN_GATHER_VIOL) && res == ncclSuccess) res = ncclSend(s.viol, nb, ncclDouble, root, m->comms[(size_t)r], s.stream);
        if ((what & QLN_GATHER_C) && res == ncclSuccess) res = ncclSend(s.c, (size_t)s.dims.c_total, ncclDouble, root, m->comms[(size_t)r], s.stream);
        // the matching receives, in the same order per peer
        if ((what & QLN_GATHER_F) && res == ncclSuccess) res = ncclRecv(m->g_f + s.lo, nb, ncclDouble, r, m->comms[(size_t)root], rs.stream);
        if ((what & QLN_GATHER_VIOL) && res == ncclSuccess) res = ncclRecv(m->g_viol + s.lo, nb, ncclDouble, r, m->comms[(size_t)root], rs.stream);
        if ((what & QLN_GATHER_C) && res == ncclSuccess) res = ncclRecv(m->g_c + s.c_displ, (size_t)s.dims.c_total, ncclDouble, r, m->comms[(size_t)root], rs.stream);
    }
    const ncclResult_t end = ncclGroupEnd();
    if (res != ncclSuccess) return fail(QLN_ERR_COMM, std::string("qln_multi_gather: ncclSend/ncclRecv: ") + ncclGetErrorString(res));
    if (end != ncclSuccess) return fail(QLN_ERR_COMM, std::string("qln_multi_gather: ncclGroupEnd: ") + ncclGetErrorString(end));
    m->gathered |= what;
    return QLN_OK;
}

int qln_multi_gathered_to_host(qln_multi* m, double* f, double* viol, double* c) {
    if (!m || m->root < 0) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_gathered_to_host: nothing has been gathered");
    if ((f && !(m->gathered & QLN_GATHER_F)) || (viol && !(m->gathered & QLN_GATHER_VIOL)) || (c && !(m->gathered & QLN_GATHER_C)))
        return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_gathered_to_host: that array has not been gathered");
    QM_OK(qln_multi_synchronize(m));
    QM_HIP(hipSetDevice(m->shards[(size_t)m->root].device));
    if (f) QM_HIP(hipMemcpy(f, m->g_f, (size_t)m->B * 8, hipMemcpyDeviceToHost));
    if (viol) QM_HIP(hipMemcpy(viol, m->g_viol, (size_t)m->B * 8, hipMemcpyDeviceToHost));
    if (c) QM_HIP(hipMemcpy(c, m->g_c, (size_t)m->c_total * 8, hipMemcpyDeviceToHost));
    return QLN_OK;
}

int qln_multi_time_constraint_and_jacobian(qln_multi* m, int32_t warmup, int32_t iters, float* ms_per_device, double* wall_ms) {
    if (!m || !ms_per_device || iters < 1 || warmup < 0) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_time_constraint_and_jacobian: bad argument");
    const size_t n = m->shards.size();
    for (auto& s : m->shards)
        if (!s.vals) return fail(QLN_ERR_INVALID_ARGUMENT, "qln_multi_time_constraint_and_jacobian: call qln_multi_alloc_vals first");
    std::vector<hipEvent_t> e0(n, nullptr), e1(n, nullptr);
    int rc = QLN_OK;
    for (size_t r = 0; r < n && rc == QLN_OK; ++r)
        if (hipSetDevice(m->shards[r].device) != hipSuccess || hipEventCreate(&e0[r]) != hipSuccess || hipEventCreate(&e1[r]) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "hipEventCreate failed");
    // One host thread per device.  Each binds its device, waits at the gate, then issues its shard's launches and waits
    // for its device: every device's queue starts within microseconds of the others', whatever `iters` is.  The error
    // slot of the evaluator is thread-local: a failing thread hands its code and message back.
    std::vector<int> trc(n, QLN_OK);
    std::vector<std::string> tmsg(n);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    auto worker = [&](size_t r, int times, bool timed) {
        auto& s = m->shards[r];
        int my = QLN_OK;
        if (hipSetDevice(s.device) != hipSuccess) my = fail(QLN_ERR_HIP, "hipSetDevice failed");
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        if (timed && my == QLN_OK && hipEventRecord(e0[r], s.stream) != hipSuccess) my = fail(QLN_ERR_HIP, "hipEventRecord failed");
        for (int i = 0; i < times && my == QLN_OK; ++i) my = qln_eval_constraint_and_jacobian(s.h, s.Z, s.c, s.vals, 0);
        if (timed && my == QLN_OK && hipEventRecord(e1[r], s.stream) != hipSuccess) my = fail(QLN_ERR_HIP, "hipEventRecord failed");
        if (my == QLN_OK && hipStreamSynchronize(s.stream) != hipSuccess) my = fail(QLN_ERR_HIP, "hipStreamSynchronize failed");
        trc[r] = my;
        if (my != QLN_OK) tmsg[r] = qln_last_error();
    };
    auto run_all = [&](int times, bool timed, double* elapsed_ms) {
        ready = 0;
        go = false;
        std::vector<std::thread> th;
        th.reserve(n);
        // nothing may throw across the C ABI: a thread that cannot be started (std::system_error) releases the ones that
        // were, waits for them and becomes an error code
        try {
            for (size_t r = 0; r < n; ++r) th.emplace_back(worker, r, times, timed);
        } catch (const std::exception& e) {
            go.store(true, std::memory_order_release);
            for (auto& t : th) t.join();
            return fail(QLN_ERR_HIP, std::string("qln_multi_time_constraint_and_jacobian: could not start an issue thread: ") + e.what());
        }
        while (ready.load() < (int)n) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true, std::memory_order_release);
        for (auto& t : th) t.join();
        if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (size_t r = 0; r < n; ++r)
            if (trc[r] != QLN_OK) return fail(trc[r], tmsg[r]);
        return (int)QLN_OK;
    };
    if (rc == QLN_OK) rc = qln_multi_synchronize(m);
    if (rc == QLN_OK && warmup) rc = run_all(warmup, false, nullptr);
    if (rc == QLN_OK) rc = run_all(iters, true, wall_ms);
    for (size_t r = 0; r < n && rc == QLN_OK; ++r)
        if (hipSetDevice(m->shards[r].device) != hipSuccess || hipEventElapsedTime(&ms_per_device[r], e0[r], e1[r]) != hipSuccess)
            rc = fail(QLN_ERR_HIP, "hipEventElapsedTime failed");
    for (size_t r = 0; r < n; ++r) {
        if (e0[r]) (void)hipEventDestroy(e0[r]);
        if (e1[r]) (void)hipEventDestroy(e1[r]);
    }
    return rc;
}

}  // extern "C"
