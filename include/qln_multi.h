/*
 * qln_multi.h -- multi-GPU layer of the batched landing-NLP evaluator, behind the same C ABI (libqln_multi.so,
 * on top of libqln_hip.so and RCCL).
 *
 * The reference has no multi-device code at all ("NCCL/MPI/collective call sites: none"); this is the
 * sharding BASELINE.json's north_star asks for: "batches shard embarrassingly across the 8 GPUs of one node with an
 * RCCL gather over xGMI only at the end".  Problems are independent, so a batch is cut into contiguous ranges of
 * problems, one per GPU; every GPU evaluates its range with the single-GPU entry points of qln_evaluator.h; there is
 * NO data-path collective.  The one exchange is the end-of-job gather of per-problem results to a root GPU.
 *
 * Two ways to drive it, same partitioning and same gather:
 *
 *   qln_multi_*   ONE process, n devices (ncclCommInitAll): what a single Julia session with 8 GPUs does.  The library
 *                 owns one evaluator handle, one stream and one set of buffers (Z, c, f, viol; vals on request) per
 *                 device; launches go to all devices before anything is waited for.
 *   qln_comm_*    one process PER GPU (ncclCommInitRank): what a launcher-started job does (bench.py under
 *                 torch.distributed.run; Julia's Distributed workers).  The host passes the 128-byte RCCL id from rank
 *                 0 to the others by whatever channel it has; each rank then owns an ordinary qln_handle for its shard.
 *
 * Shards are ragged in general (a shard's constraint vector holds sum_b round_up(18N - k_trans(b) + 16, align)
 * doubles), so the gather is n (recv on the root, send on every rank) pairs inside ncclGroupStart/End -- what
 * ncclGather does internally -- with per-rank counts.  Jacobian values are not gathered: at BASELINE.json configs[4]
 * they are 6.2 GB per GPU, ~40 ms into one GPU's seven xGMI links against a ~1.1 ms evaluation; they
 * stay on the GPU that produced them for a consumer there (qln_solver.h).
 *
 * Every entry point returns QLN_OK or a negative QLN_ERR_* code (QLN_ERR_COMM for RCCL failures); the message is in
 * qln_last_error() of qln_evaluator.h (same thread-local slot).
 */
#ifndef QLN_MULTI_H
#define QLN_MULTI_H

#include "qln_evaluator.h"

#ifdef __cplusplus
extern "C" {
#endif

#define QLN_ERR_COMM (-5)

/* what to move in a gather (bit mask) */
#define QLN_GATHER_F 1u    /* objective, one double per problem                     (src/costs.jl:6-16)        */
#define QLN_GATHER_VIOL 2u /* constraint violation, one double per problem          (qln_constraint_violation) */
#define QLN_GATHER_C 4u    /* the full constraint vectors                            (src/constraints.jl:145-158) */

/* Contiguous, balanced range [begin, end) of `n_problems` owned by `rank` of `world` (the first n_problems % world
 * ranks get one problem more).  The one partitioning rule used everywhere (Python: distributed.shard_range). */
int qln_shard_range(int64_t n_problems, int rank, int world, int64_t* begin, int64_t* end);

/* ------------------------------------------------------------------ one process per GPU */
typedef struct qln_comm qln_comm;

#define QLN_COMM_ID_BYTES 128
/* rank 0 calls this and hands the 128 bytes to every other rank (ncclGetUniqueId) */
int qln_comm_get_unique_id(void* id /*[QLN_COMM_ID_BYTES]*/);
/* collective over all `world` ranks: every rank calls it with the same id, its own rank and its HIP device */
int qln_comm_init_rank(const void* id, int rank, int world, int device, qln_comm** out);
int qln_comm_destroy(qln_comm* comm);
int qln_comm_rank(const qln_comm* comm, int* rank, int* world);
/* Every rank's element count on every rank (host array, [world]); collective, waits for the result.  Shards are
 * ragged, so the root sizes its receive buffer from this before the gather. */
int qln_comm_exchange_counts(qln_comm* comm, int64_t count, int64_t* counts /*[world]*/);
/* Gather `count` doubles from every rank's device buffer `send` into `recv` on the root (device pointer, root only;
 * rank r's block starts at the sum of the counts of the ranks before it).  `counts`: [world], as returned by
 * qln_comm_exchange_counts -- read on the root only (may be NULL elsewhere).  Stream-ordered on `hip_stream` (a
 * hipStream_t of the rank's device; NULL = default stream); returns when the transfers are enqueued. */
int qln_comm_gather(qln_comm* comm, const double* send, int64_t count, double* recv, const int64_t* counts, int root,
                    void* hip_stream);
/* value <- max over all ranks (the timing reduction of bench.py); waits for the result */
int qln_comm_max(qln_comm* comm, double* value);
int qln_comm_barrier(qln_comm* comm);

/* ------------------------------------------------------------------ one process, n devices */
typedef struct qln_multi qln_multi;

/* Where everything of shard r lies -- the host bookkeeping of qln_multi_create, which needs no device: contiguous balanced
 * ranges (qln_shard_range), the cuts of the batch's host arrays, and the shard's place in the gathered constraint vector
 * (shards are ragged: a shard's constraint buffer holds sum_b round_up(18N - k_trans(b) + 16, align) doubles).
 * qln_multi_create / qln_multi_set_Z / qln_multi_gather work from exactly this plan. */
typedef struct qln_shard_plan {
    int64_t b_begin, b_end;  /* problems [b_begin, b_end) of the batch                                              */
    int64_t z_begin;         /* doubles: the shard's first row in the batch's host Z / grad = b_begin * z_stride      */
    int64_t cost_begin;      /* doubles into desc->cost: b_begin * N * 41 for a per-problem table, 0 for a shared one */
    int32_t cost_batch;      /* 1 (shared table) or the shard's problem count                                       */
    int32_t reserved;
    int64_t c_displ;         /* where the shard's constraint vector starts in the gathered one                      */
    int64_t z_total, c_total, j_total; /* buffer sizes of the shard's evaluator handle (doubles)                     */
} qln_shard_plan;
/* plan: [n_devices]; c_off: [B] or NULL, global problem b's offset in the GATHERED constraint vector; c_total (may be
 * NULL): its length.  Host arithmetic only: works without a GPU. */
int qln_multi_plan(const qln_batch_desc* desc, int n_devices, qln_shard_plan* plan, int64_t* c_off, int64_t* c_total);

/* Shards the batch described by `desc` (all of it in host memory, as for qln_create) over `n_devices` HIP devices
 * (`devices` = their ordinals, NULL = 0 .. n_devices-1), creates one evaluator handle + stream + buffer set per
 * device and the RCCL clique.  A per-problem cost table (cost_batch == B) is sharded like the problems. */
int qln_multi_create(const qln_batch_desc* desc, int n_devices, const int* devices, qln_multi** out);
/* REHEARSAL of the n > 1 paths on a box with one GPU: the same sharding, handles, streams, issue threads, buffers and gather
 * offsets with all `n_shards` shards on ONE device.  RCCL admits one rank per device, so no clique is created and the gather's
 * send / receive pairs are device copies on the root's stream ordered behind the sending shard's stream (what the pairs are
 * over RCCL).  Everything else of this header behaves as after qln_multi_create.  Not a way to run faster. */
int qln_multi_create_on_one_device(const qln_batch_desc* desc, int n_shards, int device, qln_multi** out);
int qln_multi_destroy(qln_multi* m);
int qln_multi_num_devices(const qln_multi* m, int* n_devices);
/* shard r: its device, its range of the global problem index, its evaluator handle (owned by m; every single-GPU entry
 * point of qln_evaluator.h may be used on it) and its device buffers (vals is NULL until qln_multi_alloc_vals).
 * Any out pointer may be NULL. */
int qln_multi_shard(const qln_multi* m, int r, int* device, int64_t* b_begin, int64_t* b_end, qln_handle** handle,
                    double** Z, double** c, double** vals, double** f, double** viol);
/* global problem b's offset in the GATHERED constraint vector (shard-local c_off + the shard's displacement) */
int qln_multi_get_offsets(const qln_multi* m, int64_t* c_off /*[B]*/, int64_t* c_total);
/* Z of the whole batch from host memory ([B][z_stride] doubles as for qln_eval_*_host), cut and copied to the shards.
 * Returns when the copies are complete: Z_host may be freed or reused as soon as the call returns. */
int qln_multi_set_Z(qln_multi* m, const double* Z_host);
/* or built where it is used: the notebook's initial guess on every device (qln_initial_guess) */
int qln_multi_initial_guess(qln_multi* m);
int qln_multi_set_lqr_cost(qln_multi* m, const double* Qdiag, const double* Rdiag, const double* Qfdiag, double dt,
                           int per_problem);
/* Jacobian buffers: one per device, j_total(shard) doubles; placed != 0 uses qln_vals_alloc_placed */
int qln_multi_alloc_vals(qln_multi* m, int placed);
/* Evaluation on every shard; launches are issued to all devices before anything is waited for (asynchronous; order
 * per device = the order of the calls).  with_jacobian needs qln_multi_alloc_vals. */
int qln_multi_eval_constraint_and_jacobian(qln_multi* m, int with_jacobian, uint32_t flags);
int qln_multi_eval_objective(qln_multi* m);         /* -> every shard's f    */
int qln_multi_constraint_violation(qln_multi* m);   /* -> every shard's viol (from its c) */
/* qln_solve on every shard, in place on the shard's Z (qln_multi_set_Z / qln_multi_initial_guess); launched on all
 * devices before anything is waited for.  Afterwards qln_multi_eval_objective + qln_multi_eval_constraint_and_jacobian(0) +
 * qln_multi_constraint_violation give the evaluator's verdict on the solutions, and qln_multi_gather brings it to the root:
 * the end-of-job gather then carries final results.  opt: NULL = defaults.  info: per-shard device buffers are owned by m;
 * qln_multi_solve_info copies them to the host ([B][QLN_SOLVE_INFO_STRIDE], global problem order). */
int qln_multi_solve(qln_multi* m, const qln_solve_options* opt);
int qln_multi_solve_info(qln_multi* m, double* info_host);
int qln_multi_synchronize(qln_multi* m);
/* The end-of-job exchange: per-problem results of every shard to device `root_shard`'s gather buffers (allocated on
 * first use), over RCCL.  `what` = QLN_GATHER_* bits.  Stream-ordered after the evaluations. */
int qln_multi_gather(qln_multi* m, uint32_t what, int root_shard);
/* the gathered arrays, copied to host memory after waiting for the gather: f, viol: [B]; c: [c_total] (see
 * qln_multi_get_offsets).  Pointers may be NULL. */
int qln_multi_gathered_to_host(qln_multi* m, double* f, double* viol, double* c);
/* Measurement helper for bench.py: `iters` back-to-back fused launches on every device, issued by ONE HOST THREAD PER
 * DEVICE that are released together, so that every device's queue starts within microseconds of the others (issuing all
 * of device 0's launches, then all of device 1's ... from one thread would start device r about r * iters launch
 * overheads late and book host issue order as lost scaling).  ms_per_device[r]: HIP events on shard r's stream around
 * its `iters` launches (events are created before, destroyed after the timed region).  wall_ms (may be NULL): host clock
 * from the release of the threads until every device is idle.  Returns after all devices are idle. */
int qln_multi_time_constraint_and_jacobian(qln_multi* m, int32_t warmup, int32_t iters, float* ms_per_device, double* wall_ms);

#ifdef __cplusplus
}
#endif
#endif
