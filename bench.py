#!/usr/bin/env python3
"""bench.py -- knot-point constraint+Jacobian evals/sec of the fused HIP hot path.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A step = one launch of eval_c! + jac_c! over every knot of every problem of the rank's shard, with
inputs and outputs resident in HBM.  Default workload = BASELINE.json configs[2] (0-based: B=65536, N=40,
k_trans=14, FP64 -- "config 3" of BASELINE.md): the HBM-bound regime the metric's roofline half is quoted on;
configs[1] (B=1024, launch-latency regime) is timed beside it and reported under "other".  The --workload names
config2/config3/config4 follow BASELINE.md's 1-based table, i.e. configs[1]/[2]/[3].

ONE measurement for every N.  Whatever N is and however the N GPUs are driven, the line is produced the same way:
  * every GPU owns a full-size shard (weak scaling), generated ON that GPU by the same device-side generator with the
    shard's index as the seed (the ragged workload: host generator, uploaded); no data-path collective;
  * `value` = knot evals of all GPUs / the wall clock of the K-launch region, which is bracketed by a barrier +
    device synchronisation on both sides and contains NOTHING but the K launches of every GPU (max over ranks);
  * the end-of-job tail -- objective + constraint violation of every problem, then the single gather of those
    per-problem results to GPU 0 (north_star: "an RCCL gather over xGMI only at the end") -- runs right after the timed
    region and is reported beside it as `gather_ms`, for N = 1 too (where the gather degenerates to a device copy);
    the gather of the full constraint vectors (379 MB per rank) is timed once more as `gather_c_ms`; Jacobian values
    stay resident on the GPU that produced them;
  * `roofline` is computed from the SLOWEST GPU's HIP-event launch average (per-GPU averages are listed beside it).
Two drivers produce that line: 'ranks' = one process per GPU (the driver's torch.distributed.run line; also plain
`python bench.py` at N = 1), RCCL through qln_comm_* -- torch.distributed (gloo) only carries RCCL's 128-byte id;
'single-process' = this process drives all N devices through qln_multi_* (one host thread per device issues its
launches).  `config.driver`, `config.workload_data` and `config.timing` say which produced the line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    "config2": dict(B=1024, N=40, k_trans=14, ragged=False, desc="B=1024 N=40 k_trans=14 init_mode=1 FP64"),
    "config3": dict(B=65536, N=40, k_trans=14, ragged=False, desc="B=65536 N=40 k_trans=14 init_mode=1 FP64"),
    "config4": dict(B=65536, N=80, k_trans=0, ragged=True, desc="B=65536 N=80 per-problem k_trans~U{2..79}, init_mode~U{1,2} FP64"),
}


def algorithmic_bytes(N, k_trans):
    """read Z + write c + write state-dependent Jacobian values, per problem."""
    k_trans = np.asarray(k_trans, dtype=np.int64)
    return 8 * (20 * N - 5) + 8 * (18 * N - k_trans + 16) + 8 * (300 * (N - 1) + N)


NNZ_CONTACT, NNZ_FLIGHT, NNZ_JUMP = 71, 57, 56  # structural non-zeros of a 15x20 step block


def strict_bytes(N, k_trans):
    """SURVEY.md 8d "strict (structural-nnz) variant": as algorithmic_bytes, but only the structurally non-zero
    entries of every step block count: 71 in contact modes 1/2, 57 in mode 3, and 56 at the jump knot k_trans-1
    (71 minus the 15 entries in the rows the jump mask zeroes: 5, 7, 11-15)."""
    kt = np.asarray(k_trans, dtype=np.int64)
    n_contact = np.clip(kt - 2, 0, N - 1)                       # knots k < k_trans-1
    n_jump = ((kt - 1 >= 1) & (kt - 1 <= N - 1)).astype(np.int64)
    n_flight = np.clip(N - kt, 0, N - 1)                        # knots k_trans <= k <= N-1
    nnz = NNZ_CONTACT * n_contact + NNZ_JUMP * n_jump + NNZ_FLIGHT * n_flight
    return 8 * (20 * N - 5) + 8 * (18 * N - kt + 16) + 8 * (nnz + N)


def build(workload, seed, device, placement_trials=1, jac_format="dense_blocks", host_data=False):
    import torch
    from quadruped_landing_amd import HybridNLP, PlanarQuadruped, problem_gen as PG

    w = WORKLOADS[workload]
    ragged_off = None
    if w["ragged"] and not host_data:
        # config 4's descriptors drawn on the device with numpy's own algorithm and stream positions; usable unless numpy
        # would have rejected a draw for this seed (probability 3e-4 at this size), which shifts the stream behind it
        kt, im, ragged_off = PG.ragged_descriptors(seed, w["B"], w["N"], device=device)
    if (w["ragged"] and ragged_off is None) or host_data:
        # host generator (numpy), uploaded.  Per-problem cost tables (config 4: 1.7 GB) are built on the device, not uploaded.
        batch = PG.make_batch(w["B"], w["N"], w["k_trans"] or 14, 1, seed=seed, ragged=w["ragged"], build_obj=not w["ragged"])
        nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                        device=device, stream=torch.cuda.current_stream(), jac_format=jac_format)
        if batch.obj is None:
            nlp.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=True)
        Z = nlp.upload_Z(batch.Z)
        build.last_data = "host generator (problem_gen.make_batch), uploaded"
    else:
        # the whole synthetic workload is generated where it is used: drop states with numpy's own
        # PCG64 stream positions (x0 bit-identical to make_batch(seed)), the LQR cost records, the notebook's initial
        # guess and its N(0, 0.05^2) perturbation -- nothing but the descriptors is uploaded
        model = PlanarQuadruped()
        B, N = w["B"], w["N"]
        off = 0
        if not w["ragged"]:
            kt = np.full(B, w["k_trans"], dtype=np.int32)
            im = np.full(B, 1, dtype=np.int32)
        else:
            off = ragged_off  # the integers consumed the first B outputs of the stream
        xf = np.tile(PG.terminal_state(model), (B, 1))
        nlp = HybridNLP(model, None, im, kt, N, np.zeros((B, 15)), xf, device=device, stream=torch.cuda.current_stream(),
                        jac_format=jac_format)
        x0 = nlp.sample_drop_states(PG.drop_state_sampler(seed, model, stream_offset=off))
        nlp.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=w["ragged"])
        Z = nlp.perturb_point(nlp.initial_guess(), PG.drop_state_sampler(seed, model, stream_offset=off + 4 * B), sigma=0.05,
                              redraw_h=w["ragged"])
        batch = PG.LandingBatch(model, N, kt, im, x0, xf, None, None)  # Z stays on the device (cpu_baseline fetches it)
        build.last_data = ("generated on the device (" + ("qln_sample_bounded_integers / " if w["ragged"] else "") +
                           "qln_sample_drop_states / qln_set_lqr_cost / qln_initial_guess / qln_perturb_point)")
    c = nlp.new_c()
    # setup: the long-lived output buffer is allocated once; among `placement_trials` candidate allocations the one
    # whose physical placement sustains the best store bandwidth is kept (HybridNLP.new_vals_placed)
    vals, trial_ms = nlp.new_vals_placed(Z, c, trials=placement_trials, regions=placement_trials > 1)
    nlp.init_jacobian_constants(vals)  # constants are written once at setup
    build.last_trials = trial_ms
    build.last_placement = ("plain allocation" if placement_trials <= 1 else
                            "placed across two 32-GiB regions at setup (qln_vals_alloc_placed)" if len(trial_ms) == 1 else
                            f"fastest of {len(trial_ms)} candidate allocations ~32 GiB apart")
    return batch, nlp, Z, c, vals


def cpu_baseline(batch, nlp, Z_dev, budget_s=12.0):
    """Oracle (C restatement of the reference algorithm, 1 thread) on a bounded sample of the SAME
    workload.  Reported baseline, not the target."""
    from tests.helpers import oracle_model
    from oracle import oracle as O

    if batch.Z is None:  # device-generated workload: the oracle gets the very point the GPU evaluated
        batch.Z = Z_dev.cpu().numpy().reshape(batch.B, -1)[:, : nlp.n_nlp]

    def run(nb, nthreads):
        Zs = np.zeros((nb, nlp.z_stride))
        Zs[:, : nlp.n_nlp] = batch.Z[:nb]
        if batch.obj is None:  # device-built cost records: fetch the sample's back for the oracle
            obj = nlp.get_cost()
            obj = obj if obj.ndim == 2 else obj[:nb]
        else:
            obj = batch.obj if batch.obj.ndim == 2 else batch.obj[:nb]
        c_off = nlp.c_off[:nb] - nlp.c_off[0]
        c_total = int(c_off[-1] + nlp.dims.m_nlp_max + 16)
        # the oracle writes the dense-block value layout whatever the handle's jac_format is: its own offsets
        N, kt = batch.N, batch.k_trans[:nb].astype(np.int64)
        nnz = 300 * (N - 1) + N + 435 + 15 * (N - 1) + 3 * N - kt + 3
        stride = (nnz + 15) // 16 * 16
        j_off = np.concatenate([[0], np.cumsum(stride)[:-1]]).astype(np.int64)
        j_total = int(j_off[-1] + stride[-1])
        t0 = time.perf_counter()
        O.batch_eval(batch.N, oracle_model(batch.model), batch.k_trans[:nb], batch.init_mode[:nb], batch.x0[:nb],
                     batch.xf[:nb], obj, Zs.reshape(-1), nlp.z_stride, c_off, j_off, c_total, j_total,
                     True, True, False, False, nthreads)
        return time.perf_counter() - t0

    probe = min(256, batch.B)
    t = run(probe, 1)
    nb = int(min(batch.B, max(probe, probe * budget_s / max(t, 1e-6))))
    t1 = run(nb, 1)
    one = {"value": nb * batch.N / t1, "unit": "knot-evals/s", "cores": 1, "kind": "port",
           "sample": f"{nb} problems x N={batch.N} ({nb * batch.N} knot evals, {t1:.1f} s) of the same synthetic batch; "
                     "C restatement of the reference algorithm (oracle/qln_oracle.c, forward-mode duals), not Julia"}
    ncores = min(os.cpu_count() or 1, 16)
    nb_all = int(min(batch.B, nb * min(ncores, 4)))
    tall = run(nb_all, ncores)
    allc = {"value": nb_all * batch.N / tall, "unit": "knot-evals/s", "cores": ncores, "kind": "port",
            "sample": f"{nb_all} problems, OpenMP over problems, {tall:.1f} s"}
    return one, allc


def cpu_solve_baseline(batch, nlp, budget_s=25.0):
    """A CPU figure beside the solve numbers: the numpy prototype of the SAME method (augmented-Lagrangian iLQR; Riccati sweep
    per problem) on the C restatement's dynamics and Jacobians -- bench/solver_prototype.py, the development aid the kernel was
    designed with: one thread, Python loops over the knots, its own (first, exact-inner-solve) penalty schedule.  Not Ipopt (not
    installed; the reference's own Ipopt run of the notebook problem: 428 iterations, 594 s, "Restoration Failed",
    src/main.ipynb:707-727), not tuned: a reported baseline for scale, on a bounded sample of the same problems."""
    import contextlib
    import importlib.util
    import io

    spec = importlib.util.spec_from_file_location("solver_prototype", os.path.join(ROOT, "bench", "solver_prototype.py"))
    SP = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(SP)
    from quadruped_landing_amd.ref_traj import reference_trajectory

    obj = nlp.get_cost()
    x0, xf = nlp.boundary_states()
    done, solved, iters = 0, 0, []
    t0 = time.perf_counter()
    while done < batch.B and (done == 0 or (time.perf_counter() - t0) * (done + 1) / done < budget_s):
        b = done
        cost = obj if obj.ndim == 2 else obj[b]
        p = SP.Problem(batch.N, int(batch.k_trans[b]), int(batch.init_mode[b]), x0[b], xf[b], cost)
        _, Ur = reference_trajectory(batch.model, batch.N, batch.k_trans[b:b + 1], xf[b:b + 1], batch.init_mode[b:b + 1], 0.009)
        with contextlib.redirect_stdout(io.StringIO()):
            _, _, _, viol, it = SP.solve(p, Ur[0], verbose=False)
        done += 1
        solved += int(viol <= 1e-6)
        iters.append(it)
    dt = time.perf_counter() - t0
    return {"solved_problems_per_s": solved / dt, "cores": 1, "kind": "port", "problems": done, "solved_to_1e-6": solved,
            "wall_s": dt, "ilqr_iterations_median": float(np.median(iters)),
            "sample": f"the first {done} problems of the same batch, {dt:.1f} s; numpy prototype of the same AL-iLQR on the C "
                      "restatement's dynamics (bench/solver_prototype.py): 1 thread, Python loops, exact inner solves -- not Ipopt, not tuned"}


def kernel_build_id():
    """Hash of the kernel sources the loaded library was built from: measured HBM traffic (profiles/traffic.json) is
    only quoted for the build it was taken on."""
    import hashlib

    h = hashlib.sha256()
    for f in ("qln_kernels.hip", "qln_kernel_common.h", "qln_device.h"):
        h.update(open(os.path.join(ROOT, "quadruped_landing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """(bytes per launch, note): PMC-measured HBM traffic of the hot kernel for this workload, from the profile of
    THIS build (bench/profile_round.sh stamps profiles/traffic.json with kernel_build_id()), or (None, why not)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(tpath)).get(key)
    except Exception:
        e = None
    if not e:
        return None, f"no PMC profile of workload '{key}' in profiles/traffic.json (bench/profile_round.sh collects one)"
    if e.get("build_id") != kernel_build_id():
        return None, (f"profiles/traffic.json['{key}'] ({e.get('hbm_bytes_per_launch'):.4g} B/launch) was taken on kernel build "
                      f"{e.get('build_id', 'unstamped')}, this is {kernel_build_id()}: not quoted")
    return e.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, {e.get('profile', 'profiles/')}"


def stamped_profile_ms(key, placed):
    """The rocprofv3 kernel average of the newest tracked profile round of this workload (profiles/traffic.json, filed by
    bench/collect_profiles.py), or None: bench.py puts it beside its own live HIP-event average when the two differ by more
    than 5 % (another box, or a kernel that changed since the profile)."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key) or {}
    except Exception:
        return None, None
    return e.get("kernel_avg_ms_rocprof_placed" if placed else "kernel_avg_ms_rocprof_plain"), e.get("build_id")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other", action="store_true")
    ap.add_argument("--jac-format", default="dense_blocks", choices=["dense_blocks", "structural"],
                    help="layout of the step blocks in vals: the reference's dense 15x20 blocks (the unit SURVEY.md 8d "
                         "prices) or only their structurally non-zero entries (priced at the strict byte count)")
    ap.add_argument("--placement-trials", type=int, default=8,
                    help="1 = the Jacobian buffer is a plain allocation; > 1 = it is placed across two 32-GiB regions of device "
                         "memory at setup (HybridNLP.new_vals_placed), falling back to this many timed candidate allocations")
    ap.add_argument("--host-data", action="store_true",
                    help="generate the synthetic workload on the host with numpy and upload it (the default is the device-side "
                         "generator; the ragged workload falls back to the host for a seed whose integer draws numpy would reject)")
    ap.add_argument("--mode", default="auto", choices=["auto", "ranks", "single-process"],
                    help="how N GPUs are driven: 'ranks' = one process per GPU (a launcher set WORLD_SIZE/RANK/LOCAL_RANK: the "
                         "driver's torch.distributed.run line), RCCL through qln_comm_*; 'single-process' = this process "
                         "drives all N devices through qln_multi_* (include/qln_multi.h); auto = ranks under a launcher, "
                         "else single-process when --gpus > 1")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}")
    mode = args.mode
    if mode == "auto":
        mode = "ranks" if env_world is not None else ("single-process" if args.gpus > 1 else "ranks")
    if mode == "single-process" and env_world is not None and int(env_world) > 1:
        raise SystemExit("bench.py: --mode single-process under a multi-rank launcher")

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner to fd 1 when a
    # communicator is created), so everything that is not the result goes to stderr: fd 1 is pointed at stderr for the
    # duration of the run and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the evaluator has no CPU fallback")
    if mode == "single-process":
        out = run_single_process(args)
    else:
        out = run_ranks(args)
    if out is not None:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())


def base_record(args, value, n_gpus, elapsed, batch_B, batch_N, z_stride, extra_config):
    return {
        "metric": "knot-point constraint+Jacobian evals/sec",
        "value": value,
        "unit": "knot-evals/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": dict({"workload": f"BASELINE.json configs[{int(args.workload[-1]) - 1}] (0-based; 'config {args.workload[-1]}' of "
                                    f"BASELINE.md): {WORKLOADS[args.workload]['desc']}; per-GPU shard, constants of the Jacobian "
                                    "pre-written",
                         "jac_format": args.jac_format, "problems_per_gpu": batch_B, "knots": batch_N, "z_stride": int(z_stride)},
                        **extra_config),
    }


def alg_bytes_of(batch, structural):
    return float(np.sum((strict_bytes if structural else algorithmic_bytes)(batch.N, batch.k_trans)))


def roofline_record(args, batch, ms_each):
    structural = args.jac_format == "structural"
    # the bytes a launch must move: dense 15x20 blocks or, when only the structural non-zeros are
    # written, 8d's strict figure
    alg_bytes = float(np.sum((strict_bytes if structural else algorithmic_bytes)(batch.N, batch.k_trans)))
    avg_ms = float(np.mean(ms_each))
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    strict = float(np.sum(strict_bytes(batch.N, batch.k_trans)))
    strict_achieved = strict / (avg_ms * 1e-3) / 1e9
    key = args.workload + ("_structural" if structural else "")
    traffic, traffic_note = measured_traffic(key)
    prof_ms, prof_build = stamped_profile_ms(key, args.placement_trials > 1)
    check = {}
    if prof_ms:
        check["profile_launch_ms_avg"] = prof_ms
        if abs(prof_ms - avg_ms) > 0.05 * prof_ms:
            check["profile_vs_live"] = (f"live HIP-event average {avg_ms:.4f} ms vs {prof_ms:.4f} ms in the tracked rocprofv3 round (kernel build "
                                        f"{prof_build}; this build {kernel_build_id()}): more than 5 % apart -- another box's placement luck or a "
                                        "changed kernel; profiles/ROUNDS.md lists every round")
    return {**check, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_note,
            "kernel": "k_constraint_jacobian", "launch_ms_avg": avg_ms, "launch_ms_min": float(np.min(ms_each)),
            "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_knot_eval": alg_bytes / (batch.B * batch.N),
            # the output format is the dense 15x20 block, so the figure that counts only structurally
            # non-zero entries is quoted beside it
            "strict_nnz": {"bytes_per_knot_eval": strict / (batch.B * batch.N), "achieved": strict_achieved,
                           "frac": strict_achieved / HBM_PEAK_GBS}}


def tail_note():
    return ("value = knot evals / wall clock of the K-launch region only (barrier + device sync on both sides); the end-of-job "
            "tail (eval_f + constraint violation + gather of the per-problem results to GPU 0) follows it and is reported as "
            "gather_ms -- the same for every N and both drivers")


def run_single_process(args):
    """--mode single-process: this process drives all N devices through include/qln_multi.h (one handle, stream and
    buffer set per device; one host thread per device issues its launches; one RCCL gather of the per-problem objective
    and constraint violation to device 0 at the end).  Same workload generator, same timed region, same tail as
    run_ranks: the two drivers' lines are comparable with each other and across N."""
    import torch
    from quadruped_landing_amd import PlanarQuadruped, multi, problem_gen as PG

    n = args.gpus
    # rehearsal of the N > 1 control flow of this driver on a one-GPU box (tests / profiles only): n shards on device 0
    # (qln_multi_create_on_one_device: the gather's exchange is device copies, RCCL admits one rank per device) -- the line
    # says so and its value is NOT a multi-GPU number
    rehearse = os.environ.get("QLN_BENCH_REHEARSE_ON_DEVICE0") == "1"
    if torch.cuda.device_count() < n and not rehearse:
        raise SystemExit(f"bench.py: --gpus {n} but {torch.cuda.device_count()} device(s) visible")
    devs = dict(devices=[0] * n, one_device=True) if rehearse else dict(devices=list(range(n)))
    w = WORKLOADS[args.workload]
    B, N = w["B"], w["N"]
    device_data = not (w["ragged"] or args.host_data)
    model = PlanarQuadruped()
    if device_data:
        kt = np.full(B * n, w["k_trans"], dtype=np.int32)
        im = np.full(B * n, 1, dtype=np.int32)
        xf = np.tile(PG.terminal_state(model), (B * n, 1))
        m = multi.MultiNLP(model, None, im, kt, N, np.zeros((B * n, 15)), xf, jac_format=args.jac_format, **devs)
        m.sample_drop_states([PG.drop_state_sampler(r, model) for r in range(n)])  # shard r = rank r's workload (seed r)
        m.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=False)
        m.initial_guess()
        m.perturb_point([PG.drop_state_sampler(r, model, stream_offset=4 * B) for r in range(n)], sigma=0.05)
        shard0 = PG.LandingBatch(model, N, kt[:B], im[:B], np.zeros((B, 15)), xf[:B], None, None)  # (sizes only: x0 lives on the device)
        data_how = "generated on the device (qln_sample_drop_states / qln_set_lqr_cost / qln_initial_guess / qln_perturb_point), seed = shard index"
    else:
        shards = [PG.make_batch(B, N, w["k_trans"] or 14, 1, seed=r, ragged=w["ragged"], build_obj=False) for r in range(n)]
        cat = lambda name: np.concatenate([getattr(b, name) for b in shards])
        m = multi.MultiNLP(model, None, cat("init_mode"), cat("k_trans"), N, cat("x0"), cat("xf"), jac_format=args.jac_format, **devs)
        m.set_lqr_cost(PG.Q_DIAG, PG.R_DIAG, PG.Q_DIAG, 0.009, per_problem=w["ragged"])
        m.set_Z(cat("Z"))
        shard0 = shards[0]
        data_how = "host generator (problem_gen.make_batch), uploaded; seed = shard index"
    m.synchronize()
    m.alloc_vals(placed=args.placement_trials > 1)
    K, W = args.steps, args.warmup
    what = multi.GATHER_F | multi.GATHER_VIOL
    m.time_c_and_jac(0, max(W, 1))  # warm-up launches, and one tail so that RCCL is initialised
    m.eval_f()
    m.constraint_violation()
    m.gather(what)
    m.synchronize()
    # ---- the timed region: K launches per device, nothing else.  qln_multi_time_* synchronises every device, releases
    # one issue thread per device together and returns when all devices are idle; its own host clock brackets exactly that.
    ms_dev, wall_ms = m.time_c_and_jac(0, K)
    elapsed = wall_ms * 1e-3
    # ---- the end-of-job tail
    tg = time.perf_counter()
    m.eval_f()
    m.constraint_violation()
    m.gather(what)                   # the single end-of-job exchange (RCCL over xGMI)
    m.synchronize()
    t_gather = time.perf_counter() - tg
    m.gather(multi.GATHER_C)         # for the record: the full (ragged) constraint vectors to device 0
    m.synchronize()
    tg = time.perf_counter()
    m.gather(multi.GATHER_C)
    m.synchronize()
    t_gather_c = time.perf_counter() - tg
    total_B = B * n
    out = base_record(args, total_B * N * K / elapsed, n, elapsed, B, N, m.z_stride,
                      {"driver": "one process, one issue thread per device, qln_multi_* (include/qln_multi.h)"
                                 + (f"; REHEARSAL: all {n} shards on device 0 (qln_multi_create_on_one_device, gather by device copies) -- "
                                    "control flow only, not a multi-GPU measurement" if rehearse else ""),
                       "jacobian_buffer": "qln_vals_alloc_placed per device" if args.placement_trials > 1 else "plain allocation",
                       "workload_data": data_how, "timing": tail_note(),
                       "multi_gpu_status": multi_gpu_status(n)})
    per_dev = [float(x) / K for x in ms_dev]
    out["roofline"] = roofline_record(args, shard0, np.array([max(per_dev)]))
    out["roofline"]["launch_ms_avg_per_device"] = per_dev
    out["roofline"]["launch_ms_source"] = "HIP events on each device's stream around its K launches; the slowest device is the one priced"
    out["gather_ms"] = t_gather * 1e3
    out["gather_c_ms"] = t_gather_c * 1e3
    f, viol, _ = m.gathered()
    assert f.shape == (total_B,) and np.all(np.isfinite(f)) and np.all(np.isfinite(viol))
    m.close()
    return out


def multi_gpu_status(n):
    return ("single GPU" if n == 1 else
            f"N = {n}: RCCL between distinct devices had never executed before this run (the build pool exposes one GPU per "
            "box; the N > 1 control flow of both drivers was rehearsed with shards / ranks on one device, shard bookkeeping is "
            "CPU-tested, the one-shard clique GPU-tested)")


class _GlooComm:
    """Stand-in with multi.Comm's interface over the gloo group (host memory): only used when RCCL could not be brought
    up on every rank, so that the job still reports (the JSON line then names the fallback)."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def gather(self, t, root: int = 0, stream=None):
        import torch
        from quadruped_landing_amd import distributed as D

        parts = D.gather_to_root(t.detach().cpu().contiguous(), dst=root)
        if self.rank != root:
            return None, None
        return torch.cat(parts).to(t.device), [int(p.numel()) for p in parts]

    def max(self, value: float) -> float:
        import torch
        import torch.distributed as dist

        v = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v.item())

    def barrier(self):
        import torch.distributed as dist

        dist.barrier()

    def close(self):
        pass


def run_ranks(args):
    """One process per GPU.  Under a launcher (the driver's `python -m torch.distributed.run --nproc-per-node N ...`)
    every rank builds its shard locally and evaluates it with its own handle; there is no data-path collective.  RCCL
    is driven through the C ABI (qln_comm_*, include/qln_multi.h): rank 0's 128-byte id reaches the others over a gloo
    (TCP) rendezvous, which is all torch.distributed is used for."""
    import torch
    import torch.distributed as dist
    from quadruped_landing_amd import multi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("QLN_BENCH_REHEARSE_ON_DEVICE0") == "1":
        # rehearsal of the N > 1 control flow on a one-GPU box (tests / profiles only): every rank uses device 0; RCCL refuses two
        # ranks on one device, so this goes with QLN_BENCH_SIMULATE_RCCL_FAILURE=1 (the tail then travels over gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("QLN_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank
    multi_rank = world > 1 or force_dist
    comm = None
    rccl_error = None
    if multi_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # RCCL through the C ABI.  Should a rank fail to create its communicator, EVERY rank falls back to carrying the
        # end-of-job exchange over the gloo group (host memory): the line is still printed, and says so.
        uid = [None]
        if rank == 0:
            try:
                uid = [multi.Comm.unique_id()]
            except Exception as e:  # noqa: BLE001 -- reported below
                rccl_error = repr(e)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                if os.environ.get("QLN_BENCH_SIMULATE_RCCL_FAILURE") == "1":  # exercises the fallback below (tests only)
                    raise RuntimeError("simulated")
                comm = multi.Comm(uid[0], rank, world, local_rank)
            except Exception as e:  # noqa: BLE001
                rccl_error = repr(e)
        failed = torch.tensor([0 if comm is not None else 1])
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()):
            if comm is not None:
                comm.close()
            rccl_error = rccl_error or "another rank could not create its RCCL communicator"
            print(f"bench.py rank {rank}: RCCL unavailable ({rccl_error}); the gather goes over gloo", file=sys.stderr)
            comm = _GlooComm(rank, world)

    # weak scaling: every rank owns a full-size shard of the global batch (seeded by its rank)
    batch, nlp, Z, c, vals = build(args.workload, seed=rank, device=local_rank, placement_trials=args.placement_trials,
                                   jac_format=args.jac_format, host_data=args.host_data)
    data_how = build.last_data
    placement_ms = list(build.last_trials)
    placement_how = build.last_placement
    f, viol = nlp.new_f(), nlp.new_f()
    K, W = args.steps, args.warmup
    structural = args.jac_format == "structural"

    def barrier():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()

    def gather1(t):
        """the end-of-job gather of one per-problem array to rank 0; with one rank and no communicator it degenerates to
        the device copy a one-rank gather is"""
        if comm is not None:
            return comm.gather(t)[0]
        return t.clone()

    # warmup (untimed): W launches (HIP events around each: the per-launch spread reported beside the average), one objective
    # pass, and one gather so RCCL's channels exist
    ms_warm = nlp.time_c_and_jac(Z, c, vals, warmup=0, iters=max(W, 1))
    nlp.eval_f(Z, f)
    nlp.constraint_violation(c, viol)
    gather1(f)
    gather1(viol)
    barrier()

    # ---- the timed region: K launches, nothing else, bracketed by barrier + device synchronisation on both sides
    # (the K launches are issued back to back with ONE pair of HIP events around them: an event between two launches is a
    # barrier packet that keeps the next launch from starting under the previous one's tail -- 1.070-1.077 ms of wall clock per
    # launch against 1.062-1.065 ms, bench/launch_gap.py; the kernel's average duration = elapsed / K)
    t0 = time.perf_counter()
    ms_total = nlp.time_c_and_jac_total(Z, c, vals, warmup=0, iters=K)
    barrier()
    elapsed = time.perf_counter() - t0
    ms_each = np.array([ms_total / K])
    # ---- the end-of-job tail (the same for every N): per-problem results, then the single gather to rank 0
    tg = time.perf_counter()
    nlp.eval_f(Z, f)
    nlp.constraint_violation(c, viol)
    f_all = gather1(f)               # RCCL over xGMI when there is more than one rank
    viol_all = gather1(viol)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - tg
    barrier()
    gather1(c)                       # for the record: the full (ragged) constraint vectors to rank 0
    barrier()
    tg = time.perf_counter()
    gather1(c)
    torch.cuda.synchronize()
    t_gather_c = time.perf_counter() - tg
    launch_avg = float(np.mean(ms_each))
    launch_avgs = [launch_avg]
    if comm is not None:
        elapsed = comm.max(elapsed)
        t_gather = comm.max(t_gather)
        t_gather_c = comm.max(t_gather_c)
        slowest = comm.max(launch_avg)   # the roofline prices the slowest GPU's launches
        launch_avgs = [launch_avg] if world == 1 else [launch_avg, slowest]

    out = None
    if rank == 0:
        out = base_record(args, batch.B * batch.N * world * K / elapsed, world, elapsed, batch.B, batch.N, nlp.z_stride,
                          {"driver": "one process per GPU" + (", no communicator (one rank: the gather is a device copy)" if comm is None
                                                          else ", RCCL through qln_comm_* (include/qln_multi.h)" if rccl_error is None
                                                          else f", RCCL UNAVAILABLE ({rccl_error}): end-of-job gather over gloo (host memory)"),
                           "jacobian_buffer": placement_how, "placement_trials_ms": placement_ms, "workload_data": data_how + ", seed = rank",
                           "timing": tail_note(), "multi_gpu_status": multi_gpu_status(world)})
        out["roofline"] = roofline_record(args, batch, ms_each if len(launch_avgs) == 1 else np.array([launch_avgs[1]]))
        out["roofline"]["launch_ms_min"] = float(np.min(ms_warm))
        out["roofline"]["launch_ms_warmup_each"] = [float(x) for x in ms_warm]
        out["roofline"]["launch_ms_source"] = ("one pair of HIP events on the rank's stream around the K back-to-back launches of the timed region, / K "
                                               "(launch_ms_min / launch_ms_warmup_each: events around each warm-up launch); rank 0's average"
                                               + ("" if len(launch_avgs) == 1 else f" {launch_avgs[0]:.4f} ms, slowest rank's {launch_avgs[1]:.4f} ms (priced)"))
        assert f_all.numel() == batch.B * world and viol_all.numel() == batch.B * world
        out["gather_ms"] = t_gather * 1e3        # eval_f + constraint violation + their gather: the end-of-job tail, after the timed region
        out["gather_c_ms"] = t_gather_c * 1e3    # full c (c_total doubles per rank)
        if world == 1 and not args.no_other:
            # all four callbacks of an NLP iteration from one read of Z in one launch (qln_eval_all): same workload, same buffers
            gg = nlp.new_Z()
            for _ in range(3):
                nlp.eval_all(Z, f, gg, c, vals, write_constants=False)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
            for a, b_ in ev:
                a.record()
                nlp.eval_all(Z, f, gg, c, vals, write_constants=False)
                b_.record()
            torch.cuda.synchronize()
            ms_all = float(np.mean([a.elapsed_time(b_) for a, b_ in ev]))
            all_bytes = alg_bytes_of(batch, structural) + 8.0 * nlp.n_nlp * batch.B
            out.setdefault("other", {})["eval_all_one_launch"] = {
                "launch_ms_avg": ms_all, "achieved_GBs": all_bytes / (ms_all * 1e-3) / 1e9,
                "roofline_frac": all_bytes / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "qln_eval_all: eval_f + grad_f! + eval_c! + jac_c! from one read of Z; bytes = the hot launch's + the gradient written"}
            del gg
            # the pair a line search asks for: f and c from one read of Z in one launch (qln_eval_objective_and_constraint)
            for _ in range(3):
                nlp.eval_f_and_c(Z, f, c)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
            for a, b_ in ev:
                a.record()
                nlp.eval_f_and_c(Z, f, c)
                b_.record()
            torch.cuda.synchronize()
            ms_fc = float(np.mean([a.elapsed_time(b_) for a, b_ in ev]))
            fc_bytes = float(np.sum(8 * (20 * batch.N - 5) + 8 * (18 * batch.N - batch.k_trans.astype(np.int64) + 16) + 8))
            out["other"]["objective_and_constraint_one_launch"] = {
                "launch_ms_avg": ms_fc, "achieved_GBs": fc_bytes / (ms_fc * 1e-3) / 1e9, "roofline_frac": fc_bytes / (ms_fc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "qln_eval_objective_and_constraint: eval_f + eval_c! from one read of Z (no derivatives); bytes = Z read + c and f written"}
        if world == 1 and not args.no_other and not structural:
            # same workload, structural format (only the non-zeros of every step block are written)
            del vals
            bs, ns, Zs, cs, vs = build(args.workload, seed=rank, device=local_rank, placement_trials=min(2, args.placement_trials), jac_format="structural")
            mss = ns.time_c_and_jac(Zs, cs, vs, warmup=5, iters=K)
            sb = float(np.sum(strict_bytes(bs.N, bs.k_trans)))
            out.setdefault("other", {})["structural_format"] = {
                "launch_ms_avg": float(np.mean(mss)), "knot_evals_per_s": bs.B * bs.N / (float(np.mean(mss)) * 1e-3),
                "bytes_per_knot_eval": sb / (bs.B * bs.N), "achieved_GBs": sb / (float(np.mean(mss)) * 1e-3) / 1e9,
                "roofline_frac": sb / (float(np.mean(mss)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "jac_format=structural (QLN_JAC_FORMAT_STRUCTURAL): 71/56/57 values per step block instead of "
                        "300; priced at SURVEY.md 8d's strict byte count"}
            del bs, ns, Zs, cs, vs
        if world == 1 and not args.no_other and args.workload != "config2":
            b2, n2, Z2, c2, v2 = build("config2", seed=0, device=local_rank, jac_format=args.jac_format)
            ms2 = n2.time_c_and_jac(Z2, c2, v2, warmup=5, iters=50)
            alg2 = float(np.sum((strict_bytes if structural else algorithmic_bytes)(b2.N, b2.k_trans)))
            out.setdefault("other", {})["config2_B1024_N40"] = {"launch_ms_avg": float(np.mean(ms2)),
                                                   "knot_evals_per_s": b2.B * b2.N / (float(np.mean(ms2)) * 1e-3),
                                                   "achieved_GBs": alg2 / (float(np.mean(ms2)) * 1e-3) / 1e9,
                                                   "roofline_frac": alg2 / (float(np.mean(ms2)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                   "note": "BASELINE.json configs[1] (0-based): one wave per problem = 4 waves per CU, "
                                                           "a single round: launch-latency regime"}
            # the caller side of the path on the same configuration: every problem of configs[1] SOLVED
            # from the notebook's initial-guess rule by qln_solve, judged by the evaluator
            n2.solve(n2.initial_guess(), max_outer=1, max_inner=1)  # untimed: the solver's scratch is allocated on first use
            Zs = n2.initial_guess()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            Zs, sinfo = n2.solve(Zs)
            torch.cuda.synchronize()
            ts = time.perf_counter() - ts
            sviol = n2.constraint_violation(n2.eval_c(Zs)).cpu().numpy()
            si = sinfo.cpu().numpy()
            out["other"]["solve_config2_B1024_N40"] = {
                "solved_problems_per_s": float((sviol <= 1e-6 * 1.0001).sum()) / ts, "wall_ms": ts * 1e3,
                "problems": int(b2.B), "solved_to_1e-6": int((sviol <= 1e-6 * 1.0001).sum()),
                "ilqr_iterations_median": float(np.median(si[:, 1])), "ilqr_iterations_max": float(si[:, 1].max()),
                "violation_max": float(sviol.max()), "objective_median": float(np.median(si[:, 2])),
                "note": "qln_solve (augmented-Lagrangian iLQR, one wave per problem) from qln_initial_guess's Z0; violation = "
                        "qln_constraint_violation of the returned Z (Ipopt's definition)"}
            if not args.no_cpu_baseline:
                out["other"]["solve_config2_B1024_N40"]["cpu_baseline"] = cpu_solve_baseline(b2, n2)
            del b2, n2, Z2, c2, v2, Zs
            # ... and every problem of the bench workload itself (65 536 of them at the default): the large-batch regime,
            # two waves per SIMD (uniform workloads; the ragged one draws transition knots no landing is feasible for)
            if not WORKLOADS[args.workload]["ragged"]:
                nlp.solve(nlp.initial_guess(), max_outer=1, max_inner=1)  # untimed: allocates the scratch (10 GB at B = 65 536)
                Zs = nlp.initial_guess()
                torch.cuda.synchronize()
                ts = time.perf_counter()
                Zs, sinfo = nlp.solve(Zs)
                torch.cuda.synchronize()
                ts = time.perf_counter() - ts
                sviol = nlp.constraint_violation(nlp.eval_c(Zs)).cpu().numpy()
                si = sinfo.cpu().numpy()
                out["other"][f"solve_{args.workload}_B{batch.B}_N{batch.N}"] = {
                    "solved_problems_per_s": float((sviol <= 1e-6 * 1.0001).sum()) / ts, "wall_ms": ts * 1e3,
                    "problems": int(batch.B), "solved_to_1e-6": int((sviol <= 1e-6 * 1.0001).sum()),
                    "ilqr_iterations_median": float(np.median(si[:, 1])), "ilqr_iterations_max": float(si[:, 1].max()),
                    "violation_max": float(sviol.max()), "objective_median": float(np.median(si[:, 2])),
                    "note": "the same on the bench workload's own problems (its drop states, the notebook's initial guess)"}
                del Zs, sinfo
        if world == 1 and not args.no_cpu_baseline:
            one, allc = cpu_baseline(batch, nlp, Z)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = allc
    if comm is not None:
        comm.barrier()
        comm.close()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
