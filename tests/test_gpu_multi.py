"""include/qln_multi.h on the GPU box: the multi-GPU layer with however many devices are visible (the pool gives one,
so the degenerate one-shard clique -- which still goes through ncclCommInitAll / ncclCommInitRank, ncclSend / ncclRecv
inside a group, ncclAllGather, ncclAllReduce).  Shard -> handle -> evaluate -> gather must give exactly what one
single-GPU handle over the whole batch gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _single(batch):
    import torch
    from quadruped_landing_amd import HybridNLP

    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z = nlp.upload_Z(batch.Z)
    c, v = nlp.eval_c_and_jac(Z, write_constants=True)
    f, viol = nlp.eval_f(Z), nlp.constraint_violation(c)
    torch.cuda.synchronize()
    return nlp, c.cpu().numpy(), v.cpu().numpy(), f.cpu().numpy(), viol.cpu().numpy()


@pytest.mark.parametrize("ragged,placed", [(False, False), (True, False), (True, True)])
def test_single_process_multi_device_path_equals_the_single_handle(ragged, placed):
    import ctypes as C
    import torch
    from quadruped_landing_amd import _lib, multi, problem_gen as PG

    ndev = torch.cuda.device_count()
    batch = PG.make_batch(37, 25, 9, 1, seed=12, ragged=ragged)
    nlp, c1, v1, f1, viol1 = _single(batch)
    m = multi.MultiNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                       devices=list(range(ndev)))
    m.set_Z(batch.Z)
    m.alloc_vals(placed=placed)
    m.eval_c_and_jac(with_jacobian=True, write_constants=True)
    m.eval_f()
    m.constraint_violation()
    m.gather(multi.GATHER_F | multi.GATHER_VIOL | multi.GATHER_C)  # stream-ordered behind the evaluations
    f, viol, c = m.gathered(c=True)
    assert np.array_equal(f, f1) and np.array_equal(viol, viol1)
    # the gathered constraint vector: problem b at c_off[b]; with one shard the layout is the single handle's
    for b in range(batch.B):
        mm = nlp.problem_dims(b)[0]
        assert np.array_equal(c[m.c_off[b] : m.c_off[b] + mm], nlp.split_c(c1, b)), b
    # shards tile the batch and their Jacobian buffers hold what the single handle computes for those problems
    end = 0
    for r in range(m.n_devices):
        s = m.shard(r)
        assert s["begin"] == end
        end = s["end"]
        dims = _lib.QlnDims()
        _lib.check(_lib.lib().qln_get_dims(s["handle"], C.byref(dims)))
        nb = s["end"] - s["begin"]
        j_off = np.zeros(nb, dtype=np.int64)
        _lib.check(_lib.lib().qln_get_offsets(s["handle"], None, j_off.ctypes.data_as(C.POINTER(C.c_int64))))
        m.synchronize()
        vals = m.shard_tensor(r, "vals").cpu().numpy()
        assert vals.size == dims.j_total
        for i in range(nb):
            b = s["begin"] + i
            nz = nlp.problem_dims(b)[1]
            assert np.array_equal(vals[j_off[i] : j_off[i] + nz], nlp.split_vals(v1, b)), b
    assert end == batch.B
    ms, wall = m.time_c_and_jac(1, 3)   # one issue thread per device, released together
    assert ms.shape == (m.n_devices,) and np.all(ms > 0) and wall >= ms.max() * 0.5
    m.close()


@pytest.mark.parametrize("n_shards,ragged,placed,fmt", [(2, False, False, "dense_blocks"), (3, True, False, "dense_blocks"),
                                                       (2, True, True, "dense_blocks"), (5, True, False, "structural"), (8, False, False, "dense_blocks")])
def test_n_shards_rehearsed_on_one_device_equal_the_single_handle(n_shards, ragged, placed, fmt):
    """The n > 1 branches of the one-process layer -- shard plan, per-shard handles / streams / buffers, the cuts of Z and of a
    per-problem cost table, one issue thread per shard, the gather's per-shard offsets and ragged counts, the sharded solve --
    EXECUTED with n shards on the one GPU this pool gives (qln_multi_create_on_one_device: the send / receive pairs of the
    gather are device copies, because RCCL admits one rank per device; everything else is the code qln_multi_create runs).
    Must give exactly what one handle over the whole batch gives."""
    import ctypes as C
    import torch
    from quadruped_landing_amd import HybridNLP, _lib, multi, problem_gen as PG

    batch = PG.make_batch(37, 25, 9, 1, seed=12 + n_shards, ragged=ragged)
    nlp = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, jac_format=fmt)
    Z1 = nlp.upload_Z(batch.Z)
    c1, v1 = nlp.eval_c_and_jac(Z1, write_constants=True)
    f1, viol1 = nlp.eval_f(Z1), nlp.constraint_violation(c1)
    torch.cuda.synchronize()
    c1, v1, f1, viol1 = (t.cpu().numpy() for t in (c1, v1, f1, viol1))
    m = multi.MultiNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf,
                       devices=[0] * n_shards, jac_format=fmt, one_device=True)
    assert m.n_devices == n_shards
    m.set_Z(batch.Z)
    m.alloc_vals(placed=placed)
    m.eval_c_and_jac(with_jacobian=True, write_constants=True)
    m.eval_f()
    m.constraint_violation()
    m.gather(multi.GATHER_F | multi.GATHER_VIOL | multi.GATHER_C)
    f, viol, c = m.gathered(c=True)
    assert np.array_equal(f, f1) and np.array_equal(viol, viol1)
    for b in range(batch.B):
        mm = nlp.problem_dims(b)[0]
        assert np.array_equal(c[m.c_off[b] : m.c_off[b] + mm], nlp.split_c(c1, b)), b
    end = 0
    for r in range(n_shards):
        s = m.shard(r)
        assert s["begin"] == end and s["device"] == 0
        end = s["end"]
        nb = s["end"] - s["begin"]
        assert nb in (batch.B // n_shards, batch.B // n_shards + 1)
        dims = _lib.QlnDims()
        _lib.check(_lib.lib().qln_get_dims(s["handle"], C.byref(dims)))
        j_off = np.zeros(nb, dtype=np.int64)
        _lib.check(_lib.lib().qln_get_offsets(s["handle"], None, j_off.ctypes.data_as(C.POINTER(C.c_int64))))
        m.synchronize()
        vals = m.shard_tensor(r, "vals").cpu().numpy()
        for i in range(nb):
            b = s["begin"] + i
            nz = nlp.problem_dims(b)[1]
            assert np.array_equal(vals[j_off[i] : j_off[i] + nz], nlp.split_vals(v1, b)), b
    assert end == batch.B
    ms, wall = m.time_c_and_jac(1, 3)   # n issue threads released together
    assert ms.shape == (n_shards,) and np.all(ms > 0)
    # the sharded solve and the gather of its results = the single handle's solve, problem for problem
    info = m.solve()
    m.eval_f()
    m.eval_c_and_jac(with_jacobian=False)
    m.constraint_violation()
    m.gather(multi.GATHER_F | multi.GATHER_VIOL)
    f, viol, _ = m.gathered()
    Zs, info1 = nlp.solve(nlp.upload_Z(batch.Z))
    torch.cuda.synchronize()
    assert np.array_equal(info[:, :10], info1.cpu().numpy()[:, :10])
    assert np.array_equal(f, nlp.eval_f(Zs).cpu().numpy())
    m.close()


def test_one_process_per_gpu_communicator_with_the_ranks_that_fit_on_this_box():
    """qln_comm_*: world = 1 on a one-GPU box (RCCL refuses two ranks on one device).  The id / init / count exchange
    / grouped send+recv / max / barrier sequence is the one bench.py runs under torch.distributed.run."""
    import torch
    from quadruped_landing_amd import multi

    uid = multi.Comm.unique_id()
    assert len(uid) == 128
    comm = multi.Comm(uid, 0, 1, 0)
    t = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    out, counts = comm.gather(t)
    torch.cuda.synchronize()
    assert counts == [1000] and torch.equal(out, t)
    out, counts = comm.gather(t[:0])  # an empty shard is legal
    assert counts == [0] and out.numel() == 0
    assert comm.max(3.25) == 3.25
    comm.barrier()
    comm.close()


def test_sharded_solve_and_the_gather_of_final_results():
    """The whole north-star job through the multi-GPU layer: the batch is sharded, every shard's problems are SOLVED where
    they live (qln_multi_solve), the evaluator judges the solutions per shard, and the one RCCL gather at the end brings
    objective and constraint violation of every problem to the root."""
    import torch
    from quadruped_landing_amd import HybridNLP, multi, problem_gen as PG

    ndev = torch.cuda.device_count()
    batch = PG.make_batch(48, 40, 14, 1, seed=21, noise=0.0)
    m = multi.MultiNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf, devices=list(range(ndev)))
    m.initial_guess()
    info = m.solve()
    m.eval_c_and_jac(with_jacobian=False)
    m.eval_f()
    m.constraint_violation()
    m.gather(multi.GATHER_F | multi.GATHER_VIOL)
    f, viol, _ = m.gathered()
    assert (info[:, 5] == 0).all() and viol.max() <= 1e-6 * 1.0001
    assert np.allclose(f, info[:, 2], rtol=1e-12, atol=0)
    # the same problems through a single handle give the same bits
    one = HybridNLP(batch.model, batch.obj, batch.init_mode, batch.k_trans, batch.N, batch.x0, batch.xf)
    Z1, info1 = one.solve(one.initial_guess())
    assert np.array_equal(one.eval_f(Z1).cpu().numpy(), f)
    m.close()


def test_bench_two_rank_control_flow_rehearsal_on_one_gpu(tmp_path):
    """The driver's N > 1 line -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- rehearsed with
    two ranks on the ONE GPU of this box: both ranks use device 0 (QLN_BENCH_REHEARSE_ON_DEVICE0) and, since RCCL refuses two
    ranks on one device, the end-of-job tail travels over gloo (QLN_BENCH_SIMULATE_RCCL_FAILURE).  Everything else of the
    N > 1 control flow runs as it will on a node: the rendezvous, per-rank shards with seed = rank, the barrier-bracketed
    K-launch region with the max over ranks, the gather of 2 x B per-problem results to rank 0, ONE JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys

    from tests.helpers import ROOT

    env = dict(os.environ, QLN_BENCH_REHEARSE_ON_DEVICE0="1", QLN_BENCH_SIMULATE_RCCL_FAILURE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--workload",
           "config2", "--placement-trials", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints, one line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak"
    B, N = d["config"]["problems_per_gpu"], d["config"]["knots"]
    assert abs(d["value"] - 2 * B * N / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]   # whole-job aggregate over both ranks
    assert "RCCL UNAVAILABLE" in d["config"]["driver"] and d["gather_ms"] > 0 and d["gather_c_ms"] > 0
    assert d["roofline"]["launch_ms_avg"] <= d["ms_per_step"] * 1.5


@pytest.mark.parametrize("workload,n", [("config2", 3), ("config3", 2)])
def test_bench_single_process_driver_rehearsed_with_n_shards_on_one_gpu(workload, n):
    """`python bench.py --mode single-process --gpus N` rehearsed with N shards on the ONE GPU of this box
    (QLN_BENCH_REHEARSE_ON_DEVICE0: qln_multi_create_on_one_device): device-generated shards with seed = shard index, per-shard buffers, N issue threads, the K-launch region, the tail with the gather of N x B
    per-problem results, ONE JSON line that says it is a rehearsal."""
    import json
    import os
    import subprocess
    import sys

    from tests.helpers import ROOT

    env = dict(os.environ, QLN_BENCH_REHEARSE_ON_DEVICE0="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "single-process", "--gpus", str(n), "--steps", "4", "--warmup", "1",
           "--workload", workload, "--placement-trials", "1", "--no-cpu-baseline", "--no-other"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 4 and d["scaling"] == "weak" and "REHEARSAL" in d["config"]["driver"]
    B, N = d["config"]["problems_per_gpu"], d["config"]["knots"]
    assert abs(d["value"] - n * B * N / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert len(d["roofline"]["launch_ms_avg_per_device"]) == n and d["gather_ms"] > 0 and d["gather_c_ms"] > 0
