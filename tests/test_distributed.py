"""N>1 path: ranks shard a batch by contiguous ranges and gather (f, c) to rank 0, exactly the exchange bench.py does
over RCCL.  On CPU two gloo ranks run it with the oracle standing in for the evaluator (uniform and ragged shards);
the GPU-marked variant runs the HIP evaluator in both ranks (same device, results gathered over gloo) and checks the
gathered vectors against the oracle evaluated on the GLOBAL problem index."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quadruped_landing_amd import distributed as D


def test_shard_ranges_cover_and_balance():
    for n in (1, 7, 64, 65536, 524288):
        for w in (1, 2, 3, 8):
            r = [D.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _offsets(N, k_trans, align=16):
    """c_off of a shard as the product lays it out: exclusive scan of m_nlp rounded up to `align`, no padding behind
    the last problem (quadruped_landing_amd/csrc/qln_api.cpp, qln_create)."""
    off, o = [], 0
    for kt in k_trans:
        o = (o + align - 1) // align * align
        off.append(o)
        o += 18 * N - int(kt) + 16
    return np.array(off, dtype=np.int64), o


def _oracle_shard(batch, lo, hi):
    from oracle import oracle as O
    from tests.helpers import oracle_model

    N, nb = batch.N, hi - lo
    c_off, c_total = _offsets(N, batch.k_trans[lo:hi])
    j_off = np.arange(nb, dtype=np.int64) * 40000
    obj = batch.obj if batch.obj.ndim == 2 else batch.obj[lo:hi]
    out = O.batch_eval(N, oracle_model(batch.model), batch.k_trans[lo:hi], batch.init_mode[lo:hi], batch.x0[lo:hi],
                       batch.xf[lo:hi], obj, batch.Z[lo:hi].reshape(-1), batch.Z.shape[1], c_off, j_off,
                       c_total, nb * 40000, True, False, True, False, 1)
    return out["f"], out["c"][:c_total]


def _hip_shard(batch, lo, hi):
    from quadruped_landing_amd import HybridNLP

    obj = batch.obj if batch.obj.ndim == 2 else batch.obj[lo:hi]
    nlp = HybridNLP(batch.model, obj, batch.init_mode[lo:hi], batch.k_trans[lo:hi], batch.N, batch.x0[lo:hi],
                    batch.xf[lo:hi], device=0)
    Z = nlp.upload_Z(batch.Z[lo:hi])
    c, f = nlp.eval_c(Z), nlp.eval_f(Z)
    torch.cuda.synchronize()
    return f.cpu().numpy(), c.cpu().numpy()


def _worker(rank, world, port, q, ragged, use_gpu):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quadruped_landing_amd import problem_gen as PG

        total, N = (11, 9) if ragged else (10, 9)  # 11 problems over 2 ranks: 6 + 5, and k_trans differs per problem
        batch = PG.make_batch(total, N, 4, 1, seed=5, ragged=ragged)
        lo, hi = D.shard_range(total, rank, world)
        f, c = (_hip_shard if use_gpu else _oracle_shard)(batch, lo, hi)
        f, c = torch.from_numpy(np.ascontiguousarray(f)), torch.from_numpy(np.ascontiguousarray(c))
        sizes = D.gather_sizes(c.numel(), c)
        fs, cs = D.gather_results(f, c)
        tmax = D.max_over_ranks(float(rank + 1))
        if rank == 0:
            ok = tmax == float(world) and [x.numel() for x in cs] == sizes
            if ragged:
                ok = ok and len(set(sizes)) > 1  # the case a plain dist.gather cannot do
            # reference: the oracle on the global problem index, one problem at a time
            for r in range(world):
                a, b = D.shard_range(total, r, world)
                fr, cr = _oracle_shard(batch, a, b)
                got_f, got_c = fs[r].numpy(), cs[r].numpy()
                if use_gpu:
                    fin = np.isfinite(cr)
                    ok = ok and np.array_equal(got_f, fr) and np.allclose(got_c[fin], cr[fin], rtol=1e-12, atol=1e-12)
                else:
                    ok = ok and np.array_equal(got_f, fr) and np.array_equal(got_c, cr, equal_nan=True)  # NaN = padding
            q.put(bool(ok))
        else:
            assert fs is None and cs is None
    finally:
        dist.destroy_process_group()


def _run_two_ranks(ragged, use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, ragged, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_shard_and_gather_gloo(ragged):
    _run_two_ranks(ragged, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_hip_shards_gathered_match_the_global_oracle(ragged):
    """The product's shard -> handle -> evaluate -> gather chain end to end: two ranks, each with its own handle over
    its shard_range of the batch (both on the one visible device), gathered to rank 0 and compared with the oracle
    evaluated on the global index."""
    _run_two_ranks(ragged, use_gpu=True)


def _gloo_comm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench

        comm = bench._GlooComm(rank, world)
        t = torch.arange(5 + 3 * rank, dtype=torch.float64) + 100 * rank  # ragged: 5 and 8 entries
        out, counts = comm.gather(t)
        m = comm.max(float(10 - rank))
        comm.barrier()
        if rank == 0:
            want = torch.cat([torch.arange(5, dtype=torch.float64), torch.arange(8, dtype=torch.float64) + 100])
            q.put(bool(torch.equal(out, want) and counts == [5, 8] and m == 10.0))
        else:
            assert out is None and m == 10.0
    finally:
        dist.destroy_process_group()


def test_bench_fallback_comm_over_gloo():
    """bench.py's stand-in for multi.Comm when RCCL cannot be brought up on every rank: ragged gather, max, barrier."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _empty_shard_worker(rank, world, port, q, empty_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 0 if rank == empty_rank else 5 + 3 * rank
        t = torch.arange(n, dtype=torch.float64) + 100.0 * rank
        out = D.gather_to_root(t)
        if rank == 0:
            ok = len(out) == world
            for r in range(world):
                want = torch.arange(0 if r == empty_rank else 5 + 3 * r, dtype=torch.float64) + 100.0 * r
                ok = ok and torch.equal(out[r], want)
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("empty_rank", [0, 1, 2])
def test_gather_with_an_empty_shard_keeps_every_rank_in_the_exchange(empty_rank):
    """A rank whose shard is empty (fewer problems than ranks, a filtered batch) still takes part in the gather -- under
    the nccl backend a rank that sits a batch out can leave its peers' lazy communicator set-up waiting.  Three gloo
    ranks, the empty one being the root, a middle rank or the last."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 3, port, q, empty_rank)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
