"""CPU, 2 processes over gloo: the N>1 path of the frame loop -- frame sharding
(apportion), the sum of the double accumulators (MPI_Reduce + MPI_Bcast in the reference,
psp_process.cpp:1866-1872, 2019-2023), the time-series exchange (global_transpose,
psp_process.cpp:707-771) and the gather-to-root option.  Per-rank rows are produced by the
oracle (test infrastructure) so the collectives are checked against a single-process run
of the same frames.

The exchanges of the PRODUCT run through the library (upsp_exchange_* over RCCL, tests/test_exchange_gpu.py and the
multi-process GPU tests); the torch.distributed form of the same bookkeeping exists for these CPU tests and is reachable
only under UPSP_ALLOW_TORCH_EXCHANGE=1 -- `test_no_silent_torch_exchange` checks that."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, F, N, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["UPSP_ALLOW_TORCH_EXCHANGE"] = "1"       # the bookkeeping on torch.distributed (no GPU here): tests only
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from upsp_processing_amd import distributed as D
    shard = D.Shard(F, N)
    assert shard.rank == rank and shard.world == world
    rng = np.random.default_rng(seed)
    rows_all = rng.normal(size=(F, N)).astype(np.float32)       # what one process would produce
    rows_all[:, ::7] = np.nan                                   # skipped nodes
    f0, nf = shard.my_frames
    mine = rows_all[f0:f0 + nf]
    s = torch.as_tensor(np.nansum(mine.astype(np.float64), axis=0))
    ss = torch.as_tensor(np.nansum(mine.astype(np.float64) ** 2, axis=0))
    D.allreduce_sums(s, ss)
    rows_t = torch.as_tensor(np.ascontiguousarray(mine.T))
    series = D.exchange_time_series(rows_t, shard)
    # the same exchange pipelined in 3 chunks must give the same slice
    ex = D.TimeSeriesExchange(shard, 3, device="cpu")
    for k in range(3):
        c0, fc = ex.my_chunk(k)
        ex.submit(rows_t[:, c0:c0 + fc].contiguous())
    assert torch.equal(ex.finish().view(torch.int32), series.view(torch.int32))
    # ... and once more sending only the rows of nodes some camera sees (the others are NaN rows
    # every rank can fill in by itself)
    ex2 = D.TimeSeriesExchange(shard, 3, device="cpu")
    ex2.set_skipped(torch.as_tensor(np.isnan(rows_all[0])))
    for k in range(3):
        c0, fc = ex2.my_chunk(k)
        ex2.submit(rows_t[:, c0:c0 + fc])
    assert torch.equal(ex2.finish().view(torch.int32), series.view(torch.int32))
    assert sum(ex2.vis_count) == int((~np.isnan(rows_all[0])).sum())
    # ... and with the rows packed by the producer (what the gather does with a row map)
    ex3 = D.TimeSeriesExchange(shard, 3, device="cpu")
    ex3.set_skipped(torch.as_tensor(np.isnan(rows_all[0])))
    rm = ex3.row_map()
    assert int((rm >= 0).sum()) == ex3.packed_rows() and torch.equal(rm[ex3.vis], torch.arange(ex3.packed_rows(), dtype=torch.int32))
    for k in range(3):
        c0, fc = ex3.my_chunk(k)
        ex3.submit(rows_t[:, c0:c0 + fc].index_select(0, ex3.vis), packed=True)
    assert torch.equal(ex3.finish().view(torch.int32), series.view(torch.int32))
    # a second pass through the same exchange delivers a complete slice again: the rows that do not travel are written
    # by every finish(), not once
    ex3.out.zero_()
    for k in range(3):
        c0, fc = ex3.my_chunk(k)
        ex3.submit(rows_t[:, c0:c0 + fc].index_select(0, ex3.vis), packed=True)
    assert torch.equal(ex3.finish().view(torch.int32), series.view(torch.int32))
    n0, nn = shard.my_nodes
    # phase-2 per-node vectors: each rank owns its node slice, every rank gets the whole vector
    whole = D.gather_node_vector(torch.arange(n0, n0 + nn, dtype=torch.float32) * 2.0, shard)
    assert torch.equal(whole, torch.arange(N, dtype=torch.float32) * 2.0)
    full = D.gather_time_series_to_root(series, shard)
    q.put((rank, s.numpy(), ss.numpy(), series.numpy(), n0, nn,
           None if full is None else full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,N", [(11, 37), (4, 5), (1, 3)])
def test_two_rank_exchange(F, N):
    world, seed = 2, 123
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, N, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(seed)
    rows_all = rng.normal(size=(F, N)).astype(np.float32)
    rows_all[:, ::7] = np.nan
    ref_t = rows_all.T
    tot = np.nansum(rows_all.astype(np.float64), axis=0)
    tot2 = np.nansum(rows_all.astype(np.float64) ** 2, axis=0)
    for rank, s, ss, series, n0, nn, full in res:
        assert np.allclose(s, tot, rtol=1e-13) and np.allclose(ss, tot2, rtol=1e-13)
        assert series.shape == (nn, F)
        assert np.array_equal(series.view(np.int32), np.ascontiguousarray(ref_t[n0:n0 + nn]).view(np.int32))
        if rank == 0:
            assert np.array_equal(full.view(np.int32), np.ascontiguousarray(ref_t).view(np.int32))
        else:
            assert full is None


def _worker_u16(rank, world, port, F, N, seed, q):
    """The exchange with u16 chunks (integer-valued series: one camera, no weights) must deliver
    the same f32 series as the f32 exchange."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["UPSP_ALLOW_TORCH_EXCHANGE"] = "1"       # the bookkeeping on torch.distributed (no GPU here): tests only
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from upsp_processing_amd import distributed as D
    shard = D.Shard(F, N)
    rng = np.random.default_rng(seed)
    rows_all = rng.integers(0, 65536, size=(F, N)).astype(np.float32)
    rows_all[0, 1], rows_all[0, 2] = 65535.0, 32768.0          # sign bit of the int16 view
    rows_all[:, ::5] = np.nan
    f0, nf = shard.my_frames
    rows_t = torch.as_tensor(np.ascontiguousarray(rows_all[f0:f0 + nf].T))
    series = D.exchange_time_series(rows_t, shard)
    ex = D.TimeSeriesExchange(shard, 3, device="cpu")
    ex.set_skipped(torch.as_tensor(np.isnan(rows_all[0])))
    vis = ex.vis.numpy()
    for k in range(3):
        c0, fc = ex.my_chunk(k)
        blk = rows_t.numpy()[vis, c0:c0 + fc].astype(np.uint16)
        ex.submit(torch.from_numpy(np.ascontiguousarray(blk)), packed=True)
    got = ex.finish()
    assert got.dtype == torch.float32
    assert torch.equal(got.view(torch.int32), series.view(torch.int32))
    q.put(rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,N", [(200, 41), (3, 6)])
def test_two_rank_exchange_u16_wire(F, N):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_u16, args=(r, world, port, F, N, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert sorted(q.get(timeout=120) for _ in range(world)) == [0, 1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_shard_matches_reference_apportion(oracle):
    from upsp_processing_amd import distributed as D, engine
    for value, bins in [(100000, 8), (10, 4), (3, 5), (0, 2), (12345, 7)]:
        st, ex = D.apportion(value, bins)
        so, eo = oracle.apportion(value, bins)
        assert st == so.tolist() and ex == eo.tolist()
        st2, ex2 = engine.apportion(value, bins)
        assert st2 == st and ex2 == ex
    sh = D.Shard(100000, 500958, rank=3, world=8)
    assert sh.my_frames == (37500, 12500) and sum(sh.node_count) == 500958


def _worker_pixels(rank, world, port, F, N, A, seed, q):
    """Pixel-series mode: the ranks exchange the series of the active pixels, each destination the pixels its node slice
    reads, and the owner of a node forms its series and its accumulators over ALL frames."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["UPSP_ALLOW_TORCH_EXCHANGE"] = "1"       # the bookkeeping on torch.distributed (no GPU here): tests only
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from upsp_processing_amd import distributed as D
    shard = D.Shard(F, N)
    rng = np.random.default_rng(seed)
    pixel_series = rng.integers(0, 65536, size=(A, F)).astype(np.int32)       # what pass A leaves: [active pixel][frame]
    pixel_series[0, 0], pixel_series[A - 1, F - 1] = 65535, 32768             # sign bit of the int16 view
    node_k = rng.integers(-1, A, size=N).astype(np.int32)                     # several nodes per pixel, some without
    skipped = (np.arange(N) % 6 == 1)
    f0, nf = shard.my_frames
    ex = D.TimeSeriesExchange(shard, 3, device="cpu")
    ex.set_pixels(torch.as_tensor(node_k), torch.as_tensor(skipped))
    for k in range(3):
        c0, fc = ex.my_chunk(k)
        comp = np.zeros((A, fc + 5), np.uint16)                                # this rank's compact buffer of the chunk (padded pitch)
        comp[:, :fc] = pixel_series[:, f0 + c0:f0 + c0 + fc].astype(np.uint16)
        ex.submit_pixels(torch.from_numpy(comp))
    s, ss = torch.zeros(N, dtype=torch.float64), torch.zeros(N, dtype=torch.float64)
    series = ex.finish_pixels(s, ss).clone()
    D.allreduce_sums(s, ss)
    rows_out, rows_in = ex.pixel_rows()
    # second pass with assume_same: verified on the "device"
    ex.set_pixels(torch.as_tensor(node_k), torch.as_tensor(skipped), assume_same=True)
    ex.verify()
    n0, nn = shard.my_nodes
    q.put((rank, series.numpy(), s.numpy(), ss.numpy(), n0, nn, rows_out, rows_in))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("F,N,A", [(23, 61, 17), (5, 9, 4), (2, 3, 1)])
def test_two_rank_pixel_series_exchange(F, N, A):
    world, seed = 2, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pixels, args=(r, world, port, F, N, A, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(seed)
    pixel_series = rng.integers(0, 65536, size=(A, F)).astype(np.int32)
    pixel_series[0, 0], pixel_series[A - 1, F - 1] = 65535, 32768
    node_k = rng.integers(-1, A, size=N).astype(np.int32)
    skipped = (np.arange(N) % 6 == 1)
    want = np.where((node_k >= 0)[:, None], pixel_series[np.clip(node_k, 0, None)], 0).astype(np.float32)
    want[skipped] = np.nan
    ws, wss = want.astype(np.float64).sum(1), (want * want).astype(np.float64).sum(1)
    for rank, series, s, ss, n0, nn, rows_out, rows_in in res:
        assert series.shape == (nn, F)
        assert np.array_equal(series.view(np.int32), want[n0:n0 + nn].view(np.int32))
        ok = ~skipped
        assert np.array_equal(s[ok], ws[ok]) and np.array_equal(ss[ok], wss[ok]) and np.isnan(s[~ok]).all()
        assert rows_in <= A and rows_out <= world * A


def test_chunk_count_bounds_every_chunk():
    """The chunked frame loop reads a chunk into buffers of `limit` frames: no chunk of any rank may be longer, although
    the cuts sit on 64-frame boundaries (ceil(n / limit) chunks are not enough: 705 -> 3 chunks, the last of 257)."""
    from upsp_processing_amd import distributed as D
    assert max(D.aligned_chunks(705, 3)[1]) == 257              # what the loop used before
    for total, world in ((705, 1), (1410, 2), (60000, 8), (20000, 4), (100000, 8), (63, 2), (1, 3)):
        counts = D.apportion(total, world)[1]
        for limit in (128, 256, 1024):
            K = D.chunk_count(counts, limit)
            for n in counts:
                starts, ext = D.aligned_chunks(n, K)
                assert max(ext) <= limit and sum(ext) == n
                assert all(s % 64 == 0 for s in starts)
            assert K == 1 or any(max(D.aligned_chunks(n, K - 1)[1]) > limit for n in counts)     # and no more chunks than needed
    for n in range(1, 3000):
        K = D.chunk_count([n], 256)
        assert max(D.aligned_chunks(n, K)[1]) <= 256
    with pytest.raises(ValueError):
        D.chunk_count([1000], 100)


def _worker_no_fallback(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("UPSP_ALLOW_TORCH_EXCHANGE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from upsp_processing_amd import distributed as D
    shard = D.Shard(10, 6)
    got = []
    for call in (lambda: D.TimeSeriesExchange(shard, 1, device="cpu"),
                 lambda: D.allreduce_sums(torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)),
                 lambda: D.exchange_time_series(torch.zeros((6, shard.my_frames[1])), shard)):
        try:
            call()
            got.append("ran")
        except D.ExchangeUnavailable:
            got.append("refused")
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_no_silent_torch_exchange():
    """Without UPSP_ALLOW_TORCH_EXCHANGE a multi-rank group whose exchanges cannot run through the library is an error on every
    rank -- the product has one exchange path (a failed RCCL bring-up used to switch to torch operators)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_no_fallback, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, ["refused"] * 3), (1, ["refused"] * 3)]
