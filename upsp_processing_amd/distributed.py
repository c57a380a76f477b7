"""Frame sharding across GPUs and the end-of-run exchanges of psp_process phase 1.

One process per GPU.  torch.distributed starts the job and carries the rendezvous; the DATA of the
end-of-run exchanges moves through the library's own C-ABI exchange (include/upsp_gpu.h section 3b:
upsp_comm_* / upsp_allreduce_sums / upsp_exchange_*, RCCL over xGMI) whenever the ranks sit on GPUs
-- the same entry points a C++ psp_process binds -- whatever backend the process group itself uses for the
rendezvous ("nccl"; "gloo" when several ranks are rehearsed on one GPU with UPSP_RCCL_LIBRARY naming an RCCL that
allows it).  There is no fallback: a communicator that does not come up ends the run.  Only under
UPSP_ALLOW_TORCH_EXCHANGE=1 (CPU tests of the bookkeeping) the same steps run here on torch.distributed calls.
Frames shard trivially -- the reference does the same across MPI ranks (cpp/exec/psp_process.cpp:1519-1529) -- so the data path has no collective until the end of the run,
where three exchanges happen:

* sum of the per-rank double accumulators: MPI_Reduce + MPI_Bcast in the reference
  (psp_process.cpp:1866-1872, 2019-2023) -> one all_reduce(SUM) of 2 x N doubles;
* the time-series exchange: every rank holds full rows [frames_r x N] and must end
  with full time series [nodes_r x F] -- global_transpose (psp_process.cpp:707-771:
  local transpose, one block per rank pair, strided placement).  On xGMI every GPU
  pair has its own link, so the exchange is ONE group of point-to-point sends / receives in which
  all 7 links of every GPU carry exactly one block at the same time;
* optional gather of everything to rank 0 (BASELINE north-star wording).
"""
import os

import torch
import torch.distributed as dist

# UPSP_FORCE_COLLECTIVES=1: issue the collectives even in a one-rank group (a one-GPU box can then run
# the all_reduce / all_to_all_single calls through RCCL itself -- two ranks on one GPU are refused by RCCL)
FORCE_COLLECTIVES = bool(os.environ.get("UPSP_FORCE_COLLECTIVES"))


_LIB_COMM = {}


def torch_exchange_allowed():
    """UPSP_ALLOW_TORCH_EXCHANGE=1: the exchanges of a multi-rank group may run on torch.distributed calls with the owner's pass B
    in torch operators (CPU tests of the bookkeeping over gloo; no GPU needed).  NOT a product path: without the switch a
    group whose exchanges cannot run through the library (upsp_comm_* / upsp_exchange_* of libupsp_gpu.so) is an error."""
    return bool(os.environ.get("UPSP_ALLOW_TORCH_EXCHANGE"))


class ExchangeUnavailable(RuntimeError):
    pass


def _need_library(what):
    raise ExchangeUnavailable(
        "upsp: %s needs the library's exchange (upsp_comm_* / upsp_exchange_* over RCCL, device tensors); there is no fallback "
        "(UPSP_ALLOW_TORCH_EXCHANGE=1 lets the CPU tests run the bookkeeping on torch.distributed)" % what)


def lib_comm(group=None):
    """The library's RCCL communicator of the process group (created on first use: rank 0 makes the
    ncclUniqueId, torch.distributed broadcasts its 128 bytes -- whatever the group's backend --, every rank calls
    upsp_comm_create on its current device).  A communicator that does not come up on EVERY rank ends the run
    (ExchangeUnavailable on every rank): the exchanges have one path.  None only without a GPU, or for a non-RCCL group
    under UPSP_ALLOW_TORCH_EXCHANGE=1 (tests)."""
    if not torch.cuda.is_available():
        return None
    if not dist.is_initialized():
        # one rank, no process group: the same entry points through the library's in-process transport
        # (upsp_comm_create_local with one rank: device-to-device copies), so that a single-GPU run of the chunked loop
        # executes the code a multi-GPU run executes
        if "local1" not in _LIB_COMM:
            import ctypes as C
            from . import _capi
            arr = (C.c_void_p * 1)()
            _capi.check(_capi.lib().upsp_comm_create_local(1, arr))
            _LIB_COMM["local1"] = C.c_void_p(arr[0])
        return _LIB_COMM["local1"]
    on_rccl = dist.get_backend(group) == "nccl"
    if not on_rccl and torch_exchange_allowed():
        return None
    key = id(group)
    if key not in _LIB_COMM:
        import ctypes as C
        from . import _capi
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = "cuda" if on_rccl else "cpu"          # (the rendezvous tensors live where the group's backend can reach them)
        buf = (C.c_uint8 * 128)()
        err = None
        if rank == 0:
            try:
                _capi.check(_capi.lib().upsp_comm_unique_id(buf))
            except _capi.UpspError as e:            # e.g. no librccl the library can resolve in this process
                err = e
        # the id travels with an ok flag (byte 128): when rank 0 has no id to give, NO rank calls upsp_comm_create -- with an all-zero
        # id the others would sit in ncclCommInitRank until the bootstrap timeout while rank 0 never joins
        t = torch.tensor(list(buf) + [0 if err else 1], dtype=torch.uint8, device=dev)
        if world > 1:
            dist.broadcast(t, src=0, group=group)
        got = t.cpu().tolist()
        ident = (C.c_uint8 * 128)(*got[:128])
        if not got[128] and err is None:
            err = ExchangeUnavailable("rank 0 could not create the communicator id")
        h = C.c_void_p()
        if err is None:
            try:
                _capi.check(_capi.lib().upsp_comm_create(ident, rank, world, C.byref(h)))
            except _capi.UpspError as e:
                err = e
        if world > 1:
            # every rank or none: a communicator that came up on some ranks only would hang the first exchange
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0:
                if err is None:
                    _capi.lib().upsp_comm_destroy(h)
                raise ExchangeUnavailable("upsp: the library's RCCL communicator could not be created on every rank of the group "
                                          "(rank %d of %d: %s)" % (rank, world, err or "another rank failed"))
        elif err is not None:
            raise ExchangeUnavailable("upsp: the library's RCCL communicator could not be created (%s)" % err)
        _LIB_COMM[key] = h
    return _LIB_COMM[key]


def comm_ranks(group=None):
    """(rank, ranks) of the library's RCCL communicator as RCCL reports them (ncclCommUserRank / ncclCommCount through
    upsp_comm_rank); None when the exchanges of this group do not run through it."""
    h = lib_comm(group)
    if h is None or not dist.is_initialized():
        return None
    import ctypes as C
    from . import _capi
    r, w = C.c_int(-1), C.c_int(-1)
    _capi.check(_capi.lib().upsp_comm_rank(h, C.byref(r), C.byref(w)))
    return int(r.value), int(w.value)


def init_from_env(backend=None):
    """One process per GPU: reads RANK / LOCAL_RANK / WORLD_SIZE (torchrun, or psp_process -ranks=N),
    selects this rank's device BEFORE any other GPU call and joins the process group ("nccl" = RCCL
    over xGMI; UPSP_BACKEND=gloo with UPSP_ONE_GPU=1 rehearses several ranks on one GPU).  The
    equivalent of MPI_Init + MPI_Comm_rank/size in the reference (cpp/exec/psp_process.cpp:1322-1330).
    Returns (rank, world).  Without the environment: (0, 1), nothing initialised."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if os.environ.get("UPSP_ONE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("UPSP_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def shutdown():
    if _LIB_COMM:
        from . import _capi
        torch.cuda.synchronize()
        for h in _LIB_COMM.values():
            if h is not None:
                _capi.lib().upsp_comm_destroy(h)
        _LIB_COMM.clear()
    if dist.is_initialized():
        dist.destroy_process_group()


def create_shared_file(path, nbytes, group=None):
    """Rank 0 creates / truncates a flat output file every rank then pwrite()s its slice into
    (the reference opens intensity_transpose once and writes at byte offsets, psp_process.cpp:958-963);
    the barrier keeps the other ranks from writing into a file that is about to be truncated."""
    import os
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(int(nbytes))
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group)
    return os.open(path, os.O_RDWR)


def apportion(value, nbins):
    """apportion (psp_process.cpp:611-624): contiguous near-equal ranges."""
    block, rem = divmod(int(value), int(nbins))
    start, extent, nxt = [], [], 0
    for b in range(nbins):
        start.append(nxt)
        extent.append(block + (1 if b < rem else 0))
        nxt += extent[-1]
    return start, extent


class Shard:
    """Frame / node ownership of this rank (rank_start_frame, rank_num_frames,
    rank_start_node, rank_num_nodes of psp_process.cpp:1519-1529)."""

    def __init__(self, nframes, nnodes, rank=None, world=None):
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank, self.world = rank, world
        self.nframes, self.nnodes = nframes, nnodes
        self.frame_start, self.frame_count = apportion(nframes, world)
        self.node_start, self.node_count = apportion(nnodes, world)

    @property
    def my_frames(self):
        return self.frame_start[self.rank], self.frame_count[self.rank]

    @property
    def my_nodes(self):
        return self.node_start[self.rank], self.node_count[self.rank]


def allreduce_sums(total, sumsq, group=None):
    """Sum the double accumulators over ranks (in place)."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES):
        comm = lib_comm(group) if total.is_cuda else None
        if comm is not None:       # RCCL through the C ABI (upsp_allreduce_sums)
            import ctypes as C
            from . import _capi
            assert total.is_contiguous() and sumsq.is_contiguous() and total.dtype == torch.float64
            _capi.check(_capi.lib().upsp_allreduce_sums(comm, C.c_void_p(total.data_ptr()), C.c_void_p(sumsq.data_ptr()),
                                                        total.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return total, sumsq
        if not torch_exchange_allowed():
            _need_library("the sum of the accumulators over the ranks")
        both = torch.stack([total, sumsq])
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        total.copy_(both[0])
        sumsq.copy_(both[1])
    return total, sumsq


def exchange_time_series(rows_t, shard, group=None, out=None):
    """global_transpose (psp_process.cpp:707-771).

    rows_t : [N, frames_r] f32 -- this rank's frames, already node-major (the local
             transpose is done by the frame pipeline / upsp_transpose_f32).
    returns [nodes_r, F] f32 -- complete time series of this rank's node slice."""
    r, w = shard.rank, shard.world
    n0, nn = shard.my_nodes
    f_r = shard.frame_count[r]
    assert rows_t.shape == (shard.nnodes, f_r) and (rows_t.shape[1] <= 1 or rows_t.stride(1) == 1)
    if w == 1 or not dist.is_initialized():
        if out is None:
            return rows_t                      # single rank: already the complete series
        out.copy_(rows_t)
        return out
    if rows_t.is_cuda and rows_t.dtype == torch.float32 and dist.is_initialized() and lib_comm(group) is not None:
        # RCCL through the C ABI (upsp_exchange_*): one chunk, every row travels
        x = TimeSeriesExchange(shard, 1, device=rows_t.device, group=group)
        x.set_skipped(None)
        x.submit(rows_t.contiguous(), packed=True)
        res = x.finish()
        if out is not None:
            out.copy_(res)
            res = out
        torch.cuda.current_stream().synchronize()
        x.close()
        return res
    if not torch_exchange_allowed():
        _need_library("the time-series exchange")
    if out is None:
        out = torch.empty((nn, shard.nframes), dtype=rows_t.dtype, device=rows_t.device)
    rows_t = rows_t.contiguous()               # padded row pitch (engine.series_ld) -> packed blocks
    in_split = [shard.node_count[d] * f_r for d in range(w)]       # block for rank d
    out_split = [nn * shard.frame_count[s] for s in range(w)]      # block from rank s
    recv = torch.empty(sum(out_split), dtype=rows_t.dtype, device=rows_t.device)
    dist.all_to_all_single(recv, rows_t.reshape(-1), out_split, in_split, group=group)
    off = 0
    for s in range(w):
        fs = shard.frame_count[s]
        if fs and nn:
            out[:, shard.frame_start[s]:shard.frame_start[s] + fs] = recv[off:off + nn * fs].view(nn, fs)
        off += nn * fs
    return out


def _widen_u16(blk):
    """u16 -> f32 with operators every backend has (uint16 tensors support little else)."""
    return blk.view(torch.int16).to(torch.int32).bitwise_and_(0xFFFF).to(torch.float32)


def _fill_rows(out, rows, value):
    """out[rows, :] = value ; library kernel on the GPU, torch indexing on the CPU."""
    if rows.numel() == 0:
        return
    if out.is_cuda:
        import ctypes as C
        from . import _capi
        _capi.check(_capi.lib().upsp_fill_rows_f32(
            C.c_float(value), rows.numel(), out.shape[1], C.c_void_p(rows.data_ptr()), C.c_void_p(out.data_ptr()),
            out.stride(0), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    else:
        out[rows] = value


def _scatter_rows(out, rows, col, blk):
    """out[rows, col:col+blk.shape[1]] = blk ; library kernel on the GPU, torch indexing on the CPU.
    blk is f32, or u16 (the exchange's wire format for integer-valued series), widened here."""
    if out.is_cuda:
        import ctypes as C
        from . import _capi
        blk = blk.contiguous()
        dst = out[:, col:]
        fn = _capi.lib().upsp_scatter_rows_u16 if blk.dtype == torch.uint16 else _capi.lib().upsp_scatter_rows_f32
        _capi.check(fn(
            C.c_void_p(blk.data_ptr()), blk.shape[0], blk.shape[1], C.c_void_p(rows.data_ptr()),
            C.c_void_p(dst.data_ptr()), out.stride(0), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    else:
        out[rows, col:col + blk.shape[1]] = _widen_u16(blk) if blk.dtype == torch.uint16 else blk


def aligned_chunks(nframes, nchunks, align=64):
    """Cuts nframes into nchunks contiguous pieces whose boundaries are multiples of `align`
    (as evenly as that allows; trailing pieces may be empty).  Returns (starts, extents) like
    apportion()."""
    bounds = [min(nframes, int(round(k * nframes / nchunks / align)) * align) for k in range(nchunks)]
    bounds.append(nframes)
    for k in range(1, nchunks + 1):
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds[:-1], [bounds[k + 1] - bounds[k] for k in range(nchunks)]


def chunk_count(frame_counts, limit, align=64):
    """Smallest K such that aligned_chunks(n, K) of every rank's frame count n holds at most `limit` frames per chunk.
    The cuts sit on multiples of `align`, so a chunk can be up to `align` frames longer than n / K (and the last one takes
    the remainder): ceil(n / limit) chunks are NOT enough in general (705 frames, limit 256: three chunks end in one of
    257).  limit >= 2 * align is always reachable."""
    limit = int(limit)
    if limit < 2 * align:
        raise ValueError("chunk limit %d is below two alignment units of %d frames" % (limit, align))
    nmax = max(int(n) for n in frame_counts)
    K = max(1, -(-nmax // limit))
    while max(max(aligned_chunks(int(n), K, align)[1]) for n in frame_counts) > limit:
        K += 1
    return K


class TimeSeriesExchange:
    """global_transpose pipelined with the frame loop: the rank's frames are produced in K
    chunks; the all-to-all of chunk k is issued asynchronously as soon as its node-major
    block exists, so the xGMI transfers overlap the gathers of chunk k+1.  Every rank uses the
    same K; rank s cuts its own frame range with aligned_chunks(frame_count[s], K), so all ranks
    know every block shape without communication.

    set_skipped(): nodes no camera sees hold NaN in every frame (psp_process.cpp:1821-1825) and
    the skipped set is the same on every rank (the projection is replicated), so only the rows
    of the other nodes travel -- on a closed model more than half of the nodes face away from a
    camera, i.e. more than half of the all-to-all bytes are NaNs every rank already knows."""

    def __init__(self, shard, nchunks, dtype=torch.float32, device="cuda", group=None, wire12=False):
        """wire12: u16 chunks are packed to 12 bits for the wire (12-bit cameras: 3 bytes per 2 frames; C-ABI
        exchange only -- a value above 4095 is an error at verify())."""
        self.shard, self.K, self.group = shard, max(1, int(nchunks)), group
        n0, nn = shard.my_nodes
        # this rank's [nodes_r, F] slice; rows start on 256-byte boundaries (tools/pitch_probe.py: the owner's pass B over 12 500
        # frames runs 1-4 % faster at a 50 176-B pitch than at the tight 50 000 B, 9 % at 1000 frames), the view handed out is
        # [nodes_r, F]
        ld = (int(shard.nframes) + 63) // 64 * 64 if (str(device).startswith("cuda") and dtype == torch.float32) else int(shard.nframes)
        self.out = torch.empty((nn, max(ld, 1)), dtype=dtype, device=device)[:, :shard.nframes]
        self.wire12 = bool(wire12)
        # chunk boundaries on multiples of 64 frames (the last chunk takes the remainder): every
        # chunk buffer then has 256-byte-aligned rows, which the gather writes as whole 128-B lines
        self.chunks = [aligned_chunks(shard.frame_count[s], self.K) for s in range(shard.world)]
        self.pending = []
        self.k = 0
        self.vis = None          # indices of the rows that travel (all ranks' slices, ascending)
        self.vis_count = None    # rows that travel per destination rank
        self.vis_mine = None     # their positions inside this rank's slice
        self._row_map = None
        self._keep = None
        self._mismatch = None    # device flag: an assume_same set_skipped() saw a different set
        self._unverified = False # an assume_same claim is pending: verify() has not been called since
        self._x = None           # C-ABI exchange (upsp_exchange_*): RCCL ranks on GPUs
        comm = lib_comm(group) if (str(device).startswith("cuda") and dtype == torch.float32
                                   and (shard.world > 1 or FORCE_COLLECTIVES or not dist.is_initialized())) else None
        if comm is not None:
            import ctypes as C
            from . import _capi
            h = C.c_void_p()
            _capi.check(_capi.lib().upsp_exchange_create(comm, shard.nframes, shard.nnodes, self.K, C.byref(h)))
            self._x = h
            # (columns nframes .. ld of self.out's allocation are padding: the owner's pass B may end its rows on a whole line)
            _capi.check(_capi.lib().upsp_exchange_set_row_padding(h, 0 if os.environ.get("UPSP_EXCHANGE_ROW_PADDING") == "0" else 1))
            self._destroy = _capi.lib().upsp_exchange_destroy
            self._sends = []
            for k in range(self.K):            # the library cuts the chunks itself: both sides must agree
                a, b = C.c_int64(), C.c_int64()
                _capi.check(_capi.lib().upsp_exchange_chunk(h, k, C.byref(a), C.byref(b)))
                assert (a.value, b.value) == self.my_chunk(k), (k, a.value, b.value, self.my_chunk(k))
        elif shard.world > 1 and dist.is_initialized() and not torch_exchange_allowed():
            _need_library("a time-series exchange between %d ranks (%s tensors, %s)" % (shard.world, device, dtype))
        if str(device).startswith("cuda"):
            self._prewarm(device)

    def close(self):
        if getattr(self, "_unverified", False):
            import warnings
            warnings.warn("TimeSeriesExchange: set_skipped(assume_same=True) was used and verify() was never called -- "
                          "a changed travelling set would have gone unnoticed")
            self._unverified = False
        if getattr(self, "_x", None):
            torch.cuda.synchronize()
            self._destroy(self._x)
            self._x = None

    __del__ = close

    @staticmethod
    def _stream():
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _prewarm(device):
        """Runs every torch operator the exchange uses once on tiny tensors, so that their device
        code is loaded before the first measured step (lazy module loading costs tens of ms)."""
        a = torch.zeros(8, dtype=torch.bool, device=device)
        a[::2] = True
        v = torch.nonzero(~a, as_tuple=False).reshape(-1)
        torch.equal(v, v.clone())
        torch.searchsorted(v, torch.tensor([0, 3, 8], device=device)).cpu()
        m = torch.full((8,), -1, dtype=torch.int32, device=device)
        m[v] = torch.arange(v.numel(), dtype=torch.int32, device=device)
        o = torch.empty((8, 4), dtype=torch.float32, device=device)
        mm = torch.ones(8, dtype=torch.bool, device=device)
        mm[v] = False
        torch.nonzero(mm, as_tuple=False)
        o.fill_(float("nan"))
        o[v - 1, 1:3] = torch.ones((v.numel(), 2), device=device)
        o.index_select(0, v)
        bad = (a != a.clone()).any()
        bad = bad | (a != a).any()          # set_skipped(assume_same=True)
        bool(bad)

    def set_skipped(self, skipped, assume_same=False):
        """skipped: bool / uint8 [N] on the exchange's device, identical on every rank; None
        switches back to sending every row.  Deriving the travelling set costs one device->host
        read of W counters (the all-to-all split sizes live on the host).

        assume_same=True: the caller states that the set is the one of the previous call (the
        projection did not change).  Nothing is read back; the claim is checked on the device and
        verify() raises if it was ever wrong."""
        sh = self.shard
        if self._x is not None:
            import ctypes as C
            from . import _capi
            sk = None if skipped is None else skipped.to(torch.uint8).contiguous()
            self._sk = sk                      # (alive until the stream has read it)
            _capi.check(_capi.lib().upsp_exchange_set_skipped(self._x, C.c_void_p(sk.data_ptr() if sk is not None else 0),
                                                              int(bool(assume_same)), self._stream()))
            self._row_map = None
            self.vis = True                    # (the lists live in the library)
            return
        if skipped is None:
            self.vis = self.vis_count = self.vis_mine = self._row_map = self._keep = None
            return
        keep = (skipped == 0) if skipped.dtype != torch.bool else ~skipped
        if assume_same and self._keep is not None and self._keep.shape == keep.shape:
            bad = (self._keep != keep).any()
            self._mismatch = bad if self._mismatch is None else (self._mismatch | bad)
            self._unverified = True            # verify() must be called before the results are trusted
            return
        self._keep = keep.clone()
        self.vis = torch.nonzero(keep, as_tuple=False).reshape(-1)
        self._row_map = None
        bounds = torch.tensor([sh.node_start[d] for d in range(sh.world)] + [sh.nnodes], device=self.vis.device)
        cuts = torch.searchsorted(self.vis, bounds).cpu().tolist()
        self.vis_count = [cuts[d + 1] - cuts[d] for d in range(sh.world)]
        n0, nn = sh.my_nodes
        r = sh.rank
        self.vis_mine = self.vis[cuts[r]:cuts[r + 1]] - n0
        # the rows of this rank's slice that do NOT travel: NaN in every frame, written by every exchange (finish())
        mine = torch.ones(nn, dtype=torch.bool, device=self.vis.device)
        mine[self.vis_mine] = False
        self.nan_mine = torch.nonzero(mine, as_tuple=False).reshape(-1)

    def verify(self):
        """Raises if a set_skipped(..., assume_same=True) call was handed a different set (one host read) -- or, C-ABI
        exchange, if a 12-bit chunk held a value above 4095.  finish() does not check (no host read in the loop): call
        this once after the last pass of a run whenever assume_same was used."""
        if self._x is not None:
            from . import _capi
            _capi.check(_capi.lib().upsp_exchange_verify(self._x, self._stream()))
            self._unverified = False
            return
        self._unverified = False
        if self._mismatch is not None and bool(self._mismatch):
            raise RuntimeError("TimeSeriesExchange: the skipped-node set changed although assume_same was given")
        self._mismatch = None

    def my_chunk(self, k):
        """(local frame offset, frame count) of this rank's chunk k."""
        st, ex = self.chunks[self.shard.rank]
        return st[k], ex[k]

    def row_map(self):
        """int32 [N]: row of the packed chunk buffer for every node that travels, -1 for the rest
        (FramePipeline.set_row_map): the gather then writes the packed rows itself."""
        sh = self.shard
        if self._x is not None:
            if self._row_map is None:
                import ctypes as C
                from . import _capi
                from .engine import _DevArray
                p, n = C.c_void_p(), C.c_int64()
                _capi.check(_capi.lib().upsp_exchange_rows(self._x, C.byref(p), C.byref(n)))
                self._row_map = torch.as_tensor(_DevArray(p.value, sh.nnodes, "<i4", self), device="cuda")
                self._packed = int(n.value)
            return self._row_map
        if self._row_map is None:
            m = torch.full((sh.nnodes,), -1, dtype=torch.int32, device=self.vis.device)
            m[self.vis] = torch.arange(self.vis.numel(), dtype=torch.int32, device=self.vis.device)
            self._row_map = m
        return self._row_map

    def packed_rows(self):
        if self._x is not None:
            self.row_map()
            return self._packed
        return sum(self.vis_count) if self.vis is not None else self.shard.nnodes

    def submit(self, rows_t_chunk, packed=False):
        """rows_t_chunk: [N, fc] (unit column stride), this rank's chunk number len(submitted);
        packed=True: [packed_rows(), fc], already reduced to the travelling rows (row_map())."""
        sh, k = self.shard, self.k
        if k >= self.K:
            raise RuntimeError("TimeSeriesExchange.submit: all %d chunks were already submitted (finish() first)" % self.K)
        if self._x is not None:
            import ctypes as C
            from . import _capi
            if self.vis is None:
                self.set_skipped(None)
            c0, fc = self.my_chunk(k)
            if not packed:                     # whole rows [N, fc]: reduce to the travelling ones
                rows_t_chunk = rows_t_chunk.index_select(0, torch.nonzero(self.row_map() >= 0, as_tuple=False).reshape(-1))
            send = rows_t_chunk.contiguous()
            assert send.shape == (self.packed_rows(), fc), (tuple(send.shape), self.packed_rows(), fc)
            wire = 4 if send.dtype == torch.float32 else (12 if self.wire12 else 2)
            assert send.dtype in (torch.float32, torch.uint16)
            self._sends.append(send)           # untouched until finish
            _capi.check(_capi.lib().upsp_exchange_submit(self._x, C.c_void_p(send.data_ptr()), wire, self._stream()))
            self.k += 1
            return
        c0, fc = self.my_chunk(k)
        assert rows_t_chunk.shape == ((self.packed_rows() if packed else sh.nnodes), fc)
        assert not packed or self.vis is not None
        # u16 chunks (FramePipeline.process with a uint16 rows_t: integer-valued series, half the
        # bytes on the links) exist only packed -- u16 cannot hold the NaN rows
        wire16 = rows_t_chunk.dtype == torch.uint16
        assert not wire16 or packed
        n0, nn = sh.my_nodes
        self.k += 1
        if (sh.world == 1 and not FORCE_COLLECTIVES) or not dist.is_initialized():
            if fc:
                if self.vis is None:
                    self.out[:, c0:c0 + fc] = rows_t_chunk
                else:
                    _scatter_rows(self.out, self.vis_mine, c0,
                                  rows_t_chunk if packed else rows_t_chunk.index_select(0, self.vis))
            return
        if self.vis is None:
            send = rows_t_chunk.contiguous()
            count_out, count_in = sh.node_count, nn
        else:
            # packed rows, ordered by destination
            send = rows_t_chunk.contiguous() if packed else rows_t_chunk.index_select(0, self.vis)
            count_out, count_in = self.vis_count, self.vis_count[sh.rank]
        esz = 2 if wire16 else 1                        # u16 travels as bytes (every backend has uint8)
        in_split = [count_out[d] * fc * esz for d in range(sh.world)]
        out_split = [count_in * self.chunks[s][1][k] * esz for s in range(sh.world)]
        flat = send.reshape(-1).view(torch.uint8) if wire16 else send.reshape(-1)
        recv = torch.empty(sum(out_split), dtype=flat.dtype, device=send.device)
        work = dist.all_to_all_single(recv, flat, out_split, in_split, group=self.group, async_op=True)
        self.pending.append((work, recv, k, send, count_in, wire16))    # keep the send buffer alive

    def finish(self):
        sh = self.shard
        if self._x is not None:
            import ctypes as C
            from . import _capi
            _capi.check(_capi.lib().upsp_exchange_finish(self._x, C.c_void_p(self.out.data_ptr()), self.out.stride(0),
                                                         self._stream()))
            self._sends = []                   # (stream-ordered after the placement of everything received)
            self.k = 0
            return self.out
        for work, recv, k, _, rows_in, wire16 in self.pending:
            work.wait()
            if wire16:
                recv = recv.view(torch.uint16)
            off = 0
            for s in range(sh.world):
                fs = self.chunks[s][1][k]
                if fs and rows_in:
                    col = sh.frame_start[s] + self.chunks[s][0][k]
                    blk = recv[off:off + rows_in * fs].view(rows_in, fs)
                    if self.vis is None:
                        self.out[:, col:col + fs] = blk
                    else:
                        _scatter_rows(self.out, self.vis_mine, col, blk)
                off += rows_in * fs
        self.pending = []
        self.k = 0                      # ready for the next pass over the chunks
        if self.vis is not None:
            _fill_rows(self.out, self.nan_mine, float("nan"))      # rows that do not travel (psp_process.cpp:1821-1825)
        return self.out


def _exchange_bytes(self):
    """(sent, received) bytes that crossed a link in the last finished pass (C-ABI exchange), else None."""
    if self._x is None:
        return None
    import ctypes as C
    from . import _capi
    a, b = C.c_uint64(), C.c_uint64()
    _capi.check(_capi.lib().upsp_exchange_bytes(self._x, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


TimeSeriesExchange.exchange_bytes = _exchange_bytes


# ---- pixel-series mode ---------------------------------------------------------------------------------------------------
# A node's series is the series of the pixel it reads; on a model finer than the pixel grid several nodes share a pixel.
# The ranks then exchange the u16 series of the ACTIVE PIXELS -- every destination gets the pixels its node slice reads,
# each once -- and the owner of a node runs pass B over all frames of the run (upsp_exchange_set_pixels / submit_pixels /
# finish_pixels; here the same bookkeeping on torch.distributed for gloo).  Sender side: FramePipeline.pixel_series
# (pass A + hot-pixel repair) per chunk instead of process().

def _set_pixels(self, node_k, skipped, assume_same=False):
    """node_k int32 [N]: row of the sender's compact buffer per node (< 0: none), identical on every rank (the projection
    is replicated); skipped uint8 / bool [N]."""
    sh = self.shard
    if self._x is not None:
        import ctypes as C
        from . import _capi
        sk = None if skipped is None else skipped.to(torch.uint8).contiguous()
        self._sk, self._nk = sk, node_k
        _capi.check(_capi.lib().upsp_exchange_set_pixels(self._x, C.c_void_p(node_k.data_ptr()),
                                                         C.c_void_p(sk.data_ptr() if sk is not None else 0),
                                                         int(bool(assume_same)), self._stream()))
        if assume_same:
            self._unverified = True
        self.vis = True
        return
    keep = torch.ones(sh.nnodes, dtype=torch.bool, device=node_k.device) if skipped is None else (skipped == 0)
    if assume_same and getattr(self, "_px", None) is not None:
        bad = (self._px["nk"] != node_k).any() | (self._px["keep"] != keep).any()
        self._mismatch = bad if self._mismatch is None else (self._mismatch | bad)
        self._unverified = True
        return
    lists, cut, local = [], [0], None
    for d in range(sh.world):
        n0, nn = sh.node_start[d], sh.node_count[d]
        nk, kp = node_k[n0:n0 + nn].long(), keep[n0:n0 + nn]
        ks = torch.unique(nk[kp & (nk >= 0)])                 # sorted
        lists.append(ks)
        cut.append(cut[-1] + ks.numel())
        if d == sh.rank:
            local = torch.where(kp & (nk >= 0), torch.searchsorted(ks, nk.clamp(min=0)), torch.full_like(nk, -1))
            sk_me = ~kp
    self._px = dict(nk=node_k.clone(), keep=keep.clone(), send_k=torch.cat(lists) if lists else None, cut=cut, local=local,
                    skipped_me=sk_me, compact_me=torch.zeros((cut[sh.rank + 1] - cut[sh.rank], sh.nframes), dtype=torch.int32,
                                                            device=node_k.device))
    self.vis = True


def _pixel_rows(self):
    """(pixel rows this rank sends per chunk, pixel rows it receives)."""
    if self._x is not None:
        import ctypes as C
        from . import _capi
        a, b = C.c_int64(), C.c_int64()
        _capi.check(_capi.lib().upsp_exchange_pixel_rows(self._x, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)
    c = self._px["cut"]
    return c[-1], c[self.shard.rank + 1] - c[self.shard.rank]


def _submit_pixels(self, compact, cpitch=None, col0=0):
    """compact: the sender's [active pixel][>= chunk frames] u16 series of its chunk number len(submitted) -- the dict
    FramePipeline.pixel_series returns (C-ABI exchange), or a tensor (torch.distributed path).  col0: column of the chunk's
    first frame in that buffer (a buffer that holds ALL frames of the rank: pass A ran once for the whole call)."""
    sh, k = self.shard, self.k
    if k >= self.K:
        raise RuntimeError("TimeSeriesExchange.submit_pixels: all %d chunks were already submitted (finish_pixels() first)" % self.K)
    c0, fc = self.my_chunk(k)
    if self._x is not None:
        import ctypes as C
        from . import _capi
        ptr, cp = (compact["ptr"], compact["cpitch"]) if isinstance(compact, dict) else (compact.data_ptr(), compact.stride(0))
        _capi.check(_capi.lib().upsp_exchange_submit_pixels(self._x, C.c_void_p(ptr + 2 * int(col0)), cp, 12 if self.wire12 else 2, self._stream()))
        self.k += 1
        return
    px = self._px
    if isinstance(compact, dict):          # the pipeline's buffer (FramePipeline.pixel_series): a view, no copy
        from .engine import _DevArray
        compact = torch.as_tensor(_DevArray(compact["ptr"], compact["rows"] * compact["cpitch"], "<i2", compact["owner"]),
                                  device="cuda").view(compact["rows"], compact["cpitch"])
    comp = compact.view(torch.int16) if compact.dtype == torch.uint16 else compact
    send = comp.index_select(0, px["send_k"])[:, int(col0):int(col0) + fc].contiguous()   # [sum |L_d|][fc], by destination
    self.k += 1
    cut, r = px["cut"], sh.rank
    rows_in = cut[r + 1] - cut[r]
    if not (dist.is_initialized() and sh.world > 1):
        self.pending.append((None, send, k, send, rows_in, False))
        return
    in_split = [(cut[d + 1] - cut[d]) * fc * 2 for d in range(sh.world)]
    out_split = [rows_in * self.chunks[s][1][k] * 2 for s in range(sh.world)]
    flat = send.reshape(-1).view(torch.uint8)
    recv = torch.empty(sum(out_split), dtype=torch.uint8, device=send.device)
    work = dist.all_to_all_single(recv, flat, out_split, in_split, group=self.group, async_op=True)
    self.pending.append((work, recv, k, send, rows_in, True))


def _finish_pixels(self, total, sumsq):
    """Places what arrived and runs pass B for this rank's nodes over all frames: returns the series [nodes_r, F] (NaN rows
    for the skipped nodes) and ADDS the sums over all frames into total / sumsq [N] at this rank's slice (float64; the other
    ranks' slices are left alone: with zeroed accumulators allreduce_sums then delivers the complete vectors)."""
    sh = self.shard
    n0, nn = sh.my_nodes
    if self._x is not None:
        import ctypes as C
        from . import _capi
        assert total.is_contiguous() and sumsq.is_contiguous() and total.dtype == torch.float64
        _capi.check(_capi.lib().upsp_exchange_finish_pixels(self._x, C.c_void_p(self.out.data_ptr()), self.out.stride(0),
                                                            C.c_void_p(total.data_ptr() + 8 * n0), C.c_void_p(sumsq.data_ptr() + 8 * n0),
                                                            self._stream()))
        self.k = 0
        return self.out
    px = self._px
    cm = px["compact_me"]
    for work, recv, k, _, rows_in, wired in self.pending:
        if work is not None:
            work.wait()
        off = 0
        for s in range(sh.world):
            fs = self.chunks[s][1][k]
            if wired:
                blk = recv[off * 1:off + rows_in * fs * 2].view(torch.int16).view(rows_in, fs) if fs and rows_in else None
                off += rows_in * fs * 2
            else:      # one rank: the send buffer is the block
                blk = recv[:, :fs] if s == sh.rank else None
            if blk is not None and fs and rows_in:
                col = sh.frame_start[s] + self.chunks[s][0][k]
                cm[:, col:col + fs] = blk.to(torch.int32) & 0xFFFF
    self.pending = []
    self.k = 0
    loc = px["local"]
    vals = torch.where((loc >= 0)[:, None], cm.index_select(0, loc.clamp(min=0)), torch.zeros((), dtype=torch.int32, device=cm.device))
    series = vals.to(torch.float32)
    series[px["skipped_me"]] = float("nan")
    self.out.copy_(series)
    d = series.double()
    total[n0:n0 + nn] += d.sum(1)
    sumsq[n0:n0 + nn] += (series * series).double().sum(1)
    return self.out


TimeSeriesExchange.set_pixels = _set_pixels
TimeSeriesExchange.pixel_rows = _pixel_rows
TimeSeriesExchange.submit_pixels = _submit_pixels
TimeSeriesExchange.finish_pixels = _finish_pixels


def gather_time_series_to_root(series, shard, group=None):
    """Single gather of the node-major slices to rank 0: [N, F] on rank 0, None elsewhere.
    Slices are ragged (apportion), so this is one point-to-point transfer per rank
    (grouped isend / irecv -- on RCCL one ncclGroup, every sender on its own xGMI link
    into the root)."""
    if shard.world == 1 or not dist.is_initialized():
        return series
    if shard.rank == 0:
        full = torch.empty((shard.nnodes, shard.nframes), dtype=series.dtype, device=series.device)
        n0, nn = shard.my_nodes
        full[n0:n0 + nn] = series
        ops = [dist.P2POp(dist.irecv, full[shard.node_start[s]:shard.node_start[s] + shard.node_count[s]],
                          s, group) for s in range(1, shard.world) if shard.node_count[s]]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return full
    if shard.node_count[shard.rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, series.contiguous(), 0, group)]):
            w.wait()
    return None


def gather_node_vector(v, shard, group=None):
    """Concatenates the per-rank node slices of a per-node vector on every rank (phase 2's
    rms / avg / gain: the reference MPI_Reduce-sums vectors that are zero outside the owner's
    slice, psp_process.cpp:2521-2527).  Slices differ by at most one node (apportion), so the
    vector is padded to the longest slice for one all_gather."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return v
    longest = max(shard.node_count)
    pad = torch.zeros(longest, dtype=v.dtype, device=v.device)
    pad[:v.numel()] = v
    parts = [torch.empty_like(pad) for _ in range(shard.world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, shard.node_count)])
