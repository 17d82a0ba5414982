"""Row-sharded search over the GPUs of one node (BASELINE.json configs[3]).

Rank r holds rows ``[r N / G, (r+1) N / G)`` of the corpus in its own :class:`TheoremIndex` whose ``row_offset``
makes the returned ids global.  A search is: every rank runs the single-GPU kernels on its shard for the
(replicated) query batch, ONE all-gather exchanges the packed per-shard top-k (``12 * nq * k`` bytes per rank,
latency-bound), and every rank merges the ``G * k`` candidates per query with the device merge kernel.  There is
no other collective on the path.

Two process models, one packed result layout (scores ``[nq x k]`` f32 at offset 0, ids ``[nq x k]`` i64 at
``packed_idx_off``):

* :class:`ShardedSearcher` - one PROCESS per GPU (``torch.distributed.run``).  The collective is either
  ``exchange="native"``: ``ncclAllGather`` inside libtsearch (``ts_comm_*``: RCCL over xGMI; ``torch.distributed``
  only carries the 128-byte unique id once, at start-up), or ``exchange="torch"``:
  ``dist.all_gather_into_tensor`` on the same packed block - device tensors with the nccl backend, host tensors
  with gloo (CPU tests, and rehearsals of several ranks on one GPU).  Device path: queries, per-shard results,
  gathered blocks and merged results all stay in HBM (`search_device`); the exchange + merge of batch i run on a
  side stream and overlap the search of batch i + 1.
* :class:`Shards` - ONE process drives all the GPUs (``ts_shards_*``: ``ncclCommInitAll``, one stream per device).

``local_search`` / ``merge`` can be injected so that the partition + exchange logic runs on CPU ranks with the gloo
backend (tests/test_distributed_cpu.py); the defaults are the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank`` (balanced to within one row)."""
    return n * rank // world, n * (rank + 1) // world


def packed_idx_off(nq: int, k: int) -> int:
    return (nq * k * 4 + 7) // 8 * 8


def packed_bytes(nq: int, k: int) -> int:
    return packed_idx_off(nq, k) + nq * k * 8


def pack_results(scores: np.ndarray, idx: np.ndarray) -> np.ndarray:
    nq, k = scores.shape
    blk = np.zeros(packed_bytes(nq, k), dtype=np.uint8)
    blk[: nq * k * 4] = np.ascontiguousarray(scores, dtype=np.float32).view(np.uint8).reshape(-1)
    off = packed_idx_off(nq, k)
    blk[off: off + nq * k * 8] = np.ascontiguousarray(idx, dtype=np.int64).view(np.uint8).reshape(-1)
    return blk


def unpack_results(blocks: np.ndarray, nparts: int, nq: int, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """``[nparts * blk]`` bytes -> scores ``[nparts x nq x k]`` f32, ids ``[nparts x nq x k]`` i64."""
    blk, off = packed_bytes(nq, k), packed_idx_off(nq, k)
    b = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(nparts, blk)
    scores = np.stack([b[p, : nq * k * 4].copy().view(np.float32).reshape(nq, k) for p in range(nparts)])
    idx = np.stack([b[p, off: off + nq * k * 8].copy().view(np.int64).reshape(nq, k) for p in range(nparts)])
    return scores, idx


class ShardedSearcher:
    """Search over a row-sharded corpus, one process per shard."""

    def __init__(self, local_search: Optional[Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]]] = None,
                 merge: Optional[Callable[[np.ndarray, np.ndarray, int], Tuple[np.ndarray, np.ndarray]]] = None,
                 group=None, index=None, exchange: str = "auto", pipeline: int = 1):
        """``exchange``: ``"torch"`` = ``dist.all_gather_into_tensor`` (RCCL under the nccl backend), ``"native"`` =
        ``ncclAllGather`` inside libtsearch (``ts_comm_*``); ``"auto"`` = ``"torch"``: the native communicator has only ever
        run with one rank on the boxes this was developed on, so it is opt-in until a run on two or more GPUs is on record.
        ``pipeline``: local searches in flight in `search_device` (1 = every search on the caller's stream, the default;
        2 = consecutive searches alternate between the index and a view of it, each on its own stream, so that the small
        kernels at the head of search i + 1 overlap the tail of search i - measured worth 2 % on a 1.25M-row shard and
        nothing on 10M rows: the full pass holds every CU)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.index = index
        self.local_search = local_search or ((lambda q, k: index.search(q, k)) if index is not None else None)
        self._merge_host = merge
        self._comm = C.c_void_p()
        self._bufs = {}
        self._step = 0
        self._side = None
        self._lanes = None
        self.pipeline = max(1, int(pipeline))
        backend = dist.get_backend(group) if dist.is_initialized() else None
        if exchange == "auto":
            exchange = "torch"
        if exchange not in ("native", "torch"):
            raise ValueError(f"exchange must be 'auto', 'native' or 'torch', got {exchange!r}")
        self.exchange = exchange
        self.backend = backend
        if exchange == "native" and self.world > 1:
            ok, why = self._init_native_comm()      # every rank comes back with the same answer
            if not ok:
                import warnings
                warnings.warn(f"native RCCL exchange unavailable ({why}); using torch.distributed for the same block")
                self.close_comm()
                self.exchange = "torch"

    # -- construction ---------------------------------------------------------------------------------------------
    @classmethod
    def from_local_rows(cls, local_rows, total_rows: int, dtype: str = "bf16", metric: str = "cos", device: int = 0,
                        group=None, exchange: str = "auto") -> "ShardedSearcher":
        """Build this rank's shard index from its slice of the corpus (rows ``shard_bounds(...)``)."""
        import torch.distributed as dist
        from .index import TheoremIndex
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(total_rows, world, rank)
        ix = TheoremIndex.from_embeddings(local_rows, dtype=dtype, metric=metric, device=device, row_offset=lo)
        assert ix.n == hi - lo, f"rank {rank} expected {hi - lo} rows, got {ix.n}"
        return cls(group=group, index=ix, exchange=exchange)

    def _agree(self, ok: bool) -> bool:
        """One all-reduce (min) of a start-up outcome: every rank learns whether EVERY rank succeeded."""
        import torch
        dev = torch.device("cuda", self.index.device) if self.backend == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def _init_native_comm(self):
        """ts_comm_create on every rank; torch.distributed only carries the unique id (start-up, not the search path).
        Returns ``(ok, reason)``, the same ``ok`` on every rank.  Every rank walks through the same three collectives
        whatever happens locally (library missing, RCCL not loadable, id refused): nothing local is allowed to raise
        before its outcome has been all-reduced, so no rank can be left waiting in a collective another rank skipped.
        What cannot be recovered: ``ncclCommInitRank`` is itself a collective, and a rank that dies INSIDE it leaves the
        others blocked there - such a job must be ended from outside (the launcher's timeout) and exits non-zero."""
        from . import _ffi
        lib, mine, err = None, None, None
        try:
            _ffi.prefer_torch_rccl()
            lib = _ffi.load()
            buf = C.create_string_buffer(128)
            _ffi.check(lib.ts_comm_unique_id(buf, 128))
            mine = bytes(buf.raw)
        except Exception as e:                     # noqa: BLE001 - reported after every rank has been heard
            err = e
        if not self._agree(mine is not None):                                   # collective 1
            return False, f"RCCL is not usable on every rank (this rank: {err or 'ok'})"
        ident = [mine if self.rank == 0 else None]
        self.dist.broadcast_object_list(ident, src=self.dist.get_global_rank(self.group, 0) if self.group else 0,
                                        group=self.group)                       # collective 2
        try:
            _ffi.check(lib.ts_comm_create(self.index.device, self.world, self.rank, ident[0], 128, C.byref(self._comm)))
        except Exception as e:                     # noqa: BLE001 - a refusal that returned (bad id, no device ...)
            err = e
        if not self._agree(bool(self._comm.value)):                             # collective 3
            return False, f"ts_comm_create failed on some rank (this rank: {err or 'ok'})"
        return True, ""

    def close_comm(self) -> None:
        if self._comm.value:
            from . import _ffi
            _ffi.load().ts_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def close(self) -> None:
        self.close_comm()
        lanes, self._lanes = self._lanes, None
        for ix, _ in (lanes or [])[1:]:           # the views; lane 0 is the caller's index
            try:
                ix.synchronize()
                ix.close()
            except Exception:
                pass
        self._bufs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the device pipeline (what bench.py times) -----------------------------------------------------------------
    _MAX_SHAPES = 4      # (nq, k) shapes whose exchange buffers are kept (least recently used goes first)
    _NBUF = 4            # result blocks in rotation: a search rewrites the block the exchange of four calls ago has read

    def _device_buffers(self, nq: int, k: int):
        import torch
        key = (nq, k)
        b = self._bufs.pop(key, None)
        if b is None:
            dev = torch.device("cuda", self.index.device)
            blk = packed_bytes(nq, k)
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)        # ONE side stream per searcher, whatever the shapes
            while len(self._bufs) >= self._MAX_SHAPES:
                old = self._bufs.pop(next(iter(self._bufs)))
                for e, used in zip(old["done"], old["used"]):
                    if used:
                        e.synchronize()                           # its buffers may still be read by an exchange in flight
            nb = self._NBUF
            b = {
                "mine": [torch.empty(blk, dtype=torch.uint8, device=dev) for _ in range(nb)],
                "all": [torch.empty(self.world * blk, dtype=torch.uint8, device=dev) for _ in range(nb)],
                "fin_s": [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(nb)],
                "fin_i": [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(nb)],
                "ready": [torch.cuda.Event() for _ in range(nb)],
                "qread": [torch.cuda.Event() for _ in range(nb)],
                "qstage": [None] * nb,                            # pipeline > 1: the lane's own copy of the query batch
                "done": [torch.cuda.Event() for _ in range(nb)],
                "used": [False] * nb,
            }
        self._bufs[key] = b                                       # most recently used last
        return b

    def _lane(self, main):
        """(index handle, stream) of the next local search: the caller's index on the caller's stream, or - pipeline > 1 -
        the index and views of it in turn, each on a stream of its own."""
        import torch
        if self.pipeline <= 1:
            return self.index, main, False
        if self._lanes is None:
            dev = torch.device("cuda", self.index.device)
            self._lanes = [(self.index, torch.cuda.Stream(device=dev))]
            for _ in range(self.pipeline - 1):
                self._lanes.append((self.index.view(), torch.cuda.Stream(device=dev)))
        ix, st = self._lanes[self._step % len(self._lanes)]
        return ix, st, True

    def search_device(self, q_ptr: int, q_dtype: str, nq: int, k: int, stream=None, algo: str = "auto", mask_ptr: int = 0,
                      overlap: bool = True):
        """Enqueue one sharded search of ``nq`` device-resident queries, ordered behind ``stream`` (a
        ``torch.cuda.Stream``; default: the current one - the queries must be ready there): the local search (on that
        stream, or with ``pipeline`` > 1 on the stream of the handle whose turn it is), then the exchange + merge on a
        side stream, so that they overlap the next call's search (``overlap=False``: everything on ``stream``).  Returns
        ``(scores, idx, done)``: merged global results as device tensors ``[nq x k]`` (four blocks in rotation: valid
        until the fourth-next call) and the ``torch.cuda.Event`` that marks them complete.  The query buffer may be
        rewritten by work enqueued on ``stream`` after this call returns (with ``pipeline`` > 1 the lane copies the batch into
        a buffer of its own first and the caller's stream waits for that copy, not for the search); work on OTHER streams
        must wait for ``done``.
        Marker packets cost the search's stream a few microseconds each (a 1.25M-row shard's step is 0.5 ms), so the
        search's stream carries none of this class's: the side stream waits on the library's own end-of-call event
        (``ts_index_wait_order``), and the wait for the block's previous reader is skipped when that reader is known
        to be done (``Event.query``)."""
        import torch
        from . import _ffi
        lib = _ffi.load()
        main = stream or torch.cuda.current_stream(self.index.device)
        b = self._device_buffers(nq, k)
        ix, lane, own_stream = self._lane(main) if overlap else (self.index, main, False)
        p = self._step % self._NBUF
        self._step += 1
        if own_stream:
            b["ready"][p].record(main)                 # the queries are complete on the caller's stream
            lane.wait_event(b["ready"][p])
        if b["used"][p] and not b["done"][p].query():
            lane.wait_event(b["done"][p])              # the exchange of four calls ago has not consumed mine[p] yet
        if own_stream:
            # The search may read its queries in place, up to its last launch - and what the caller enqueues next on ITS stream
            # (the next batch's encoder writing the same buffer) must come after that read.  Waiting for the END of the
            # search would serialise the lanes (search i + 1 could not start before search i had finished), so the lane takes
            # its own copy of the batch first (a few hundred KB, device to device) and the caller's stream waits for that
            # copy only: the lanes then really run side by side.
            nbytes = int(nq) * self.index.d * (2 if q_dtype in ("bf16", "bfloat16") else 4)
            if b["qstage"][p] is None or b["qstage"][p].numel() < nbytes:
                b["qstage"][p] = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", self.index.device))
            _ffi.check(lib.ts_copy_device(self.index.device, C.c_void_p(b["qstage"][p].data_ptr()), C.c_void_p(q_ptr), nbytes,
                                          C.c_void_p(lane.cuda_stream)))
            b["qread"][p].record(lane)
            main.wait_event(b["qread"][p])
            q_ptr = b["qstage"][p].data_ptr()
        blk, off = packed_bytes(nq, k), packed_idx_off(nq, k)
        base = b["mine"][p].data_ptr()
        ix.search_device(q_ptr, q_dtype, nq, k, base, base + off, lane.cuda_stream, algo=algo, mask_ptr=mask_ptr)
        side = self._side if overlap else main
        if overlap:
            _ffi.check(lib.ts_index_wait_order(ix.handle, C.c_void_p(side.cuda_stream)))   # behind the search just enqueued
        if self.world == 1:
            src = b["mine"][p]
        else:
            src = b["all"][p]
            if self.exchange == "native":
                _ffi.check(lib.ts_comm_allgather(self._comm, C.c_void_p(base), C.c_void_p(src.data_ptr()), blk,
                                                 C.c_void_p(side.cuda_stream)))
            else:
                with torch.cuda.stream(side):
                    if self.backend == "nccl":
                        self.dist.all_gather_into_tensor(src, b["mine"][p], group=self.group)
                    else:                                     # gloo moves host memory: rehearsal of N ranks on one GPU
                        side.synchronize()
                        host = torch.empty(src.shape, dtype=src.dtype)
                        self.dist.all_gather_into_tensor(host, b["mine"][p].cpu(), group=self.group)
                        src.copy_(host)
        _ffi.check(lib.ts_merge_topk_packed(self.index.device, C.c_void_p(src.data_ptr()), blk, off, self.world, nq, k, k,
                                            C.c_void_p(b["fin_s"][p].data_ptr()), C.c_void_p(b["fin_i"][p].data_ptr()),
                                            C.c_void_p(side.cuda_stream)))
        b["done"][p].record(side)
        b["used"][p] = True
        return b["fin_s"][p], b["fin_i"][p], b["done"][p]

    def measure_exchange(self, nq: int, k: int, iters: int = 50) -> dict:
        """What the collective of `search_device` saw, and what one costs: ``iters`` back-to-back all-gathers of a packed
        block of this shape on the side stream (after 5 untimed ones), timed with events on that stream (gloo: host clock).
        ``world`` is the communicator's own size, ``devices`` one ``(rank, device index, PCI bus id)`` per rank gathered
        through that same communicator - N distinct bus ids are N GPUs.  Collective: every rank must call it."""
        import time
        import torch
        b = self._device_buffers(nq, k)
        blk = packed_bytes(nq, k)
        side = self._side
        torch.cuda.synchronize(self.index.device)
        us = None
        if self.world > 1:
            def one():
                if self.exchange == "native":
                    from . import _ffi
                    _ffi.check(_ffi.load().ts_comm_allgather(self._comm, C.c_void_p(b["mine"][0].data_ptr()),
                                                             C.c_void_p(b["all"][0].data_ptr()), blk, C.c_void_p(side.cuda_stream)))
                elif self.backend == "nccl":
                    with torch.cuda.stream(side):
                        self.dist.all_gather_into_tensor(b["all"][0], b["mine"][0], group=self.group)
                else:
                    host = torch.empty(b["all"][0].shape, dtype=torch.uint8)
                    self.dist.all_gather_into_tensor(host, b["mine"][0].cpu(), group=self.group)
            for _ in range(5):
                one()
            side.synchronize()
            if self.backend == "nccl" or self.exchange == "native":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for _ in range(iters):
                    one()
                e1.record(side)
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
            else:
                t0 = time.perf_counter()
                for _ in range(iters):
                    one()
                us = (time.perf_counter() - t0) * 1e6 / iters
        try:
            bus = torch.cuda.get_device_properties(self.index.device).pci_bus_id
        except Exception:                           # noqa: BLE001 - an older torch without the field
            bus = None
        mine = (self.rank, int(self.index.device), bus)
        devices = [mine]
        if self.world > 1:
            devices = [None] * self.world
            self.dist.all_gather_object(devices, mine, group=self.group)
        return {"backend": self.backend or "none", "world": int(self.dist.get_world_size(self.group)) if self.dist.is_initialized() else 1,
                "native": self.exchange == "native", "collective": "all_gather of the packed per-shard top-k",
                "bytes_per_rank": blk, "allgather_us": None if us is None else round(us, 2), "allgather_iters": iters if us is not None else 0,
                "devices": [list(d_) for d_ in devices],
                "distinct_gpus": len({(d_[1], d_[2]) for d_ in devices})}

    # -- host arrays in, host arrays out -----------------------------------------------------------------------------
    def search(self, queries: np.ndarray, k: int, mask: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """``mask``: bool per row of the WHOLE corpus (a metadata filter); every rank applies its slice of it."""
        import torch
        ix = self.index
        if ix is not None and mask is None and self._merge_host is None:
            # the device pipeline end to end
            q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(-1, ix.d))
            qd = torch.from_numpy(q).to(torch.device("cuda", ix.device))
            s, i, done = self.search_device(qd.data_ptr(), "f32", q.shape[0], int(k), overlap=False)
            done.synchronize()
            return s.cpu().numpy(), i.cpu().numpy()
        if mask is not None:
            lo = ix.row_offset
            scores, idx = ix.search(queries, k, mask=np.asarray(mask, dtype=bool)[lo:lo + ix.n])
        else:
            scores, idx = self.local_search(queries, k)
        if self.world == 1:
            return scores, idx
        nq = scores.shape[0]
        mine = torch.from_numpy(pack_results(scores, idx))
        on_gpu = self.backend == "nccl"
        if on_gpu:
            mine = mine.to(torch.device("cuda", ix.device))
        gathered = torch.empty(self.world * mine.numel(), dtype=torch.uint8, device=mine.device)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)     # ONE exchange, rank-major
        all_scores, all_idx = unpack_results(gathered.cpu().numpy(), self.world, nq, k)
        merge = self._merge_host
        if merge is None:
            from .index import merge_topk
            merge = lambda s_, i_, k_: merge_topk(s_, i_, k_, device=ix.device if ix is not None else 0)
        return merge(all_scores, all_idx, k)

    def rank_of(self, queries: np.ndarray, rows, local_rank_of=None, local_count_above=None) -> np.ndarray:
        """Rank of global row ``rows[i]`` for query ``i`` over the WHOLE sharded corpus (0 = best; -1 = no such row):
        the owning shard scores the row (`TheoremIndex.rank_of`), one all-reduce (max) makes the scores known to every
        rank, every other shard counts its rows that rank before ``(score, row)`` (`count_above`), one all-reduce
        (sum) adds the counts.  The two callables default to this rank's index; they can be injected for CPU tests."""
        import torch
        ix = getattr(self, "index", None)
        local_rank_of = local_rank_of or (lambda q, r: ix.rank_of(q, r))
        local_count_above = local_count_above or (lambda q, s, r: ix.count_above(q, s, r))
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        ranks, scores = local_rank_of(queries, rows)
        if self.world == 1:
            return ranks
        dev = "cuda" if self.dist.get_backend(self.group) == "nccl" else "cpu"
        mine = ranks >= 0
        sc = torch.from_numpy(np.where(mine, scores, -np.inf).astype(np.float32)).to(dev)
        self.dist.all_reduce(sc, op=self.dist.ReduceOp.MAX, group=self.group)  # exactly one shard owns each row
        sc_host = sc.cpu().numpy()
        known = np.isfinite(sc_host)
        counts = np.where(mine, ranks, 0).astype(np.int64)
        others = known & ~mine
        if others.any():
            c = local_count_above(queries[others], sc_host[others], rows[others])
            counts[others] = c
        total = torch.from_numpy(counts).to(dev)
        self.dist.all_reduce(total, op=self.dist.ReduceOp.SUM, group=self.group)
        out = total.cpu().numpy()
        out[~known] = -1
        return out


class Shards:
    """One process, all GPUs of the node: ``ts_shards_*`` (``ncclCommInitAll``, one stream per device, the exchange
    inside libtsearch).  ``devices=None`` uses devices ``0 .. ngpu-1``; repeating a device id puts several shards on one
    GPU (rehearsal on a one-GPU box; the exchange then uses device copies)."""

    def __init__(self, n_total: int, d: int, ngpu: int, dtype: str = "bf16", metric: str = "ip", devices=None):
        from . import _ffi
        from .index import _DTYPES, _METRICS
        _ffi.prefer_torch_rccl()
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        self.n_total, self.d, self.ngpu = int(n_total), int(d), int(ngpu)
        dev = None
        if devices is not None:
            dev = (C.c_int32 * ngpu)(*[int(x) for x in devices])
        _ffi.check(self._lib.ts_shards_create(ngpu, dev, self.n_total, self.d, _DTYPES[dtype], _METRICS[metric],
                                              C.byref(self._h)))
        rc = C.c_int32(0)
        _ffi.check(self._lib.ts_shards_info(self._h, None, None, C.byref(rc)))
        self.uses_rccl = bool(rc.value)

    def upload(self, rows, row0: int = 0) -> None:
        from . import _ffi
        from .index import _host_rows
        rows = _host_rows(rows)
        if rows.shape[1] != self.d:
            raise ValueError(f"rows have d={rows.shape[1]}, corpus has d={self.d}")
        _ffi.check(self._lib.ts_shards_upload(self._h, _ffi.as_ptr(rows), _ffi.np_dtype_code(rows), int(row0), rows.shape[0]))

    def bounds(self, g: int) -> Tuple[int, int]:
        from . import _ffi
        lo, hi = C.c_int64(0), C.c_int64(0)
        _ffi.check(self._lib.ts_shards_shard(self._h, int(g), None, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def search(self, queries, k: int) -> Tuple[np.ndarray, np.ndarray]:
        from . import _ffi
        from .index import _host_rows
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, corpus has d={self.d}")
        scores = np.empty((q.shape[0], int(k)), dtype=np.float32)
        idx = np.empty((q.shape[0], int(k)), dtype=np.int64)
        _ffi.check(self._lib.ts_shards_search(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), q.shape[0], int(k),
                                              _ffi.as_ptr(scores), _ffi.as_ptr(idx)))
        return scores, idx

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ts_shards_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
