"""TheoremIndex: the [N x d] theorem-embedding matrix resident in HBM, and exact top-k search.

Host-side mirror of the two-line search idiom of the reference apps::

    cosine_scores = util.cos_sim(query_emb, embeddings_db)[0]           # app_showcase_model.py:93
    top = torch.topk(cosine_scores, k=min(200, N), sorted=True)          # app_showcase_model.py:96
    top_indices = np.argsort(-cosine_scores.cpu())[:5]                  # app_scratchpad.py:130

and of ``ORDER BY e.embedding <#> q ASC LIMIT k`` (streamlit_app.py:282-283).  All arithmetic
runs in libtsearch.so's HIP kernels; numpy is only the container of host inputs and results.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _ffi
from ._ffi import TS_ALGO_AUTO, TS_ALGO_MFMA, TS_ALGO_SCAN, TS_BF16, TS_F32, TS_METRIC_COS, TS_METRIC_IP

_DTYPES = {"f32": TS_F32, "fp32": TS_F32, "float32": TS_F32, "bf16": TS_BF16, "bfloat16": TS_BF16}
_METRICS = {"ip": TS_METRIC_IP, "dot": TS_METRIC_IP, "cos": TS_METRIC_COS, "cosine": TS_METRIC_COS}
_ALGOS = {"auto": TS_ALGO_AUTO, "scan": TS_ALGO_SCAN, "mfma": TS_ALGO_MFMA}


def _host_rows(x) -> np.ndarray:
    """numpy float32 / uint16(bf16 bits) 2-D, C-contiguous; accepts lists and CPU torch tensors."""
    if hasattr(x, "detach") and hasattr(x, "cpu"):          # torch.Tensor without importing torch
        t = x.detach().cpu()
        if str(t.dtype) == "torch.bfloat16":
            x = t.view(dtype=__import__("torch").int16).numpy().view(np.uint16)
        else:
            x = t.float().numpy()
    a = np.asarray(x)
    if a.dtype != np.uint16:
        a = a.astype(np.float32, copy=False)
    if a.ndim == 1:
        a = a[None, :]
    if a.ndim != 2:
        raise ValueError("expected a 1-D or 2-D array of embeddings")
    return np.ascontiguousarray(a)


class TheoremIndex:
    """Device-resident embedding matrix with brute-force exact top-k search."""

    _cpu_rows = None          # only the explicit host-only index (device=-1) holds its rows in a numpy array

    def __init__(self, n: int, d: int, dtype: str = "f32", metric: str = "cos", device: int = 0,
                 row_offset: int = 0):
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        self.n, self.d = int(n), int(d)
        self.dtype, self.metric, self.device = dtype, metric, int(device)
        self._cpu_rows = None
        if self.device == -1:
            # The explicit host-only index (BASELINE.json configs[0]: the reference's CPU-runnable plumbing case,
            # compare_embeddings.py:55-92): rows stay in a numpy array, `search` calls ts_search_cpu.  Never chosen by the
            # library itself - a missing GPU still raises for every other device number - and only upload / search / close exist.
            _DTYPES[dtype], _METRICS[metric]                      # validate the names
            self._cpu_rows = np.zeros((self.n, self.d), dtype=np.float32)
            self.row_offset = int(row_offset)
            return
        _ffi.check(self._lib.ts_index_create(self.device, self.n, self.d, _DTYPES[dtype], _METRICS[metric],
                                             C.byref(self._h)))
        self.row_offset = 0
        if row_offset:
            self.set_row_offset(row_offset)

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def from_embeddings(cls, embeddings, dtype: str = "f32", metric: str = "cos", device: int = 0,
                        row_offset: int = 0) -> "TheoremIndex":
        """Build from a host matrix (numpy, list of lists, or the CPU tensor that
        ``torch.load('corpus_embeddings.pt')`` returns, app_showcase_model.py:52)."""
        rows = _host_rows(embeddings)
        ix = cls(rows.shape[0], rows.shape[1], dtype=dtype, metric=metric, device=device, row_offset=row_offset)
        ix.upload(rows, 0)
        return ix

    def view(self) -> "TheoremIndex":
        """A second handle on the same rows (no copy) with its own stream and scratch: searches through the two handles
        can be in flight at once on two streams (serving loops over independent batches).  Read-only; close it before
        the index it views."""
        v = object.__new__(TheoremIndex)
        v._lib, v._h = self._lib, C.c_void_p()
        v.n, v.d, v.dtype, v.metric, v.device, v.row_offset = self.n, self.d, self.dtype, self.metric, self.device, self.row_offset
        v._parent = self                      # keeps the owner of the rows alive
        _ffi.check(self._lib.ts_index_view(self._h, C.byref(v._h)))
        return v

    def subset(self, rows_or_mask) -> "TheoremIndex":
        """A new index over a subset of this one's rows (bool mask of length n, or ascending global row ids).
        Its searches return this index's ids: the filtered search for query batches / long-lived filters."""
        a = np.asarray(rows_or_mask)
        if a.dtype == bool:
            if a.shape[0] != self.n:
                raise ValueError(f"mask has {a.shape[0]} entries, index has {self.n} rows")
            a = np.flatnonzero(a) + self.row_offset
        ids = np.ascontiguousarray(a, dtype=np.int64).reshape(-1)
        sub = object.__new__(TheoremIndex)
        sub._lib, sub._h = self._lib, C.c_void_p()
        sub.n, sub.d, sub.dtype, sub.metric, sub.device, sub.row_offset = int(ids.shape[0]), self.d, self.dtype, self.metric, self.device, 0
        _ffi.check(self._lib.ts_index_subset(self._h, _ffi.as_ptr(ids), ids.shape[0], C.byref(sub._h)))
        return sub

    def upload(self, rows, row0: int = 0) -> None:
        rows = _host_rows(rows)
        if rows.shape[1] != self.d:
            raise ValueError(f"rows have d={rows.shape[1]}, index has d={self.d}")
        if self._cpu_rows is not None:
            if row0 < 0 or row0 + rows.shape[0] > self.n:
                raise ValueError("rows outside the index")
            vals = rows if rows.dtype == np.float32 else (rows.astype(np.uint32) << np.uint32(16)).view(np.float32)
            self._cpu_rows[row0:row0 + rows.shape[0]] = vals
            return
        _ffi.check(self._lib.ts_index_upload(self._h, _ffi.as_ptr(rows), _ffi.np_dtype_code(rows), int(row0),
                                             rows.shape[0]))

    def upload_device(self, dev_ptr: int, src_dtype: str, src_ld: int, row0: int, nrows: int, stream: int = 0) -> None:
        """Rows already in device memory (e.g. ``tensor.data_ptr()`` of the encoder output)."""
        _ffi.check(self._lib.ts_index_upload_device(self._h, C.c_void_p(dev_ptr), _DTYPES[src_dtype], int(src_ld),
                                                    int(row0), int(nrows), C.c_void_p(stream)))

    def reserve(self, capacity: int) -> None:
        """Make room for ``capacity`` rows (no change to ``n``): later `append` calls then never move the rows."""
        _ffi.check(self._lib.ts_index_reserve(self._h, int(capacity)))

    def append(self, rows) -> int:
        """Add rows behind the last one (new ``slogan_id``s of the upsert pipeline, ec2/generate_embeddings/__main__.py:85-99);
        returns the global id of the first new row.  The allocation grows 1.5x when it is full."""
        rows = _host_rows(rows)
        if rows.shape[1] != self.d:
            raise ValueError(f"rows have d={rows.shape[1]}, index has d={self.d}")
        first = C.c_int64(-1)
        _ffi.check(self._lib.ts_index_append(self._h, _ffi.as_ptr(rows), _ffi.np_dtype_code(rows), rows.shape[0],
                                             C.byref(first)))
        self.n += rows.shape[0]
        return first.value

    def append_device(self, dev_ptr: int, src_dtype: str, src_ld: int, nrows: int, stream: int = 0) -> int:
        """`append` for rows already in device memory (the encoder output)."""
        first = C.c_int64(-1)
        _ffi.check(self._lib.ts_index_append_device(self._h, C.c_void_p(dev_ptr), _DTYPES[src_dtype], int(src_ld), int(nrows),
                                                    C.c_void_p(stream), C.byref(first)))
        self.n += int(nrows)
        return first.value

    def attach_device(self, dev_ptr: int, capacity_rows: int, keep_alive=None) -> None:
        """Adopt rows already in device memory (zero-copy; e.g. ``tensor.data_ptr()`` of an encoder output in the index's
        dtype with ``capacity_rows >= n`` rounded up to 256).  ``keep_alive``: an object to hold on to (the tensor)."""
        _ffi.check(self._lib.ts_index_attach_device(self._h, C.c_void_p(dev_ptr), int(capacity_rows)))
        self._attached = keep_alive

    def set_row_offset(self, offset: int) -> None:
        _ffi.check(self._lib.ts_index_set_row_offset(self._h, int(offset)))
        self.row_offset = int(offset)

    def set_option(self, name: str, value: Optional[int]) -> None:
        """Tuning / diagnostic knob of this handle (``"TS_MFMA_FIRST_ROWS"`` ...); ``None`` restores the default.
        The knobs' initial values come from the environment when the handle is created."""
        if value is None:
            _ffi.check(self._lib.ts_index_reset_option(self._h, name.encode()))
        else:
            _ffi.check(self._lib.ts_index_set_option(self._h, name.encode(), int(value)))

    def download(self, row0: int = 0, nrows: Optional[int] = None) -> np.ndarray:
        """Stored rows (normalised / bf16-rounded as the kernels see them)."""
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty((nrows, self.d), dtype=np.uint16 if _DTYPES[self.dtype] == TS_BF16 else np.float32)
        _ffi.check(self._lib.ts_index_download(self._h, _ffi.as_ptr(out), int(row0), int(nrows)))
        return out

    def synchronize(self) -> None:
        """Wait for everything this handle has enqueued (uploads / searches with device buffers return early)."""
        _ffi.check(self._lib.ts_index_synchronize(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _ffi.check(self._lib.ts_index_stream(self._h, C.byref(s)))
        return s.value or 0

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def info(self) -> dict:
        """Shape and placement of the rows as the kernels see them (``ts_index_info``): ``ld`` = row stride in
        elements, ``rows_ptr`` = device address of row 0."""
        n, ld, off = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        d, dt, me = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        rows = C.c_void_p()
        _ffi.check(self._lib.ts_index_info(self._h, C.byref(n), C.byref(d), C.byref(dt), C.byref(me), C.byref(ld), C.byref(off),
                                           C.byref(rows)))
        return {"n": n.value, "d": d.value, "dtype": dt.value, "metric": me.value, "ld": ld.value, "row_offset": off.value,
                "rows_ptr": rows.value or 0}

    # -- search -----------------------------------------------------------------------------
    def search(self, queries, k: int, algo: str = "auto", return_stats: bool = False, mask=None):
        """Exact top-k.  Returns ``(scores [nq x k] float32, indices [nq x k] int64)`` ordered by
        score descending then index ascending; padding is ``(-inf, -1)``.  ``mask`` (bool per row)
        restricts the search to the rows where it is true (metadata filters): the k best allowed rows."""
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, index has d={self.d}")
        nq, k = q.shape[0], int(k)
        scores = np.empty((nq, k), dtype=np.float32)
        idx = np.empty((nq, k), dtype=np.int64)
        if self._cpu_rows is not None:
            if mask is not None or algo != "auto":
                raise ValueError("the host-only index (device=-1) has one algorithm and no filters")
            _ffi.check(self._lib.ts_search_cpu(_ffi.as_ptr(self._cpu_rows), TS_F32, self.n, self.d, _DTYPES[self.dtype],
                                               _METRICS[self.metric], _ffi.as_ptr(q), _ffi.np_dtype_code(q), nq, k,
                                               _ffi.as_ptr(scores), _ffi.as_ptr(idx), 0))
            if self.row_offset:
                idx[idx >= 0] += self.row_offset
            return (scores, idx, {"algo": 0, "levels": 0, "fallback_queries": 0, "candidates": 0}) if return_stats else (scores, idx)
        if mask is not None:
            m = np.asarray(mask, dtype=bool).reshape(-1)
            if m.shape[0] != self.n:
                raise ValueError(f"mask has {m.shape[0]} entries, index has {self.n} rows")
            bits = np.packbits(m, bitorder="little")
            words = np.zeros((self.n + 31) // 32 * 4, dtype=np.uint8)
            words[: bits.shape[0]] = bits
            words = words.view(np.uint32)
            stats = _ffi.SearchStats()
            _ffi.check(self._lib.ts_search_filtered_ex(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, nq, k,
                                                       _ffi.as_ptr(words), 0, _ffi.as_ptr(scores), _ffi.as_ptr(idx), 0, None,
                                                       _ALGOS[algo], C.byref(stats)))
            if return_stats:
                return scores, idx, {"algo": stats.algo, "levels": stats.levels,
                                     "fallback_queries": stats.fallback_queries, "candidates": stats.candidates}
            return scores, idx
        stats = _ffi.SearchStats()
        _ffi.check(self._lib.ts_search_ex(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, nq, k,
                                          _ffi.as_ptr(scores), _ffi.as_ptr(idx), 0, None, _ALGOS[algo],
                                          C.byref(stats)))
        if return_stats:
            return scores, idx, {"algo": stats.algo, "levels": stats.levels,
                                 "fallback_queries": stats.fallback_queries, "candidates": stats.candidates}
        return scores, idx

    def search_device(self, q_ptr: int, q_dtype: str, nq: int, k: int, out_scores_ptr: int, out_idx_ptr: int,
                      stream: int = 0, algo: str = "auto", mask_ptr: int = 0) -> None:
        """Asynchronous search on device buffers (queries [nq x d] dense; outputs [nq x k] f32 / i64),
        enqueued on ``stream`` (0 = the index's own stream).  ``mask_ptr``: device uint32 bitmask
        (ceil(n / 32) words) restricting the rows, see `search`.
        Queries that already have the form the matrix kernels multiply (the index's dtype, an inner-product index, a
        whole launch's worth: 64 / 128 / 192 / 256 of them) are read in place - keep them unchanged until the enqueued work
        has run (work enqueued later on the same stream is ordered behind it anyway)."""
        if mask_ptr:
            _ffi.check(self._lib.ts_search_filtered(self._h, C.c_void_p(q_ptr), _DTYPES[q_dtype], 1, int(nq), int(k),
                                                    C.c_void_p(mask_ptr), 1, C.c_void_p(out_scores_ptr),
                                                    C.c_void_p(out_idx_ptr), 1, C.c_void_p(stream)))
            return
        _ffi.check(self._lib.ts_search_ex(self._h, C.c_void_p(q_ptr), _DTYPES[q_dtype], 1, int(nq), int(k),
                                          C.c_void_p(out_scores_ptr), C.c_void_p(out_idx_ptr), 1,
                                          C.c_void_p(stream), _ALGOS[algo], None))

    def search_biased(self, queries, k: int, bias, weight: float, mask=None):
        """Top-k of ``score + weight * bias[row]`` over all rows (all rows ``mask`` allows): the citation-weighted
        ranking of streamlit_app.py:348-364 without its candidate pool.  ``bias``: float32 per row of this index.
        Returns ``(weighted scores, raw similarities, indices)``, each ``[nq x k]``."""
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, index has d={self.d}")
        b = np.ascontiguousarray(np.asarray(bias, dtype=np.float32).reshape(-1))
        if b.shape[0] != self.n:
            raise ValueError(f"bias has {b.shape[0]} entries, index has {self.n} rows")
        nq, k = q.shape[0], int(k)
        scores = np.empty((nq, k), dtype=np.float32)
        sims = np.empty((nq, k), dtype=np.float32)
        idx = np.empty((nq, k), dtype=np.int64)
        words = None
        if mask is not None:
            m = np.asarray(mask, dtype=bool).reshape(-1)
            if m.shape[0] != self.n:
                raise ValueError(f"mask has {m.shape[0]} entries, index has {self.n} rows")
            bits = np.packbits(m, bitorder="little")
            words = np.zeros((self.n + 31) // 32 * 4, dtype=np.uint8)
            words[: bits.shape[0]] = bits
        _ffi.check(self._lib.ts_search_biased(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, nq, k, _ffi.as_ptr(b), 0,
                                              float(weight), _ffi.as_ptr(words) if words is not None else None, 0,
                                              _ffi.as_ptr(scores), _ffi.as_ptr(sims), _ffi.as_ptr(idx), 0, None))
        return scores, sims, idx

    def rank_of(self, queries, rows) -> Tuple[np.ndarray, np.ndarray]:
        """0-based rank of ``rows[i]`` among all index rows for query ``i`` (score descending, index ascending)
        and its score: what ``np.flatnonzero(np.argsort(-sim[i]) == rows[i])`` yields on the full score matrix
        (compare_embeddings.py:96-123), computed by one counting pass.  ``-1`` / NaN for rows not in the index."""
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, index has d={self.d}")
        t = np.ascontiguousarray(np.asarray(rows, dtype=np.int64).reshape(-1))
        if t.shape[0] != q.shape[0]:
            raise ValueError("one target row per query")
        ranks = np.empty(q.shape[0], dtype=np.int64)
        scores = np.empty(q.shape[0], dtype=np.float32)
        _ffi.check(self._lib.ts_rank_of(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, q.shape[0], _ffi.as_ptr(t),
                                        _ffi.as_ptr(ranks), _ffi.as_ptr(scores), None))
        return ranks, scores

    def count_above(self, queries, target_scores, target_ids) -> np.ndarray:
        """Rows of this index that rank before a document with the given score and GLOBAL id (it may live on another
        shard); summed over the shards of a corpus this is the document's rank (`rank_of` for one index)."""
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, index has d={self.d}")
        sc = np.ascontiguousarray(np.asarray(target_scores, dtype=np.float32).reshape(-1))
        ids = np.ascontiguousarray(np.asarray(target_ids, dtype=np.int64).reshape(-1))
        if sc.shape[0] != q.shape[0] or ids.shape[0] != q.shape[0]:
            raise ValueError("one target per query")
        out = np.empty(q.shape[0], dtype=np.int64)
        _ffi.check(self._lib.ts_count_above(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, q.shape[0], _ffi.as_ptr(sc),
                                            _ffi.as_ptr(ids), _ffi.as_ptr(out), None))
        return out

    def scores(self, queries) -> np.ndarray:
        """Full ``[nq x N]`` fp32 score matrix (small N): ``util.cos_sim(q_emb, s_emb)`` of
        compare_embeddings.py:24,61 when the index metric is "cos"."""
        q = _host_rows(queries)
        if q.shape[1] != self.d:
            raise ValueError(f"queries have d={q.shape[1]}, index has d={self.d}")
        out = np.empty((q.shape[0], self.n), dtype=np.float32)
        _ffi.check(self._lib.ts_scores(self._h, _ffi.as_ptr(q), _ffi.np_dtype_code(q), 0, q.shape[0],
                                       _ffi.as_ptr(out), 0, None))
        return out

    # -- profiling ---------------------------------------------------------------------------
    def profile_enable(self, enable: bool = True) -> None:
        _ffi.check(self._lib.ts_index_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self) -> dict:
        """Launch count and summed hipEvent duration of the dominant kernel since the last read."""
        n, ms, rows = C.c_int64(0), C.c_double(0.0), C.c_int64(0)
        _ffi.check(self._lib.ts_index_profile_read(self._h, C.byref(n), C.byref(ms), C.byref(rows)))
        return {"launches": n.value, "total_ms": ms.value, "rows_per_launch": rows.value}

    def probe_read(self) -> dict:
        """Last in-kernel clock probe of the MFMA full pass (option ``TS_MFMA_VARIANT`` = 3)."""
        g, c, u = C.c_double(0), C.c_double(0), C.c_double(0)
        _ffi.check(self._lib.ts_index_probe_read(self._h, C.byref(g), C.byref(c), C.byref(u)))
        return {"ghz": g.value, "cycles_per_unit": c.value, "units_per_workgroup": u.value}

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        self._cpu_rows = None
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.check(self._lib.ts_index_destroy(self._h))      # refused while views of this index are alive
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


def merge_topk(scores: np.ndarray, idx: np.ndarray, k_out: int, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Merge ``[nparts x nq x k_in]`` partial results into the global top-``k_out`` on the device."""
    s = np.ascontiguousarray(scores, dtype=np.float32)
    i = np.ascontiguousarray(idx, dtype=np.int64)
    nparts, nq, k_in = s.shape
    os_ = np.empty((nq, k_out), dtype=np.float32)
    oi = np.empty((nq, k_out), dtype=np.int64)
    _ffi.check(_ffi.load().ts_merge_topk(device, _ffi.as_ptr(s), _ffi.as_ptr(i), nparts, nq, k_in, k_out,
                                         _ffi.as_ptr(os_), _ffi.as_ptr(oi), 0, None))
    return os_, oi


class Timer:
    """hipEvent pair on a given stream (ts_timer_*)."""

    def __init__(self, device: int = 0):
        self._lib = _ffi.load()
        self._h = C.c_void_p()
        _ffi.check(self._lib.ts_timer_create(device, C.byref(self._h)))

    def start(self, stream: int = 0):
        _ffi.check(self._lib.ts_timer_start(self._h, C.c_void_p(stream)))

    def stop(self, stream: int = 0):
        _ffi.check(self._lib.ts_timer_stop(self._h, C.c_void_p(stream)))

    def elapsed_ms(self) -> float:
        ms = C.c_float(0)
        _ffi.check(self._lib.ts_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self._h.value:
                self._lib.ts_timer_destroy(self._h)
        except Exception:
            pass
