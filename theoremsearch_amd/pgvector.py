"""pgvector-shaped adapter: the in-database form of the search (streamlit_app.py:253-283, 317-364).

The production app runs ``ORDER BY e.embedding <#> q ASC LIMIT k`` over stored-normalised vectors and
reports ``similarity = 1.0 - (e.embedding <#> q)``.  ``<#>`` is the NEGATIVE inner product, so that
number is ``1 + <e, q>``; the adapter reproduces it so a row built from its
output is what the SQL returned.  With a citation weight it fetches the pool ``max(50, 10 k)`` and
re-ranks by ``similarity + w * ln(citations)`` (citations > 0), ties by similarity.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .index import TheoremIndex


def _parse_piece(buf: bytes, d: int, cap: int):
    import ctypes as C

    from . import _ffi
    out = np.empty((cap, int(d)), dtype=np.float32)
    rows, used = C.c_int64(0), C.c_int64(0)
    _ffi.check(_ffi.load().ts_parse_pgvector_text(buf, len(buf), int(d), _ffi.as_ptr(out), cap, C.byref(rows), C.byref(used)))
    return out[: rows.value], used.value


def parse_vectors(text, d: int, max_rows: Optional[int] = None, threads: Optional[int] = None):
    """pgvector text rows (``[v1,v2,...]``, anything in between: ids, tabs, newlines) -> ``(fp32 [rows x d], bytes
    consumed)``.  ``text``: bytes or str, e.g. a chunk of ``COPY (SELECT slogan_id, embedding FROM ...) TO STDOUT``;
    a row cut off by the end of the chunk is left for the next call (resume at ``consumed``).  Large buffers are cut
    at row ends and parsed by several threads (the C parser runs without the GIL; one thread does ~130 MB/s)."""
    buf = text.encode() if isinstance(text, str) else bytes(text)
    if max_rows is not None:
        return _parse_piece(buf, d, int(max_rows))
    if threads is None:
        import os
        threads = min(16, os.cpu_count() or 1)
    last = buf.rfind(b"]") + 1                      # the complete rows end here
    nparts = max(1, min(int(threads), last >> 22))  # pieces of at least 4 MiB
    if nparts == 1:
        return _parse_piece(buf, d, buf.count(b"["))
    cuts = [0]
    for i in range(1, nparts):
        c = buf.find(b"]", last * i // nparts, last) + 1
        if c > cuts[-1]:
            cuts.append(c)
    cuts.append(last)
    from concurrent.futures import ThreadPoolExecutor
    pieces = [buf[a:b] for a, b in zip(cuts, cuts[1:]) if b > a]
    with ThreadPoolExecutor(len(pieces)) as ex:
        parts = list(ex.map(lambda piece: _parse_piece(piece, d, piece.count(b"["))[0], pieces))
    return np.concatenate(parts, axis=0), last


def index_from_copy_stream(chunks, n: int, d: int, dtype: str = "f32", metric: str = "ip", device: int = 0) -> TheoremIndex:
    """Build an index of ``n`` rows from an iterable of text chunks of a pgvector COPY / SELECT stream (chunks may cut
    rows anywhere).  ``metric="ip"`` stores the rows as given - the RDS tables hold normalised vectors
    (ec2/generate_embeddings/embeddings.py:27,35) and the app ranks by ``<#>``."""
    ix = TheoremIndex(n, d, dtype=dtype, metric=metric, device=device)
    row0, tail = 0, b""
    for chunk in chunks:
        buf = tail + (chunk.encode() if isinstance(chunk, str) else bytes(chunk))
        rows, used = parse_vectors(buf, d)
        if rows.shape[0]:
            if row0 + rows.shape[0] > n:
                raise ValueError(f"the stream holds more than n = {n} vectors")
            ix.upload(rows, row0)
            row0 += rows.shape[0]
        tail = buf[used:]
    if row0 != n:
        raise ValueError(f"the stream held {row0} vectors, expected {n}")
    return ix


def pool_size(top_k: int) -> int:
    return max(50, int(top_k) * 10)  # streamlit_app.py:317


def citation_bias(citations: Sequence[Optional[int]]) -> np.ndarray:
    """``CASE WHEN citations IS NOT NULL AND citations > 0 THEN ln(citations::float) ELSE 0 END`` per row
    (streamlit_app.py:353-357) as the fp32 side array `TheoremIndex.search_biased` takes."""
    return np.array([math.log(float(c)) if (c is not None and c > 0) else 0.0 for c in citations], dtype=np.float32)


def search(index: TheoremIndex, query_vec, top_k: int, citation_weight: float = 0.0,
           citations: Optional[Sequence[Optional[int]]] = None, mask=None, exact: bool = False, bias=None):
    """Returns a list of dicts ``{"row", "similarity", "score"}`` ordered like the SQL result.  ``mask`` (bool per row,
    e.g. `filters.sql_filter_mask`) plays the WHERE clause: only those rows are ranked.
    ``exact=False`` (default) is the reference's form: the ``max(50, 10 k)`` nearest rows re-ranked by the weighted score.
    ``exact=True`` ranks EVERY (allowed) row by the weighted score on the device (``ts_search_biased``): the same answer
    whenever the pool holds it, and the right one when a heavily cited theorem sits outside the pool; ``bias`` may carry
    a ready-made `citation_bias` array for repeated searches."""
    q = np.asarray(query_vec, dtype=np.float32).reshape(1, -1)
    if citation_weight == 0.0:
        scores, idx = index.search(q, int(top_k), mask=mask)
        return [{"row": int(i), "similarity": 1.0 + float(s), "score": 1.0 + float(s)}
                for s, i in zip(scores[0], idx[0]) if i >= 0]
    if citations is None and bias is None:
        raise ValueError("citation-weighted search needs the per-row citation counts")
    if exact:
        b = citation_bias(citations) if bias is None else np.asarray(bias, dtype=np.float32)
        _, sims, idx = index.search_biased(q, int(top_k), b, float(citation_weight), mask=mask)
        rows = []
        for s, i in zip(sims[0], idx[0]):
            if i < 0:
                continue
            sim = 1.0 + float(s)
            rows.append({"row": int(i), "similarity": sim,
                         "score": sim + citation_weight * float(b[int(i) - index.row_offset])})
        rows.sort(key=lambda r: (-r["score"], -r["similarity"]))
        return rows
    if citations is None:
        raise ValueError("the pool form re-ranks by the citation counts themselves")
    pool = pool_size(top_k)                      # max(50, 10 k): the reference's slider stops at k = 20 -> 200
    if pool > _ffi.TS_MAX_K:
        raise ValueError(f"citation pool of {pool} rows (top_k = {top_k}) exceeds the library's k limit of {_ffi.TS_MAX_K}; "
                         "the reference's UI allows top_k <= 20 (streamlit_app.py:317)")
    scores, idx = index.search(q, pool, mask=mask)
    rows = []
    for s, i in zip(scores[0], idx[0]):
        if i < 0:
            continue
        c = citations[int(i) - index.row_offset]
        sim = 1.0 + float(s)
        bonus = math.log(float(c)) if (c is not None and c > 0) else 0.0
        rows.append({"row": int(i), "similarity": sim, "score": sim + citation_weight * bonus})
    rows.sort(key=lambda r: (-r["score"], -r["similarity"]))
    return rows[: int(top_k)]
