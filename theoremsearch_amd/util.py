"""Drop-in for the two ``sentence_transformers.util`` / selection idioms on the reference's hot path.

    util.cos_sim(a, b)                               compare_embeddings.py:24,61, app_showcase_model.py:93
    torch.topk(scores, k, sorted=True) / argsort     app_showcase_model.py:96, app_scratchpad.py:130

``cos_sim`` returns the full ``[B x N]`` matrix and is meant for the small shapes of the evaluation
script; the apps' "scores then top-k" pair is one fused call here (``semantic_search``), which never
materialises the scores.  Everything runs in libtsearch.so on the GPU; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .index import TheoremIndex, _host_rows


def cos_sim(a, b, dtype: str = "f32") -> np.ndarray:
    """``[len(a) x len(b)]`` fp32 cosine matrix (numpy).  1-D inputs are promoted to one row, as
    ``util.cos_sim`` does.  Builds a throw-away index over ``b``; keep a :class:`TheoremIndex`
    when the corpus is reused."""
    a, b = _host_rows(a), _host_rows(b)
    if a.shape[1] != b.shape[1]:
        raise ValueError(f"dimension mismatch: {a.shape[1]} vs {b.shape[1]}")
    with TheoremIndex.from_embeddings(b, dtype=dtype, metric="cos") as ix:
        return ix.scores(a)


def semantic_search(query_embeddings, corpus, top_k: int = 10, dtype: str = "f32") -> Tuple[np.ndarray, np.ndarray]:
    """Fused ``cos_sim`` + ``topk``: ``(scores [B x k], indices [B x k])``, score descending then
    index ascending.  ``corpus`` is a :class:`TheoremIndex` (reused) or a host matrix (indexed once)."""
    if isinstance(corpus, TheoremIndex):
        return corpus.search(query_embeddings, top_k)
    with TheoremIndex.from_embeddings(corpus, dtype=dtype, metric="cos") as ix:
        return ix.search(query_embeddings, top_k)


def topk(scores, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host-side ``torch.topk(scores, k, sorted=True)`` for an already materialised score row or
    matrix: values and indices, score descending then index ascending, NaN never selected.
    (Result post-processing only; the search itself is :func:`semantic_search`.)"""
    s = np.asarray(scores, dtype=np.float32)
    one = s.ndim == 1
    s2 = s[None, :] if one else s
    kk = min(int(k), s2.shape[1])
    key = np.where(np.isnan(s2), -np.inf, s2)
    part = np.argpartition(-key, kk - 1, axis=1)[:, :kk] if kk < s2.shape[1] else np.broadcast_to(
        np.arange(s2.shape[1]), s2.shape).copy()
    vals = np.take_along_axis(key, part, axis=1)
    order = np.lexsort((part, -vals), axis=1)
    idx = np.take_along_axis(part, order, axis=1).astype(np.int64)
    vals = np.take_along_axis(vals, order, axis=1)
    # ties that straddle the partition boundary: prefer the lowest index among equal scores (rows where the k-th score
    # occurs more often in the row than among the selected are found in one vector pass; usually none)
    kth = vals[:, -1:]
    straddle = np.flatnonzero((key == kth).sum(axis=1) > (vals == kth).sum(axis=1))
    for b in straddle:
        tied = np.flatnonzero(key[b] == kth[b, 0])
        keep = vals[b] > kth[b, 0]
        need = kk - int(keep.sum())
        idx[b] = np.concatenate([idx[b][keep], tied[:need]])
        vals[b] = key[b][idx[b]]
    return (vals[0], idx[0]) if one else (vals, idx)
