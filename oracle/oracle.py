"""CPU oracle for the TheoremSearch embedding-similarity + top-k hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``theoremsearch_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker.

This is a numpy restatement of the arithmetic the reference delegates to
third-party libraries on its hot path:

* ``sentence_transformers.util.cos_sim`` (unpinned in the reference's
  ``requirements.txt:9`` / ``ec2/requirements.txt:4``): promote to 2-D, divide
  every row by ``max(||row||_2, 1e-12)``, then ``a_n @ b_n.T`` in fp32.  Call
  sites: ``compare_embeddings.py:24,61``, ``app_showcase_model.py:93``,
  ``app_scratchpad.py:129``, ``test_app.py:76``.
* selection: ``np.argsort(-scores)[:k]`` (``app_scratchpad.py:130``,
  ``compare_embeddings.py:52``), ``sims.argsort()[::-1][:top_k]``
  (``compare_embeddings.py:31``), ``torch.topk(scores, k, sorted=True)``
  (``app_showcase_model.py:96``): the k largest scores, descending.  The
  reference's tie order is unspecified, so the oracle fixes the canonical order
  (score descending, then index ascending) and NaN scores are never returned.
* pgvector ``<#>`` (negative inner product, fp32) with ``ORDER BY .. ASC LIMIT k``
  (``streamlit_app.py:253-283``) and the citation-weighted re-rank
  (``streamlit_app.py:317-364``).
* the six retrieval metrics of ``compare_embeddings.py:95-371``.

Pinning status: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4).  The metric restatements are pinned against outputs of the
reference's own function bodies executed in the authoring container
(``oracle/gen_golden.py`` -> ``tests/golden/metrics_*.json``).  ``cos_sim`` and
the selection primitives live in dependencies that are absent from
``/root/reference``; they are pinned against the published algorithm executed
with torch-CPU (``F.normalize`` + ``mm``, ``torch.topk``, ``np.argsort``) by
the same script (``tests/golden/search_*.npz``).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

EPS = np.float32(1e-12)

# ----------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even, NaN kept a NaN)
# ----------------------------------------------------------------------------

def _threads() -> int:
    import os
    try:
        return max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(16, os.cpu_count() or 1))


def _by_row_blocks(fn, x: np.ndarray, out: np.ndarray, rows: int = 16384) -> np.ndarray:
    """``out[r0:r1] = fn(x[r0:r1])`` over blocks of rows, on a few threads for large inputs (numpy's elementwise loops
    release the GIL; a 2.5M x 768 corpus is ten passes over 7.7 GB otherwise)."""
    n = x.shape[0] if x.ndim >= 1 else 0
    if x.ndim < 2 or x.size < (1 << 22):
        out[...] = fn(x)
        return out
    blocks = [(r0, min(n, r0 + rows)) for r0 in range(0, n, rows)]

    def one(b):
        out[b[0]:b[1]] = fn(x[b[0]:b[1]])

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(_threads()) as ex:
        list(ex.map(one, blocks))
    return out


def _bf16_bits_block(x: np.ndarray) -> np.ndarray:
    u = x.view(np.uint32)
    rounded = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    nan = np.isnan(x)
    if nan.any():
        rounded = np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x0040)).astype(np.uint16), rounded)
    return rounded


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit patterns (uint16), round-to-nearest-even; NaN stays NaN."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return _by_row_blocks(_bf16_bits_block, x, np.empty(x.shape, dtype=np.uint16))


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return _by_row_blocks(lambda v: (v.astype(np.uint32) << np.uint32(16)).view(np.float32), b, np.empty(b.shape, dtype=np.float32))


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bf16, returned as fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return _by_row_blocks(lambda v: (_bf16_bits_block(v).astype(np.uint32) << np.uint32(16)).view(np.float32), x,
                          np.empty(x.shape, dtype=np.float32))


# ----------------------------------------------------------------------------
# cos_sim / inner product  (sentence_transformers.util.cos_sim semantics)
# ----------------------------------------------------------------------------

def _as_2d(x) -> np.ndarray:
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[None, :]
    if x.ndim != 2:
        raise ValueError("expected a 1-D or 2-D array")
    return x


def l2_normalize(x: np.ndarray) -> np.ndarray:
    """Row-wise ``x / max(||x||_2, 1e-12)`` in fp32 (``F.normalize(p=2, dim=1)``).

    The squared norm is accumulated in fp64 and rounded once to fp32; the
    reference's accumulation order is unspecified (it is torch's), so the
    oracle fixes this one.  A zero row stays zero.
    """
    x = _as_2d(x).astype(np.float32, copy=False)

    def block(v):
        ss = np.einsum("ij,ij->i", v, v, dtype=np.float64)
        denom = np.maximum(np.sqrt(ss).astype(np.float32), EPS)
        return (v / denom[:, None]).astype(np.float32)

    return _by_row_blocks(block, x, np.empty(x.shape, dtype=np.float32))


def dot_scores(q: np.ndarray, c: np.ndarray) -> np.ndarray:
    """``q @ c.T`` in fp32 -> [B x N] (inner product; pgvector ``-(a <#> b)``)."""
    q = _as_2d(q).astype(np.float32, copy=False)
    c = _as_2d(c).astype(np.float32, copy=False)
    return (q @ c.T).astype(np.float32)


def cos_sim(a, b) -> np.ndarray:
    """Restatement of ``util.cos_sim(a, b)``: [B x N] fp32 cosine matrix."""
    return dot_scores(l2_normalize(a), l2_normalize(b))


def scores_fp64(q: np.ndarray, c: np.ndarray) -> np.ndarray:
    """fp64 "truth" inner products of the same (possibly bf16-rounded) inputs."""
    q64, c = _as_2d(q).astype(np.float64), _as_2d(c)
    if c.size < (1 << 22):
        return q64 @ c.astype(np.float64).T
    out = np.empty((q64.shape[0], c.shape[0]), dtype=np.float64)       # block by block: no fp64 copy of the whole corpus
    for r0 in range(0, c.shape[0], 65536):
        out[:, r0:r0 + 65536] = q64 @ c[r0:r0 + 65536].astype(np.float64).T
    return out


# ----------------------------------------------------------------------------
# selection
# ----------------------------------------------------------------------------

def topk_canonical(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """k largest per row, ordered (score desc, index asc); NaN never selected.

    Returns ``(values [B x k] same dtype as scores, indices [B x k] int64)``;
    rows with fewer than k non-NaN scores are padded with (-inf, -1).
    Equivalent to ``np.argsort(-s, kind="stable")[:k]`` on NaN-free input
    (``app_scratchpad.py:130``, ``compare_embeddings.py:52``).
    """
    s = _as_2d(scores)
    B, N = s.shape
    k = int(k)
    vals = np.full((B, k), -np.inf, dtype=s.dtype)
    idx = np.full((B, k), -1, dtype=np.int64)
    for b in range(B):
        row = s[b]
        valid = np.flatnonzero(~np.isnan(row))
        if valid.size == 0:
            continue
        r = row[valid]
        kk = min(k, valid.size)
        if valid.size > 4 * kk + 64:
            # partition first, but keep every element tied with the k-th value
            part = np.argpartition(-r, kk - 1)[:kk]
            kth = r[part].min()
            cand = np.flatnonzero(r >= kth)
        else:
            cand = np.arange(valid.size)
        order = np.lexsort((valid[cand], -r[cand]))[:kk]
        sel = cand[order]
        vals[b, :kk] = r[sel]
        idx[b, :kk] = valid[sel]
    return vals, idx


def search(q: np.ndarray, c: np.ndarray, k: int, metric: str = "ip",
           dtype: str = "f32") -> Tuple[np.ndarray, np.ndarray]:
    """Oracle for one brute-force search call.

    metric "cos": both sides L2-normalised first (``util.cos_sim``);
    metric "ip": inputs used as they are (pgvector ``<#>`` on stored-normalised
    rows, ``ec2/generate_embeddings/embeddings.py:27,35``).
    dtype "bf16": corpus rows and queries are rounded to bf16 (after the
    normalisation) and the product is taken in fp32 on the rounded values.
    """
    q = _as_2d(q).astype(np.float32, copy=False)
    c = _as_2d(c).astype(np.float32, copy=False)
    if metric == "cos":
        q, c = l2_normalize(q), l2_normalize(c)
    elif metric != "ip":
        raise ValueError(metric)
    if dtype == "bf16":
        q, c = round_to_bf16(q), round_to_bf16(c)
    elif dtype != "f32":
        raise ValueError(dtype)
    return topk_canonical(dot_scores(q, c), k)


def prepared_inputs(q, c, metric="ip", dtype="f32"):
    """The exact operand values ``search`` multiplies (for fp64 truth checks)."""
    q = _as_2d(q).astype(np.float32, copy=False)
    c = _as_2d(c).astype(np.float32, copy=False)
    if metric == "cos":
        q, c = l2_normalize(q), l2_normalize(c)
    if dtype == "bf16":
        q, c = round_to_bf16(q), round_to_bf16(c)
    return q, c


# ----------------------------------------------------------------------------
# parity protocol ("Hard parts", first item)
# ----------------------------------------------------------------------------

def _check_one_query(b, ext, sv, score_of, got, got_scores_row, kk, k, gap, score_tol):
    """One query of the protocol.  ``ext`` / ``sv``: indices and fp64 scores of the best rows in canonical order (at least
    kk + 1 of them unless the corpus is smaller); ``score_of(j)``: fp64 score of row j.  Returns (pinned, recall hits)."""
    pinned = 0
    assert len(set(got.tolist())) == kk, f"query {b}: duplicate indices {got}"
    pos = 0
    while pos < kk:
        end = pos + 1
        while end < len(ext) and sv[end - 1] - sv[end] <= gap:
            end += 1
        members = set(int(x) for x in ext[pos:end])
        take = min(end, kk) - pos
        got_run = [int(x) for x in got[pos:pos + take]]
        if end - pos == 1:
            pinned += 1
            assert got_run[0] == ext[pos], (
                f"query {b} rank {pos}: got {got_run[0]} want {ext[pos]} "
                f"(fp64 scores {score_of(got_run[0])!r} vs {sv[pos]!r})")
        else:
            assert set(got_run) <= members, (
                f"query {b} ranks {pos}..{pos + take}: {got_run} not within tie run {sorted(members)}")
        pos = end
    true_set = set(int(x) for x in ext[:kk])
    kth = sv[kk - 1]
    hits = 0
    ref = np.array([score_of(int(j)) for j in got], dtype=np.float64)
    for j, r in zip(got.tolist(), ref):
        if j in true_set or r >= kth - gap:
            hits += 1
    if got_scores_row is not None:
        gs = np.asarray(got_scores_row, dtype=np.float64)[:kk]
        assert np.all(np.abs(gs - ref) <= score_tol), f"query {b}: scores off by {np.max(np.abs(gs - ref))}"
    return pinned, hits


def check_topk_against_truth(truth_scores: np.ndarray, got_idx: np.ndarray,
                             got_scores: Optional[np.ndarray], k: int,
                             gap: float = 1e-6, score_tol: float = 1e-5) -> Dict[str, float]:
    """Compare a top-k answer with fp64 truth scores of the same inputs.

    A rank position is *pinned* when its fp64 gaps to both neighbours (and, for
    position k-1, to the (k+1)-th score) exceed ``gap``: there the index must
    match exactly.  Inside an unpinned run (scores closer than ``gap``) indices
    must match as a set; the run that straddles the k boundary may pick any of
    its members.  Scores must be within ``score_tol`` of truth at the returned
    index.  Raises AssertionError on the first violation; returns statistics.
    """
    t = _as_2d(truth_scores)
    gi = _as_2d(got_idx)
    B, N = t.shape
    kk = min(k, N)
    pinned_total = 0
    recall_hits = 0
    for b in range(B):
        row = t[b].copy()
        row[np.isnan(row)] = -np.inf
        if kk == 0:          # nothing to rank (empty corpus): the answer is all padding
            assert np.all(gi[b] == -1), f"query {b}: padding must be -1"
            continue
        # the kk + 64 best rows in canonical order (score descending, index ascending): every row that reaches the
        # m-th best score is a candidate (ties at the cut included), and only those are sorted
        m = min(N, kk + 64)
        if m < N:
            cut = -np.partition(-row, m - 1)[m - 1]
            cands = np.flatnonzero(row >= cut)
        else:
            cands = np.arange(N)
        ext = cands[np.lexsort((cands, -row[cands]))][:m]
        gs_row = None if got_scores is None else _as_2d(got_scores)[b]
        p, h = _check_one_query(b, ext, row[ext], lambda j: row[j], gi[b, :kk], gs_row, kk, k, gap, score_tol)
        pinned_total += p
        recall_hits += h
        if kk < k:
            assert np.all(gi[b, kk:] == -1), f"query {b}: padding must be -1"
    return {"pinned": pinned_total, "positions": B * kk,
            "recall": recall_hits / float(B * kk) if B * kk else 1.0}


class ChunkedTruth:
    """The same protocol for corpora whose [B x N] fp64 score matrix does not fit: feed the corpus chunk by chunk
    (`add`), each chunk's fp64 scores are reduced at once to the ``k + 64`` best rows per query plus the fp64 scores of
    the rows the answer under test returned; `check` then applies `check_topk_against_truth`'s rules.  ``queries`` are
    the prepared operand values (what the kernels multiply); ``got_idx`` are global row ids."""

    def __init__(self, queries: np.ndarray, got_idx: np.ndarray, k: int, keep_extra: int = 64):
        self.q64 = _as_2d(queries).astype(np.float64)
        self.got = _as_2d(got_idx).astype(np.int64)
        self.k = int(k)
        self.keep = self.k + keep_extra
        B = self.q64.shape[0]
        self.best_s = np.full((B, 0), -np.inf)
        self.best_i = np.zeros((B, 0), dtype=np.int64)
        self.got_s = np.full(self.got.shape, np.nan)
        self.n = 0

    def scores_of_chunk(self, rows: np.ndarray) -> np.ndarray:
        return self.q64 @ _as_2d(rows).astype(np.float64).T

    def add(self, rows: np.ndarray, row0: int, allowed: Optional[np.ndarray] = None) -> None:
        """``rows``: prepared values of global rows ``[row0, row0 + len(rows))``; ``allowed``: optional bool per row
        (filtered search: disallowed rows do not exist)."""
        self.add_scores(self.scores_of_chunk(rows), row0, allowed)

    def add_scores(self, s: np.ndarray, row0: int, allowed: Optional[np.ndarray] = None) -> None:
        s = np.array(s, dtype=np.float64)
        s[np.isnan(s)] = -np.inf
        if allowed is not None:
            s[:, ~np.asarray(allowed, dtype=bool)] = -np.inf
        B, m = s.shape
        self.n += m if allowed is None else int(np.count_nonzero(allowed))
        inside = (self.got >= row0) & (self.got < row0 + m)
        bb, jj = np.nonzero(inside)
        self.got_s[bb, jj] = s[bb, self.got[bb, jj] - row0]
        keep = min(self.keep, m)
        part = np.argpartition(-s, keep - 1, axis=1)[:, :keep] if keep < m else np.tile(np.arange(m), (B, 1))
        cand_s = np.take_along_axis(s, part, axis=1)
        cand_i = part.astype(np.int64) + row0
        # everything tied with the chunk's keep-th value would be needed for exactness of ties across the cut; the
        # extra 64 rows make a 65-fold near-tie at the k boundary the only case that could be missed
        all_s = np.concatenate([self.best_s, cand_s], axis=1)
        all_i = np.concatenate([self.best_i, cand_i], axis=1)
        order = np.lexsort((all_i, -all_s), axis=1)[:, : self.keep]
        self.best_s = np.take_along_axis(all_s, order, axis=1)
        self.best_i = np.take_along_axis(all_i, order, axis=1)

    def merge(self, other_best_s: np.ndarray, other_best_i: np.ndarray, other_got_s: np.ndarray, other_n: int) -> None:
        """Fold in another accumulator's state (sharded checks: every rank accumulates its own rows)."""
        all_s = np.concatenate([self.best_s, other_best_s], axis=1)
        all_i = np.concatenate([self.best_i, other_best_i], axis=1)
        order = np.lexsort((all_i, -all_s), axis=1)[:, : self.keep]
        self.best_s = np.take_along_axis(all_s, order, axis=1)
        self.best_i = np.take_along_axis(all_i, order, axis=1)
        self.got_s = np.where(np.isnan(self.got_s), other_got_s, self.got_s)
        self.n += other_n

    def check(self, got_scores: Optional[np.ndarray], gap: float = 1e-6, score_tol: float = 1e-5) -> Dict[str, float]:
        B = self.q64.shape[0]
        kk = min(self.k, self.n)
        pinned_total = recall_hits = 0
        for b in range(B):
            if kk == 0:
                assert np.all(self.got[b] == -1), f"query {b}: padding must be -1"
                continue
            valid = self.best_s[b] > -np.inf
            ext, sv = self.best_i[b][valid], self.best_s[b][valid]
            got = self.got[b, :kk]
            assert np.all(got >= 0), f"query {b}: padding inside the first {kk} results: {got}"
            lookup = {int(j): float(v) for j, v in zip(got, self.got_s[b, :kk])}
            assert not any(np.isnan(v) for v in lookup.values()), f"query {b}: returned rows outside the corpus: {got}"
            gs_row = None if got_scores is None else _as_2d(got_scores)[b]
            p, h = _check_one_query(b, ext, sv, lambda j: lookup[j], got, gs_row, kk, self.k, gap, score_tol)
            pinned_total += p
            recall_hits += h
            if kk < self.k:
                assert np.all(self.got[b, kk:] == -1), f"query {b}: padding must be -1"
        return {"pinned": pinned_total, "positions": B * kk, "recall": recall_hits / float(B * kk) if B * kk else 1.0}


def recall_at_k(truth_idx: np.ndarray, got_idx: np.ndarray) -> float:
    t, g = _as_2d(truth_idx), _as_2d(got_idx)
    hits = sum(len(set(t[b].tolist()) & set(g[b].tolist())) for b in range(t.shape[0]))
    return hits / float(t.size)


# ----------------------------------------------------------------------------
# pgvector-shaped search (streamlit_app.py:253-283 and :317-364)
# ----------------------------------------------------------------------------

def pgvector_search(query_vec: np.ndarray, embeddings: np.ndarray, top_k: int):
    """``ORDER BY e.embedding <#> q ASC LIMIT k`` with
    ``similarity = 1.0 - (e.embedding <#> q)`` (``streamlit_app.py:275,282``).

    ``<#>`` is the *negative* inner product, so similarity is ``1 + <e, q>``.
    Returns (indices int64 [k], similarity float64 [k]).
    """
    q = np.asarray(query_vec, dtype=np.float32).reshape(1, -1)
    ip = dot_scores(q, embeddings)
    vals, idx = topk_canonical(ip, top_k)
    dist = -vals[0].astype(np.float64)
    return idx[0], 1.0 - dist


def citation_weighted_rerank(indices: np.ndarray, similarity: np.ndarray,
                             citations: Sequence[Optional[int]], weight: float, top_k: int):
    """Re-rank a candidate pool by ``similarity + w * ln(citations)`` when
    citations > 0 else ``similarity`` (``streamlit_app.py:351-363``); ties broken
    by similarity descending.  The pool is ``max(50, 10*top_k)`` (``:317``)."""
    sim = np.asarray(similarity, dtype=np.float64)
    bonus = np.array([math.log(float(c)) if (c is not None and c > 0) else 0.0 for c in citations])
    weighted = sim + weight * bonus
    order = np.lexsort((np.arange(len(sim)), -sim, -weighted))[:top_k]
    return np.asarray(indices)[order], sim[order], weighted[order]


def pool_size(top_k: int) -> int:
    """``streamlit_app.py:317``."""
    return max(50, int(top_k) * 10)


# ----------------------------------------------------------------------------
# retrieval metrics (compare_embeddings.py:47-371)
# ----------------------------------------------------------------------------

def rank_concepts(sim_matrix) -> List[np.ndarray]:
    """``compare_embeddings.py:47-52``: per row, doc indices by descending score."""
    return [np.argsort(-np.asarray(row)) for row in sim_matrix]


def _ranking(sim_matrix: np.ndarray) -> np.ndarray:
    return np.argsort(-np.asarray(sim_matrix), axis=1)


def _gold_doc(rels: Mapping[int, float]) -> int:
    # compare_embeddings.py:111,135,158 - first doc whose grade equals 1
    for doc, grade in rels.items():
        if grade == 1:
            return doc
    raise StopIteration("qrels row has no document with grade 1")


def precision_at_k(sim_matrix, qrels, k=5) -> float:
    """``compare_embeddings.py:95-117``: mean over queries of hit/k."""
    ranked = _ranking(sim_matrix)
    vals = []
    for q in range(ranked.shape[0]):
        gold = _gold_doc(qrels[q])
        vals.append((1 if gold in ranked[q, :k] else 0) / k)
    return float(np.mean(vals))


def hit_at_k(sim_matrix, qrels, k=5) -> float:
    """``compare_embeddings.py:120-140``."""
    ranked = _ranking(sim_matrix)
    vals = [1.0 if _gold_doc(qrels[q]) in ranked[q, :k] else 0.0 for q in range(ranked.shape[0])]
    return float(np.mean(vals))


def mrr_at_k(sim_matrix, qrels, k=None) -> float:
    """``compare_embeddings.py:143-173``."""
    ranked = _ranking(sim_matrix)
    vals = []
    for q in range(ranked.shape[0]):
        gold = _gold_doc(qrels[q])
        row = ranked[q] if k is None else ranked[q][:k]
        where = np.flatnonzero(row == gold)
        vals.append(1.0 / (int(where[0]) + 1) if where.size else 0.0)
    return float(np.mean(vals))


def generate_qrels(queries, slogans) -> Dict[int, Dict[int, float]]:
    """``compare_embeddings.py:175-182``: grade 0.5 for a same-paper slogan, else 0."""
    out = {}
    for i, (_, paper) in enumerate(queries):
        out[i] = {j: (0.5 if slogans[j][1] == paper else 0) for j in range(len(slogans))}
    return out


def _grades_in_order(order, rels: Mapping[int, float], k) -> np.ndarray:
    # compare_embeddings.py:185-193 / :246-254
    if k is not None:
        order = order[:k]
    return np.array([rels.get(int(d), 0.0) for d in order], dtype=float)


def _dcg(grades: np.ndarray, gain: str) -> float:
    # compare_embeddings.py:196-213
    if grades.size == 0:
        return 0.0
    if gain == "exp":
        g = np.power(2.0, grades) - 1.0
    elif gain == "linear":
        g = grades
    else:
        raise ValueError(f"Unknown gain scheme: {gain}")
    return float(np.sum(g * (1.0 / np.log2(np.arange(2, grades.size + 2)))))


def ndcg_at_k(sim_matrix, qrels, k=10, gain="exp") -> float:
    """``compare_embeddings.py:216-243`` (the first argument is the similarity
    matrix; it is re-argsorted inside, ``:223``)."""
    ranked = _ranking(sim_matrix)
    vals = []
    for q in range(ranked.shape[0]):
        rels = qrels.get(q, {})
        dcg = _dcg(_grades_in_order(ranked[q], rels, k), gain)
        ideal = np.sort(np.array(list(rels.values()), dtype=float))[::-1]
        if k is not None:
            ideal = ideal[:k]
        idcg = _dcg(ideal, gain)
        vals.append(0.0 if idcg == 0.0 else dcg / idcg)
    return float(np.mean(vals))


def _infer_max_rel(qrels) -> float:
    # compare_embeddings.py:271-277 / :330-336
    m = 0.0
    for rels in qrels.values():
        if rels:
            m = max(m, max(rels.values()))
    return m


def err_at_k(sim_matrix, qrels, k=10, max_rel=None) -> float:
    """``compare_embeddings.py:257-311``."""
    ranked = _ranking(sim_matrix)
    if max_rel is None:
        max_rel = _infer_max_rel(qrels)
        if max_rel <= 0.0:
            return 0.0
    denom = 2.0 ** max_rel
    vals = []
    for q in range(ranked.shape[0]):
        rels = qrels.get(q, None)
        if not rels:
            vals.append(0.0)
            continue
        grades = _grades_in_order(ranked[q], rels, k)
        if grades.size == 0:
            vals.append(0.0)
            continue
        p = (np.power(2.0, grades) - 1.0) / denom
        err, keep_going = 0.0, 1.0
        for rank, pi in enumerate(p, start=1):
            if pi > 0.0:
                err += keep_going * pi * (1.0 / rank)
            keep_going *= (1.0 - pi)
            if pi > 0.0 and keep_going <= 1e-12:
                break
        vals.append(err)
    return float(np.mean(vals)) if vals else 0.0


def q_measure_at_k(sim_matrix, qrels, k=10, max_rel=None) -> float:
    """``compare_embeddings.py:315-371``."""
    ranked = _ranking(sim_matrix)
    if max_rel is None:
        max_rel = _infer_max_rel(qrels)
        if max_rel <= 0.0:
            return 0.0
    denom = 2.0 ** max_rel
    vals = []
    for q in range(ranked.shape[0]):
        rels = qrels.get(q, None)
        if not rels:
            vals.append(0.0)
            continue
        all_gain = (np.power(2.0, np.array(list(rels.values()), dtype=float)) - 1.0) / denom
        cg_star = all_gain.sum()
        if cg_star <= 0.0:
            vals.append(0.0)
            continue
        gains = (np.power(2.0, _grades_in_order(ranked[q], rels, k)) - 1.0) / denom
        cg = acc = 0.0
        for rank, g in enumerate(gains, start=1):
            if g > 0.0:
                cg += g
                acc += g * (cg / rank)
        vals.append(acc / cg_star)
    return float(np.mean(vals)) if vals else 0.0


# ----------------------------------------------------------------------------
# text assembly (app_create_embeddings.py:48-70)
# ----------------------------------------------------------------------------

def global_context(paper: Mapping) -> str:
    """``app_create_embeddings.py:48-52``: the three headed blocks joined by a
    blank line (each block is always non-empty because of its heading)."""
    parts = [
        f"**Global Notations:**\n{paper.get('global_notations', '')}",
        f"**Global Definitions:**\n{paper.get('global_definitions', '')}",
        f"**Global Assumptions:**\n{paper.get('global_assumptions', '')}",
    ]
    return "\n\n".join(p for p in parts if p)


def text_to_embed(paper: Mapping, theorem: Mapping) -> str:
    """``app_create_embeddings.py:69``."""
    return f"{global_context(paper)}\n\n**{theorem['type'].capitalize()}:**\n{theorem['content']}"


# ----------------------------------------------------------------------------
# synthetic data recipe shared by tests and bench
# ----------------------------------------------------------------------------

CHUNK_ROWS = 250_000


def synth_chunk(chunk_id: int, rows: int = CHUNK_ROWS, d: int = 768, seed: int = 1234,
                bf16: bool = False) -> np.ndarray:
    """The benchmark's synthetic rows, restated with the oracle's own `l2_normalize` / `f32_to_bf16_bits` (the
    generator bench.py uses lives in /synthetic.py; tests/test_oracle_golden.py checks the two agree bit for bit)."""
    x = np.random.default_rng([seed, chunk_id]).standard_normal((rows, d), dtype=np.float32)
    out = np.empty((rows, d), dtype=np.uint16) if bf16 else x
    for r0 in range(0, rows, 16384):
        blk = l2_normalize(x[r0:r0 + 16384])
        if bf16:
            out[r0:r0 + 16384] = f32_to_bf16_bits(blk)
        else:
            x[r0:r0 + 16384] = blk
    return out


def synth_queries(batch_id: int, nq: int, d: int = 768, seed: int = 5678, bf16: bool = False) -> np.ndarray:
    return synth_chunk(batch_id, nq, d, seed, bf16)


def golden_inputs(N: int, B: int, d: int, seed: int, metric: str):
    """Inputs of the ``tests/golden/search_*.npz`` cases (see oracle/gen_golden.py).

    Only elementwise operations, so the arrays regenerate bit-identically on any
    host.  metric "cos": raw rows with a spread of norms (the normalisation is part
    of what is tested); metric "ip": rows scaled by 1/sqrt(d) (norm close to 1, like
    the stored-normalised RDS rows) and used as they are.
    """
    c = np.random.default_rng([seed, 0]).standard_normal((N, d), dtype=np.float32)
    q = np.random.default_rng([seed, 1]).standard_normal((B, d), dtype=np.float32)
    if metric == "cos":
        c *= np.random.default_rng([seed, 2]).uniform(0.25, 4.0, size=(N, 1)).astype(np.float32)
    else:
        s = np.float32(1.0 / math.sqrt(d))
        c *= s
        q *= s
    return q, c


def fast_inputs(N: int, B: int, d: int, seed: int, metric: str):
    """`golden_inputs`' recipe with the corpus drawn block by block from independent streams on a few threads: for the
    large cases of the GPU tests, which no committed fixture depends on (one stream takes 8 s for 310,000 x 768)."""
    from concurrent.futures import ThreadPoolExecutor
    c = np.empty((N, d), dtype=np.float32)
    blocks = [(r0, min(N, r0 + 32768)) for r0 in range(0, N, 32768)]

    def one(b):
        rng = np.random.default_rng([seed, 0, b[0]])
        rng.standard_normal(out=c[b[0]:b[1]], dtype=np.float32)
        if metric == "cos":
            c[b[0]:b[1]] *= rng.uniform(0.25, 4.0, size=(b[1] - b[0], 1)).astype(np.float32)
        else:
            c[b[0]:b[1]] *= np.float32(1.0 / math.sqrt(d))

    with ThreadPoolExecutor(_threads()) as ex:
        list(ex.map(one, blocks))
    q = np.random.default_rng([seed, 1]).standard_normal((B, d), dtype=np.float32)
    if metric != "cos":
        q *= np.float32(1.0 / math.sqrt(d))
    return q, c


def inputs(N: int, B: int, d: int, seed: int, metric: str):
    """Seeded test inputs: the fixtures' own generator for small cases, the threaded one from 2^24 corpus values up."""
    return fast_inputs(N, B, d, seed, metric) if N * d >= (1 << 24) else golden_inputs(N, B, d, seed, metric)


# ----------------------------------------------------------------------------
# C restatement of the pgvector scan (oracle/pgvector_ip.c), loaded on demand
# ----------------------------------------------------------------------------

def pgvector_c_search(query_vec: np.ndarray, embeddings: np.ndarray, top_k: int):
    """``ORDER BY embedding <#> q LIMIT k`` by the sequential fp32 loop of oracle/pgvector_ip.c.
    Returns (rows int64 [k], similarity float64 [k]) with similarity = 1 - distance."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libpgvector_ip.so")
    lib = ctypes.CDLL(path)
    lib.pgv_order_by_ip_limit.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
    e = np.ascontiguousarray(embeddings, dtype=np.float32)
    q = np.ascontiguousarray(query_vec, dtype=np.float32).reshape(-1)
    rows = np.empty(top_k, dtype=np.int64)
    dist = np.empty(top_k, dtype=np.float32)
    rc = lib.pgv_order_by_ip_limit(e.ctypes.data, e.shape[0], e.shape[1], q.ctypes.data, int(top_k), rows.ctypes.data,
                                   dist.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"pgv_order_by_ip_limit failed: {rc}")
    return rows, 1.0 - dist.astype(np.float64)


def showcase_filter_search(cos_scores, theorems_data, filters, pool=200):
    """The reference's filtered search, restated (app_showcase_model.py:93-129): rank all rows, keep the best
    min(pool, N), walk them in order and keep those passing every sidebar predicate until top_k are found.
    Returns (rows, exhausted) - `exhausted` is True when the pool ran out before top_k matches."""
    cos_scores = np.asarray(cos_scores)
    n = len(theorems_data)
    order = topk_canonical(cos_scores[None, :].astype(np.float64), min(pool, n))[1][0]
    want = filters["top_k"]
    rows = []
    for i in order:
        it = theorems_data[int(i)]
        ok = (not filters["types"] or it["type"].lower() in filters["types"])
        ok = ok and (not filters["tags"] or it["primary_math_tag"] in filters["tags"])
        ok = ok and (not filters["authors"] or any(a in it["authors"] for a in filters["authors"]))
        ok = ok and it["source"] in filters["sources"]
        ok = ok and filters["citation_range"][0] <= it["citations"] <= filters["citation_range"][1]
        if it["source"] == "arXiv":
            if filters["year_range"]:
                ok = ok and filters["year_range"][0] <= it.get("year", 0) <= filters["year_range"][1]
            if filters["journal_status"] == "Journal Article":
                ok = ok and bool(it.get("journal_published", False))
            elif filters["journal_status"] == "Preprint Only":
                ok = ok and not it.get("journal_published", False)
        if ok:
            rows.append(int(i))
        if len(rows) >= want:
            break
    return rows, len(rows) < want


def rank_of(truth: np.ndarray, rows) -> np.ndarray:
    """0-based position of rows[q] in the canonical order (score desc, index asc) of truth[q] - what
    ``np.flatnonzero(np.argsort(-truth[q], kind="stable") == rows[q])[0]`` gives (the lookup inside the reference's
    mrr_at_k, compare_embeddings.py:104-123); -1 when the row is out of range or its score is NaN."""
    truth = _as_2d(truth)
    out = np.full(truth.shape[0], -1, dtype=np.int64)
    for q, t in enumerate(np.asarray(rows, dtype=np.int64)):
        if 0 <= t < truth.shape[1] and truth[q, t] == truth[q, t]:
            s = truth[q]
            out[q] = int(np.sum(s > s[t]) + np.sum(s[:t] == s[t]))
    return out


def sql_where(row, f) -> bool:
    """The WHERE clause of the production app restated clause by clause with SQL's NULL semantics spelled out as
    None-propagating values (streamlit_app.py:175-243); a row is kept when every clause evaluates to TRUE."""
    def ilike(text, pat):            # text ILIKE '%pat%'  ->  True / False / None
        return None if text is None else (pat.lower() in text.lower())

    def NOT(v):
        return None if v is None else (not v)

    def AND(*vs):
        if any(v is False for v in vs):
            return False
        return None if any(v is None for v in vs) else True

    def OR(*vs):
        if any(v is True for v in vs):
            return True
        return None if any(v is None for v in vs) else False

    clauses = []
    link = row.get("link")
    arx = ilike(link, "arxiv.org")
    if f["sources"]:
        cases = []
        if "arXiv" in f["sources"]:
            cases.append(arx)
        if "Stacks Project" in f["sources"]:
            cases.append(NOT(arx))
        if cases:
            clauses.append(OR(*cases))
    if f["authors"]:
        a = row.get("authors")
        clauses.append(None if a is None else bool(set(a) & set(f["authors"])))
    if f["tags"]:
        c = row.get("primary_category")
        clauses.append(None if c is None else (c in f["tags"]))
    if f["year_range"]:
        y = row.get("year")
        between = None if y is None else (f["year_range"][0] <= y <= f["year_range"][1])
        clauses.append(OR(AND(arx, between), NOT(arx)))
    if f["journal_status"] == "Journal Article":
        clauses.append(AND(arx, row.get("journal_ref") is not None))
    elif f["journal_status"] == "Preprint Only":
        clauses.append(AND(arx, row.get("journal_ref") is None))
    pf = f.get("paper_filter", {"ids": set(), "titles": set()})
    pfc = []
    if pf.get("ids"):
        pfc.append(OR(*[ilike(link, i) for i in pf["ids"]]))
    if pf.get("titles"):
        pfc.append(OR(*[ilike(row.get("title"), t) for t in pf["titles"]]))
    if pfc:
        clauses.append(OR(*pfc))
    if f["types"]:
        name = row.get("type_name")
        clauses.append(OR(*[ilike(None if name is None else name.lower(), t) for t in f["types"]]))
    lo, hi = f["citation_range"]
    cit = row.get("citations")
    between = None if cit is None else (lo <= cit <= hi)
    if f["include_unknown_citations"]:
        clauses.append(OR(between, cit is None))
    else:
        clauses.append(AND(cit is not None, between))
    return all(c is True for c in clauses)


def cpu_reference_topk(q: np.ndarray, c: np.ndarray, k: int, threads: Optional[int] = None):
    """The reference's own CPU formulation, as its scripts run it on the host: ``util.cos_sim`` = L2-normalise both sides
    and one fp32 matmul (sentence_transformers: ``torch.mm(F.normalize(a), F.normalize(b).T)``), then
    ``np.argsort(-S, axis=1)[:, :k]`` (compare_embeddings.py:61,105).  torch on the CPU, ``threads`` host threads
    (default: all).  This is what bench.py times as ``cpu_baseline`` (kind "port"); returns (indices, seconds)."""
    import time

    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(int(threads))
    a, b = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
    t0 = time.perf_counter()
    S = torch.mm(F.normalize(a, p=2, dim=1), F.normalize(b, p=2, dim=1).T).numpy()
    top = np.argsort(-S, axis=1)[:, :k]
    return top, time.perf_counter() - t0
