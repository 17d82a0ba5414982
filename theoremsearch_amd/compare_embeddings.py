"""Mirror of the function surface of the reference's ``compare_embeddings.py`` (lines 11-371).

Same names, parameters, defaults and printed report; the similarity matrix comes from the HIP score
kernel (``util.cos_sim``) instead of ``sentence_transformers.util.cos_sim``.  The six metrics take
the similarity matrix (the reference's graded metrics call that parameter ``ranked`` but re-argsort
it inside, reference lines 223/268/327) and a qrels dict ``{query: {doc: grade}}``.

Selection is done with a partial sort of the top ``k`` instead of the reference's full
``np.argsort(-sim_matrix, axis=1)`` per metric; on tie-free scores the results are identical.
The notebook cells of the reference file (RDS pulls, lines 375-556) are out of scope.
"""
from __future__ import annotations

import numpy as np

from . import util
from .encoder import SentenceEncoder


# --- 1. model -------------------------------------------------------------------------------------
def load_model(model_name="math-similarity/Bert-MLM_arXiv-MP-class_zbMath"):
    return SentenceEncoder(model_name)


def compare_embeddings(model, latex_texts, concept_texts, top_k=3):
    """Pairwise similarities and the top-k concept matches of each latex token (reference 14-35)."""
    l_emb = model.encode(latex_texts, convert_to_tensor=True)
    c_emb = model.encode(concept_texts, convert_to_tensor=True)
    sim_matrix = util.cos_sim(l_emb, c_emb)
    for i, latex in enumerate(latex_texts):
        sims = sim_matrix[i]
        # the reference's two selections, including what they do on exact ties: the best match is the FIRST maximum
        # (argmax, :28), the list is the reversed ascending sort (:31: among equal scores the higher index comes first)
        best_idx = int(np.argmax(sims))
        print(f"{latex!r}  -> best match: {concept_texts[best_idx]!r} (score {sims[best_idx]:.4f})")
        order = np.lexsort((-np.arange(sims.shape[0]), -sims))[:top_k]     # score descending, then index descending
        print("  top matches:")
        for rank, j in enumerate(order, start=1):
            print(f"    {rank}. {concept_texts[j]!r} (score {sims[j]:.4f})")
        print()


# --- 2. ranking helpers -----------------------------------------------------------------------------
def rank_concepts(sim_matrix):
    """List of arrays: doc indices by descending score, one per query (reference 47-52)."""
    return [np.argsort(-np.asarray(row)) for row in sim_matrix]


class IndexRanking:
    """Stands in for the ``[Q x N]`` similarity matrix when the corpus lives in a `TheoremIndex` and the matrix
    would not fit (10M slogans x 1k queries = 40 GB): the metric functions below only ever need the k best docs
    per query (``index.search``) or the position of one relevant doc in the full ranking (``index.rank_of``, a
    counting pass), never the matrix itself.  Pass it wherever they take ``sim_matrix`` / ``ranked``."""

    def __init__(self, index, query_emb):
        self.index = index
        self.query_emb = np.asarray(query_emb.detach().cpu().float().numpy() if hasattr(query_emb, "detach") else query_emb,
                                    dtype=np.float32)
        self.shape = (self.query_emb.shape[0], index.n)
        self._cache = {}

    def top(self, k):
        k = min(int(k), self.index.n)
        if k > 256:
            raise ValueError("IndexRanking serves k <= 256; use rank_of for positions further down")
        if k not in self._cache:
            self._cache[k] = self.index.search(self.query_emb, k)[1]
        return self._cache[k]

    def rank_of(self, docs):
        return self.index.rank_of(self.query_emb, docs)[0]


class _SharedRanking(np.ndarray):
    """The similarity matrix `evaluate_retrieval` hands to its six metrics: an ndarray (the metrics take ``sim_matrix``, as
    in the reference) that remembers the rankings already taken of it - the reference argsorts the same matrix six times
    (compare_embeddings.py:105,129,152,223,268,327); here the selection runs once per k."""

    def __new__(cls, a):
        obj = np.asarray(a).view(cls)
        obj._tops = {}
        return obj

    def __array_finalize__(self, obj):
        self._tops = {}                    # a slice or a copy is another matrix: nothing is remembered for it


def _top(sim_matrix, k):
    """Indices of the k best docs per query, best first (all docs when k is None)."""
    if isinstance(sim_matrix, IndexRanking):
        if k is None:
            raise ValueError("a full ranking is not materialised for an IndexRanking")
        return sim_matrix.top(k)
    shared = sim_matrix._tops if isinstance(sim_matrix, _SharedRanking) else None
    if shared is not None:
        if k in shared:
            return shared[k]
        if k is not None:
            # the partial selection's order is canonical (score descending, then index ascending): the k best are the
            # head of any longer list it made
            wider = [kk for kk in shared if kk is not None and kk > k]
            if wider:
                return shared[wider[0]][:, :k]
    s = np.asarray(sim_matrix)
    if k is None or k >= s.shape[1]:
        idx = np.argsort(-s, axis=1)
    else:
        _, idx = util.topk(s, k)
    if shared is not None:
        shared[k] = idx
    return idx


def _exact_doc(rels):
    for doc, grade in rels.items():
        if grade == 1:
            return doc
    raise StopIteration  # what next() raises in the reference (lines 111, 135, 158)


# --- 3. metrics -------------------------------------------------------------------------------------
def evaluate_retrieval(model, theorems, queries, qrels, top_k_report=3):
    print("Encoding...")
    s_emb = model.encode([item[0] for item in theorems], convert_to_tensor=True)
    q_emb = model.encode([item[0] for item in queries], convert_to_tensor=True)
    print("Creating sim_matrix...")
    sim_matrix = _SharedRanking(util.cos_sim(q_emb, s_emb))
    print("Cos-sim matrix dim", sim_matrix.shape)
    print("Ranking concepts...")
    print("=" * 50)
    _top(sim_matrix, max(1, top_k_report))          # P@1 reads the head of the same selection
    print("Binary metrics")
    bin_metrics = {"P@1": precision_at_k, f"H@{top_k_report}": hit_at_k, f"MRR@{top_k_report}": mrr_at_k}
    for item, fn in bin_metrics.items():
        res = fn(sim_matrix, qrels, k=(1 if item[0] == "P" else top_k_report))
        print(f"{item} | {res}")
    print("=" * 50)
    print("Graded metrics")
    grad_metrics = {f"nDCG@{top_k_report}": ndcg_at_k, f"ERR@{top_k_report}": err_at_k,
                    f"Q-measure@{top_k_report}": q_measure_at_k}
    for item, fn in grad_metrics.items():
        res = fn(sim_matrix, qrels, k=top_k_report)
        print(f"{item} | {res}")


def evaluate_retrieval_index(model, index, queries, qrels, top_k_report=3):
    """`evaluate_retrieval` against a corpus that is already a `TheoremIndex` (10M slogans rather than the script's
    few hundred): same six metrics and printed report, from top-k searches and one counting pass - no ``[Q x N]``
    similarity matrix.  Returns the metrics as a dict as well."""
    q_emb = model.encode([item[0] for item in queries], convert_to_numpy=True)
    ranking = IndexRanking(index, q_emb)
    out = {"P@1": precision_at_k(ranking, qrels, k=1), f"H@{top_k_report}": hit_at_k(ranking, qrels, k=top_k_report),
           f"MRR@{top_k_report}": mrr_at_k(ranking, qrels, k=top_k_report), "MRR": mrr_at_k(ranking, qrels, k=None),
           f"nDCG@{top_k_report}": ndcg_at_k(ranking, qrels, k=top_k_report),
           f"ERR@{top_k_report}": err_at_k(ranking, qrels, k=top_k_report),
           f"Q-measure@{top_k_report}": q_measure_at_k(ranking, qrels, k=top_k_report)}
    print("=" * 50)
    for item, res in out.items():
        print(f"{item} | {res}")
    return out


def _exact_docs(qrels, nq):
    return np.array([_exact_doc(qrels[q]) for q in range(nq)])


def precision_at_k(sim_matrix, qrels, k=5):
    top = _top(sim_matrix, k)
    hit = (top == _exact_docs(qrels, top.shape[0])[:, None]).any(axis=1)
    return float(np.mean(hit / k))


def hit_at_k(sim_matrix, qrels, k=5):
    top = _top(sim_matrix, k)
    hit = (top == _exact_docs(qrels, top.shape[0])[:, None]).any(axis=1)
    return float(np.mean(hit.astype(float)))


def mrr_at_k(sim_matrix, qrels, k=None):
    if isinstance(sim_matrix, IndexRanking) and (k is None or k > 256):
        # position of the exact doc in the full ranking, by counting the rows that beat it
        docs = [_exact_doc(qrels[q]) for q in range(sim_matrix.shape[0])]
        ranks = sim_matrix.rank_of(docs)
        return float(np.mean([1.0 / (r + 1) if (r >= 0 and (k is None or r < k)) else 0.0 for r in ranks]))
    top = _top(sim_matrix, k)
    match = top == _exact_docs(qrels, top.shape[0])[:, None]
    if match.shape[1] == 0:
        return float(np.mean(np.zeros(match.shape[0])))
    return float(np.mean(np.where(match.any(axis=1), 1.0 / (match.argmax(axis=1) + 1), 0.0)))


def _generate_qrels(queries, slogans):
    return {i: {j: 0.5 if slogans[j][1] == queries[i][1] else 0 for j in range(len(slogans))}
            for i in range(len(queries))}


def _get_rels_for_query(order, rels_dict, k=None, default=0.0):
    if k is not None:
        order = order[:k]
    return np.array([rels_dict.get(d, default) for d in order], dtype=float)


def _get_rels_sparse(order, rels_dict, k=None, default=0.0):
    if k is not None:
        order = order[:k]
    return np.array([rels_dict.get(int(d), default) for d in order], dtype=float)


def _dcg_from_rels(rels, gain="exp"):
    if rels.size == 0:
        print("TOO SMALL")          # the reference says so on stdout (compare_embeddings.py:201-203): same printed report
        return 0.0
    if gain == "exp":
        gains = np.exp2(rels) - 1.0
    elif gain == "linear":
        gains = rels
    else:
        raise ValueError(f"Unknown gain scheme: {gain}")
    return float(np.sum(gains * (1.0 / np.log2(np.arange(2, rels.size + 2)))))


def _grade_matrix(top, qrels):
    """``[Q x k]`` grades of the selected docs (`_get_rels_sparse` for every query at once: one dict lookup per selected doc)."""
    empty = {}
    return np.array([[(qrels.get(q) or empty).get(d, 0.0) for d in row] for q, row in enumerate(top.tolist())],
                    dtype=float).reshape(top.shape)


def _per_query(arrays, fn):
    """``fn`` (a reduction over axis 1) of every query's array, queries with arrays of one length stacked and reduced together:
    the same values, summed in the same order as one array at a time."""
    out = np.zeros(len(arrays))
    lengths = np.array([a.size for a in arrays], dtype=np.int64)
    for ln in np.unique(lengths):
        if ln:
            rows = np.flatnonzero(lengths == ln)
            out[rows] = fn(np.stack([arrays[r] for r in rows]))
    return out


def _gains(rels, gain):
    if gain == "exp":
        return np.exp2(rels) - 1.0
    if gain == "linear":
        return rels
    raise ValueError(f"Unknown gain scheme: {gain}")


def ndcg_at_k(ranked, qrels, k=10, gain="exp"):
    top = _top(ranked, k)
    nq, kk = top.shape
    dicts = [qrels.get(q, {}) for q in range(nq)]
    if kk == 0 or not all(dicts):
        # some DCG is taken of nothing: the reference says "TOO SMALL" once per such call, in query order - one query at a time
        out = []
        for q in range(nq):
            dcg = _dcg_from_rels(_get_rels_sparse(top[q], dicts[q], k), gain=gain)
            ideal = np.sort(np.array(list(dicts[q].values()), dtype=float))[::-1]
            idcg = _dcg_from_rels(ideal if k is None else ideal[:k], gain=gain)
            out.append(0.0 if idcg == 0.0 else dcg / idcg)
        return float(np.mean(out))
    # every query at once; sums run over the same values in the same order as `_dcg_from_rels` on one query's row
    dcg = np.sum(_gains(_grade_matrix(top, qrels), gain) * (1.0 / np.log2(np.arange(2, kk + 2))), axis=1)
    ideals = [np.sort(np.array(list(d.values()), dtype=float))[::-1][:k] for d in dicts]
    idcg = _per_query(ideals, lambda a: np.sum(_gains(a, gain) * (1.0 / np.log2(np.arange(2, a.shape[1] + 2))), axis=1))
    return float(np.mean(np.divide(dcg, idcg, out=np.zeros(nq), where=idcg != 0.0)))


def _max_grade(qrels):
    best = 0.0
    for rels_dict in qrels.values():
        if rels_dict:
            best = max(best, max(rels_dict.values()))
    return best


def err_at_k(ranked, qrels, k=10, max_rel=None):
    top = _top(ranked, k)
    if max_rel is None:
        max_rel = _max_grade(qrels)
        if max_rel <= 0.0:
            return 0.0
    denom = 2.0 ** max_rel
    nq, kk = top.shape
    if nq == 0:
        return 0.0
    judged = np.array([bool(qrels.get(q, None)) for q in range(nq)])
    if kk == 0:
        return float(np.mean(np.zeros(nq)))
    # the cascade of reference lines 295-307 for every query at once: `going` before rank i is the running product of
    # (1 - p) over the ranks above it (accumulated in rank order, as the loop does), a query stops counting behind the first
    # rank that leaves going <= 1e-12, and the terms are added in rank order
    p = (np.exp2(_grade_matrix(top, qrels)) - 1.0) / denom
    after = np.multiply.accumulate(1.0 - p, axis=1)
    before = np.concatenate([np.ones((nq, 1)), after[:, :-1]], axis=1)
    stop = np.logical_or.accumulate((p > 0.0) & (after <= 1e-12), axis=1)
    counted = (p > 0.0) & ~np.concatenate([np.zeros((nq, 1), dtype=bool), stop[:, :-1]], axis=1)
    terms = np.where(counted, before * p * (1.0 / np.arange(1, kk + 1)), 0.0)
    return float(np.mean(np.where(judged, np.add.accumulate(terms, axis=1)[:, -1], 0.0)))


def q_measure_at_k(ranked, qrels, k=10, max_rel=None):
    top = _top(ranked, k)
    if max_rel is None:
        max_rel = _max_grade(qrels)
        if max_rel <= 0.0:
            return 0.0
    denom = 2.0 ** max_rel
    nq, kk = top.shape
    if nq == 0:
        return 0.0
    dicts = [qrels.get(q, None) or {} for q in range(nq)]
    ideal_gain = _per_query([np.array(list(d.values()), dtype=float) for d in dicts],
                            lambda a: ((np.exp2(a) - 1.0) / denom).sum(axis=1))
    if kk == 0:
        return float(np.mean(np.zeros(nq)))
    # reference lines 347-369 for every query at once: the cumulated gain and the total are running sums in rank order
    gains = (np.exp2(_grade_matrix(top, qrels)) - 1.0) / denom
    pos = gains > 0.0
    cum = np.add.accumulate(np.where(pos, gains, 0.0), axis=1)
    total = np.add.accumulate(np.where(pos, gains * (cum / np.arange(1, kk + 1)), 0.0), axis=1)[:, -1]
    return float(np.mean(np.divide(total, ideal_gain, out=np.zeros(nq), where=ideal_gain > 0.0)))
