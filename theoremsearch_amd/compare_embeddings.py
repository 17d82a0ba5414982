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
    if shared is not None and k in shared:
        return shared[k]
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


def precision_at_k(sim_matrix, qrels, k=5):
    top = _top(sim_matrix, k)
    return float(np.mean([(1 if _exact_doc(qrels[q]) in top[q] else 0) / k for q in range(top.shape[0])]))


def hit_at_k(sim_matrix, qrels, k=5):
    top = _top(sim_matrix, k)
    return float(np.mean([1.0 if _exact_doc(qrels[q]) in top[q] else 0.0 for q in range(top.shape[0])]))


def mrr_at_k(sim_matrix, qrels, k=None):
    if isinstance(sim_matrix, IndexRanking) and (k is None or k > 256):
        # position of the exact doc in the full ranking, by counting the rows that beat it
        docs = [_exact_doc(qrels[q]) for q in range(sim_matrix.shape[0])]
        ranks = sim_matrix.rank_of(docs)
        return float(np.mean([1.0 / (r + 1) if (r >= 0 and (k is None or r < k)) else 0.0 for r in ranks]))
    top = _top(sim_matrix, k)
    rr = []
    for q in range(top.shape[0]):
        pos = np.flatnonzero(top[q] == _exact_doc(qrels[q]))
        rr.append(1.0 / (int(pos[0]) + 1) if pos.size else 0.0)
    return float(np.mean(rr))


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


def ndcg_at_k(ranked, qrels, k=10, gain="exp"):
    top = _top(ranked, k)
    out = []
    for q in range(top.shape[0]):
        rels_dict = qrels.get(q, {})
        dcg = _dcg_from_rels(_get_rels_sparse(top[q], rels_dict, k), gain=gain)
        ideal = np.sort(np.array(list(rels_dict.values()), dtype=float))[::-1]
        idcg = _dcg_from_rels(ideal if k is None else ideal[:k], gain=gain)
        out.append(0.0 if idcg == 0.0 else dcg / idcg)
    return float(np.mean(out))


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
    out = []
    for q in range(top.shape[0]):
        rels_dict = qrels.get(q, None)
        if not rels_dict:
            out.append(0.0)
            continue
        rels = _get_rels_sparse(top[q], rels_dict, k=k)
        stop_prob = (np.exp2(rels) - 1.0) / denom
        err_q, going = 0.0, 1.0
        for i, p in enumerate(stop_prob, start=1):
            if p > 0.0:
                err_q += going * p * (1.0 / i)
            going *= (1.0 - p)
            if p > 0.0 and going <= 1e-12:
                break
        out.append(err_q)
    return float(np.mean(out)) if out else 0.0


def q_measure_at_k(ranked, qrels, k=10, max_rel=None):
    top = _top(ranked, k)
    if max_rel is None:
        max_rel = _max_grade(qrels)
        if max_rel <= 0.0:
            return 0.0
    denom = 2.0 ** max_rel
    out = []
    for q in range(top.shape[0]):
        rels_dict = qrels.get(q, None)
        if not rels_dict:
            out.append(0.0)
            continue
        ideal_gain = ((np.exp2(np.array(list(rels_dict.values()), dtype=float)) - 1.0) / denom).sum()
        if ideal_gain <= 0.0:
            out.append(0.0)
            continue
        gains = (np.exp2(_get_rels_sparse(top[q], rels_dict, k=k)) - 1.0) / denom
        cum = total = 0.0
        for i, g in enumerate(gains, start=1):
            if g > 0.0:
                cum += g
                total += g * (cum / i)
        out.append(total / ideal_gain)
    return float(np.mean(out)) if out else 0.0
