"""The search lines of the reference's scratch app (``app_scratchpad.py:128-130``)::

    query_emb      = model.encode(query, convert_to_tensor=True)
    cosine_scores  = util.cos_sim(query_emb, embeddings_db)[0]
    top_indices    = np.argsort(-cosine_scores.cpu())[:5]

as one fused top-5 search of a device-resident index.  `search_theorems` returns the five hits; rendering them
(``app_scratchpad.py:132-154``) stays the app's own code.
"""
from __future__ import annotations

from .index import TheoremIndex


def search_theorems(query, model, theorems_data, embeddings_db, top_k: int = 5):
    """``[{"info": theorems_data[row], "similarity": float}]`` of the ``top_k`` (the app: 5) most similar theorems, best first;
    ``None`` for an empty query (the reference returns before searching).  ``embeddings_db``: a `TheoremIndex` (kept across
    calls) or the ``[N x d]`` matrix / tensor ``load_embedding_library`` returns (indexed for this call)."""
    if not query:
        return None
    query_emb = model.encode(query, convert_to_tensor=True)
    own = not isinstance(embeddings_db, TheoremIndex)
    index = TheoremIndex.from_embeddings(embeddings_db, metric="cos") if own else embeddings_db
    try:
        scores, top_indices = index.search(query_emb, min(int(top_k), index.n))
        offset = index.row_offset
    finally:
        if own:
            index.close()
    return [{"info": theorems_data[int(idx) - offset], "similarity": float(similarity)}
            for idx, similarity in zip(top_indices[0], scores[0]) if idx >= 0]
