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

from .index import TheoremIndex


def pool_size(top_k: int) -> int:
    return max(50, int(top_k) * 10)  # streamlit_app.py:317


def search(index: TheoremIndex, query_vec, top_k: int, citation_weight: float = 0.0,
           citations: Optional[Sequence[Optional[int]]] = None):
    """Returns a list of dicts ``{"row", "similarity", "score"}`` ordered like the SQL result."""
    q = np.asarray(query_vec, dtype=np.float32).reshape(1, -1)
    if citation_weight == 0.0:
        scores, idx = index.search(q, int(top_k))
        return [{"row": int(i), "similarity": 1.0 + float(s), "score": 1.0 + float(s)}
                for s, i in zip(scores[0], idx[0]) if i >= 0]
    if citations is None:
        raise ValueError("citation-weighted search needs the per-row citation counts")
    pool = min(pool_size(top_k), 256)
    scores, idx = index.search(q, pool)
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
