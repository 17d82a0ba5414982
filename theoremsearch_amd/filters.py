"""Metadata filters evaluated to a row bitmask, applied INSIDE the top-k scan.

The showcase app ranks all rows, keeps the best ``min(200, N)`` and walks that pool until ``top_k`` rows
pass the sidebar filters (reference app_showcase_model.py:93-129); a selective filter therefore returns
fewer than ``top_k`` hits even when more exist.  Here the predicates are evaluated once per filter state
to one bit per row (`filter_mask`), and `TheoremIndex.search(..., mask=)` returns the ``top_k`` best rows
that pass - identical to the reference's answer whenever its pool held ``top_k`` matches, and complete
when it did not.  The SQL form (streamlit_app.py:175-243 builds the same predicates as a WHERE clause in
front of ``ORDER BY <#> LIMIT k``) maps to the same mask.
"""
from __future__ import annotations

from typing import Mapping, Sequence

import numpy as np


def _passes(item: Mapping, f: Mapping) -> bool:
    """One theorem against the sidebar state (predicates of app_showcase_model.py:102-121)."""
    if f.get("types") and item["type"].lower() not in f["types"]:
        return False
    if f.get("tags") and item["primary_math_tag"] not in f["tags"]:
        return False
    if f.get("authors") and not any(a in item["authors"] for a in f["authors"]):
        return False
    if item["source"] not in f["sources"]:
        return False
    lo, hi = f["citation_range"]
    if not (lo <= item["citations"] <= hi):
        return False
    if item["source"] == "arXiv":
        if f.get("year_range"):
            y0, y1 = f["year_range"]
            if not (y0 <= item.get("year", 0) <= y1):
                return False
        published = bool(item.get("journal_published", False))
        status = f.get("journal_status")
        if status == "Journal Article" and not published:
            return False
        if status == "Preprint Only" and published:
            return False
    return True


def filter_mask(theorems_data: Sequence[Mapping], filters: Mapping) -> np.ndarray:
    """bool[N]: which rows of the library pass ``filters`` (keys as the app's sidebar builds them:
    types, tags, authors, sources, citation_range, year_range, journal_status)."""
    return np.fromiter((_passes(it, filters) for it in theorems_data), dtype=bool, count=len(theorems_data))


def filtered_index(index, theorems_data: Sequence[Mapping], filters: Mapping):
    """A sub-index of the rows that pass ``filters`` (`TheoremIndex.subset`): for query batches, or a filter state that
    stays put over many searches - each search then costs only the allowed rows and runs the batched MFMA path.
    Its results carry the ids of ``index``."""
    return index.subset(filter_mask(theorems_data, filters))


def search_filtered(index, query_emb, theorems_data: Sequence[Mapping], filters: Mapping, mask: np.ndarray | None = None):
    """The filtered result list of the showcase app: ``[{"info": item, "similarity": cos}]``, best first,
    at most ``filters["top_k"]`` entries, exact over ALL rows that pass (not only a top-200 pool)."""
    if mask is None:
        mask = filter_mask(theorems_data, filters)
    k = max(1, min(int(filters["top_k"]), 256, len(theorems_data)))
    scores, idx = index.search(query_emb, k, mask=mask)
    out = []
    for s, i in zip(scores[0], idx[0]):
        if i < 0:
            break
        out.append({"info": theorems_data[int(i)], "similarity": float(s)})
    return out
