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

import re

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


# ---- the in-database form: WHERE clauses of the production app ---------------------------------------------------------
# the "paper" box of the production app (streamlit_app.py:44-47,118-143): arXiv ids and title fragments, comma-separated
ARXIV_ID_RE = re.compile(r"(?:arxiv\.org/(?:abs|pdf)/)?((?:\d{4}\.\d{4,5}|[a-z\-]+/\d{7}))", re.IGNORECASE)


def extract_arxiv_id(s: str):
    """The arXiv id inside ``s`` (a URL or the bare id), or None."""
    if not s:
        return None
    m = ARXIV_ID_RE.search(s.strip())
    return m.group(1) if m else None


def normalize_title(s: str) -> str:
    return (s or "").casefold().strip()


def parse_paper_filter(raw: str) -> dict:
    """``"2401.12345, Optimal Transport"`` -> ``{"ids": {"2401.12345"}, "titles": {"optimal transport"}}``: the
    ``paper_filter`` entry of the filter state `sql_filter_mask` evaluates (ILIKE on the paper's link / title)."""
    ids, titles = set(), set()
    for token in (t.strip() for t in (raw or "").split(",")):
        if not token:
            continue
        arx = extract_arxiv_id(token)
        if arx:
            ids.add(arx.lower())
        else:
            titles.add(normalize_title(token))
    return {"ids": ids, "titles": titles}


def _ilike(text, needle: str) -> bool:
    """``text ILIKE '%needle%'`` for a non-NULL text (case-insensitive substring)."""
    return needle.lower() in text.lower()


def _sql_row_passes(row: Mapping, f: Mapping) -> bool:
    """One joined row (paper p, theorem t) against the WHERE clauses streamlit_app.py:175-243 assembles.  SQL's
    three-valued logic is kept: a comparison with NULL is not true, so the row is dropped - e.g. a paper without a
    link matches neither ``ILIKE '%arxiv.org%'`` nor ``NOT ILIKE '%arxiv.org%'``."""
    link = row.get("link")
    is_arxiv = None if link is None else _ilike(link, "arxiv.org")
    if f.get("sources"):
        ok = ("arXiv" in f["sources"] and is_arxiv is True) or ("Stacks Project" in f["sources"] and is_arxiv is False)
        if (("arXiv" in f["sources"]) or ("Stacks Project" in f["sources"])) and not ok:
            return False
    if f.get("authors"):
        authors = row.get("authors")
        if authors is None or not set(authors) & set(f["authors"]):           # p.authors && %s
            return False
    if f.get("tags"):
        if row.get("primary_category") is None or row["primary_category"] not in f["tags"]:
            return False
    if f.get("year_range"):
        y0, y1 = f["year_range"]
        year = row.get("year")                                                  # EXTRACT(YEAR FROM p.last_updated)
        arxiv_in_range = is_arxiv is True and year is not None and y0 <= year <= y1
        if not (arxiv_in_range or is_arxiv is False):
            return False
    status = f.get("journal_status", "All")
    if status == "Journal Article" and not (is_arxiv is True and row.get("journal_ref") is not None):
        return False
    if status == "Preprint Only" and not (is_arxiv is True and row.get("journal_ref") is None):
        return False
    pf = f.get("paper_filter") or {}
    ids, titles = pf.get("ids") or (), pf.get("titles") or ()
    if ids or titles:
        by_id = link is not None and any(_ilike(link, i) for i in ids)
        by_title = row.get("title") is not None and any(_ilike(row["title"], t) for t in titles)
        if not (by_id or by_title):
            return False
    if f.get("types"):
        name = row.get("type_name")
        if name is None or not any(_ilike(name.lower(), t) for t in f["types"]):   # lower(t.name) ILIKE ANY
            return False
    lo, hi = f["citation_range"]
    cit = row.get("citations")
    if cit is None:
        return bool(f.get("include_unknown_citations"))
    return lo <= cit <= hi


def sql_filter_mask(rows: Sequence[Mapping], filters: Mapping) -> np.ndarray:
    """bool[N] for the joined rows behind the embedding table (keys: link, authors, primary_category, year,
    journal_ref, title, type_name, citations), in index order: the rows the production app's WHERE clause keeps.
    Feed it to ``pgvector.search(..., mask=...)``: ORDER BY <#> LIMIT k over exactly those rows."""
    return np.fromiter((_sql_row_passes(r, filters) for r in rows), dtype=bool, count=len(rows))
