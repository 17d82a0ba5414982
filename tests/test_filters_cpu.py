"""filter_mask (the bitmask the scan kernel consumes) against the restated reference loop
(oracle.showcase_filter_search, app_showcase_model.py:93-129) - host logic only."""
import numpy as np
import pytest

from filters_common import filter_states, make_sql_rows, make_theorems, sql_filter_states
from oracle import oracle
from theoremsearch_amd.filters import filter_mask, sql_filter_mask


@pytest.mark.parametrize("state", list(filter_states()))
def test_mask_agrees_with_reference_loop(state):
    n = 3000
    data = make_theorems(n)
    f = filter_states(top_k=10)[state]
    mask = filter_mask(data, f)
    assert mask.dtype == bool and mask.shape == (n,)
    rng = np.random.default_rng(3)
    cos = rng.standard_normal(n).astype(np.float32)
    # with the pool widened to N the reference loop IS "best top_k rows that pass"
    rows, exhausted = oracle.showcase_filter_search(cos, data, f, pool=n)
    allowed = np.flatnonzero(mask)
    best = allowed[np.argsort(-cos[allowed].astype(np.float64), kind="stable")][: f["top_k"]]
    assert rows == [int(i) for i in best]
    assert exhausted == (mask.sum() < f["top_k"])
    # and every row the 200-pool loop returns passes the mask
    rows200, _ = oracle.showcase_filter_search(cos, data, f)
    assert all(mask[i] for i in rows200)
    assert rows200 == rows[: len(rows200)]


def test_states_cover_empty_full_and_selective():
    data = make_theorems(3000)
    s = filter_states()
    assert filter_mask(data, s["open"]).all()
    assert not filter_mask(data, s["nothing"]).any()
    sel = filter_mask(data, s["selective"]).mean()
    assert 0 < sel < 0.01


@pytest.mark.parametrize("state", list(sql_filter_states()))
def test_sql_where_mask_agrees_with_the_clause_by_clause_restatement(state):
    """filters.sql_filter_mask (what the scan kernel consumes as a bitmask) against oracle.sql_where, the WHERE clause
    of streamlit_app.py:175-243 restated with explicit NULL logic - rows with NULL links, authors, years, titles,
    type names and citation counts included."""
    rows = make_sql_rows(4000)
    f = sql_filter_states()[state]
    mask = sql_filter_mask(rows, f)
    want = np.array([oracle.sql_where(r, f) for r in rows])
    assert np.array_equal(mask, want)
    if state == "open":
        assert 0.85 < mask.mean() < 0.95          # the rows with a NULL link match neither source clause
    else:
        assert 0 < mask.sum() < len(rows)


def test_paper_filter_parsing_is_the_references():
    """filters.parse_paper_filter / extract_arxiv_id / normalize_title against the outputs of the reference's own functions
    (streamlit_app.py:118-143, tests/golden/paper_filter.json), and the parsed state drives sql_filter_mask's paper clause."""
    from conftest import load_json
    from theoremsearch_amd import filters
    case = load_json("paper_filter.json")
    for x, want, wid, wt in zip(case["inputs"], case["parsed"], case["ids"], case["titles"]):
        got = filters.parse_paper_filter(x)
        assert {"ids": sorted(got["ids"]), "titles": sorted(got["titles"])} == want, x
        assert filters.extract_arxiv_id(x) == wid and filters.normalize_title(x) == wt
    rows = [{"link": "https://arxiv.org/abs/2401.12345", "title": "On Optimal Transport", "citations": 3},
            {"link": "https://arxiv.org/abs/1706.03762", "title": "Attention", "citations": 3},
            {"link": None, "title": "Stacks: optimal TRANSPORT of schemes", "citations": 3}]
    f = {"sources": [], "citation_range": (0, 10), "paper_filter": filters.parse_paper_filter("2401.12345")}
    assert filters.sql_filter_mask(rows, f).tolist() == [True, False, False]
    f["paper_filter"] = filters.parse_paper_filter("Optimal Transport")
    assert filters.sql_filter_mask(rows, f).tolist() == [True, False, True]
