"""The mirrors of the reference's call-site bodies against what those bodies themselves printed when executed
(tests/golden/callsites.json, made by oracle/gen_golden.py from compare_embeddings.py:14-35,55-92 and
app_scratchpad.py:120-154).  On CPU the score matrix / top-k come from the oracle's numpy restatement (the HIP library
has no CPU path); tests/test_mirrors_gpu.py repeats the comparison through libtsearch."""
import numpy as np

from callsites_common import RecordingStreamlit, StubModel, run_compare, run_evaluate
from conftest import load_json
from oracle import oracle


def test_compare_embeddings_prints_what_the_reference_prints(monkeypatch, capsys):
    from theoremsearch_amd import compare_embeddings as ce
    monkeypatch.setattr(ce.util, "cos_sim", lambda a, b: oracle.cos_sim(np.asarray(a), np.asarray(b)))
    case = load_json("callsites.json")["cases"]["compare_embeddings"]
    assert run_compare(ce, case, capsys) == case["stdout"]


def test_compare_embeddings_tie_order_is_the_references(monkeypatch, capsys):
    """argmax = first maximum; argsort()[::-1] = among equal scores the higher index first."""
    from theoremsearch_amd import compare_embeddings as ce
    sims = np.array([[0.5, 0.9, 0.9, 0.1, 0.9, 0.5]], np.float32)
    monkeypatch.setattr(ce.util, "cos_sim", lambda a, b: sims)
    names = [f"c{j}" for j in range(6)]
    ce.compare_embeddings(StubModel(1, 8), ["x"], names, top_k=4)
    out = capsys.readouterr().out
    want_best = names[int(sims[0].argmax())]
    want_top = [names[j] for j in sims[0].argsort()[::-1][:4]]        # the reference's two expressions on this row
    assert f"best match: {want_best!r}" in out and want_best == "c1"
    got_top = [line.split("'")[1] for line in out.splitlines() if line.strip()[:2] in ("1.", "2.", "3.", "4.")]
    assert got_top == want_top == ["c4", "c2", "c1", "c5"]


def test_evaluate_retrieval_prints_what_the_reference_prints(monkeypatch, capsys):
    from theoremsearch_amd import compare_embeddings as ce
    monkeypatch.setattr(ce.util, "cos_sim", lambda a, b: oracle.cos_sim(np.asarray(a), np.asarray(b)))
    case = load_json("callsites.json")["cases"]["evaluate_retrieval"]
    assert run_evaluate(ce, case, capsys) == case["stdout"]


def test_search_theorems_displays_what_the_reference_displays(monkeypatch):
    from theoremsearch_amd import app_scratchpad

    class OracleIndex:                     # stands in for TheoremIndex on a host without a GPU
        def __init__(self, rows):
            self.rows, self.n, self.row_offset = rows, rows.shape[0], 0

        def search(self, q, k):
            return oracle.search(np.asarray(q).reshape(1, -1), self.rows, k, "cos", "f32")

    case = load_json("callsites.json")["cases"]["search_theorems"]
    model = StubModel(case["seed"], case["d"])
    data = case["theorems_data"]
    db = model.encode([t["text_to_embed"] for t in data])
    monkeypatch.setattr(app_scratchpad, "TheoremIndex", OracleIndex)
    st = RecordingStreamlit()
    app_scratchpad.search_theorems(case["query"], model, data, OracleIndex(db), st)
    assert [c[1] for c in st.calls if c[0] == "expander"] == case["expander_titles"]
    assert st.calls[0] == ["subheader", "Top 5 Most Similar Theorems"]
    # everything except the LaTeX clean-up of the bodies (UI code, out of scope) is the reference's call sequence
    ref = [c for c in case["calls"]]
    assert [c[0] for c in st.calls] == [c[0] for c in ref]
    st2 = RecordingStreamlit()
    app_scratchpad.search_theorems("", model, data, OracleIndex(db), st2)
    assert st2.calls == case["empty_query_calls"]


def _showcase_filters(f):
    f = dict(f)
    f["citation_range"] = tuple(f["citation_range"])
    if f["year_range"] is not None:
        f["year_range"] = tuple(f["year_range"])
    return f


def test_search_and_display_displays_what_the_reference_displays(monkeypatch):
    """theoremsearch_amd.app_showcase_model.search_and_display against the recorded streamlit calls of the reference's own
    function (app_showcase_model.py:79-156, tests/golden/showcase.json) for seven sidebar states, an empty query included;
    the index is stood in for by the oracle over the allowed rows (no GPU here; tests/test_mirrors_gpu.py runs libtsearch)."""
    from theoremsearch_amd import app_showcase_model

    class OracleIndex:
        def __init__(self, rows):
            self.rows, self.n, self.row_offset = rows, rows.shape[0], 0

        def search(self, q, k, mask=None):
            allowed = np.flatnonzero(mask) if mask is not None else np.arange(self.n)
            scores = np.full((1, k), -np.inf, np.float32)
            idx = np.full((1, k), -1, np.int64)
            if allowed.size:
                s, i = oracle.search(np.asarray(q).reshape(1, -1), self.rows[allowed], min(k, allowed.size), "cos", "f32")
                scores[0, :s.shape[1]], idx[0, :s.shape[1]] = s[0], allowed[i[0]]
            return scores, idx

    case = load_json("showcase.json")
    model = StubModel(case["seed"], case["d"])
    data = case["theorems_data"]
    db = model.encode([t["text_to_embed"] for t in data])
    monkeypatch.setattr(app_showcase_model, "TheoremIndex", OracleIndex)
    for name, state in case["states"].items():
        st = RecordingStreamlit()
        app_showcase_model.search_and_display(case["query"], model, data, OracleIndex(db), _showcase_filters(state["filters"]), st)
        assert st.calls == state["calls"], name
    st = RecordingStreamlit()
    app_showcase_model.search_and_display("", model, data, OracleIndex(db), _showcase_filters(case["states"]["open"]["filters"]), st)
    assert st.calls == case["empty_query_calls"]


def test_load_and_prepare_data_builds_the_references_records(tmp_path):
    """app_scratchpad.load_and_prepare_data against the records (and warnings) of the reference's own function
    (app_scratchpad.py:23-63, tests/golden/scratchpad_data.json): text_to_embed is what the corpus embeddings are made of."""
    import json
    from theoremsearch_amd import app_scratchpad
    case = load_json("scratchpad_data.json")
    for name, body in case["papers"].items():
        (tmp_path / name).write_text(json.dumps(body), encoding="utf-8")
    (tmp_path / "broken.json").write_text(case["broken"])
    st = RecordingStreamlit()
    got = app_scratchpad.load_and_prepare_data([str(tmp_path / n) for n in case["order"]], st)
    assert got == case["records"]
    assert [[c[0], c[1].replace(str(tmp_path) + "/", "<dir>/")] for c in st.calls] == case["warnings"]
    assert app_scratchpad.load_and_prepare_data([str(tmp_path / "missing.json")]) == []        # no streamlit: skipped silently
