"""Host-side mirrors of the reference modules (no GPU): metric functions against the reference's
own outputs, text assembly, selection helper, encoder call surface, shard partition."""
import json
import os
import pickle

import numpy as np
import pytest

from conftest import load_json, load_qrels, metric_cases
from theoremsearch_amd import app_create_embeddings as ace
from theoremsearch_amd import compare_embeddings as ce
from theoremsearch_amd import distributed, pgvector, util
from theoremsearch_amd.encoder import HashingTokenizer, SentenceEncoder


@pytest.mark.parametrize("name", metric_cases())
def test_metrics_match_reference_outputs(name):
    doc = load_json(f"metrics_{name}.json")
    sim = np.array(doc["sim_matrix"], dtype=np.float32)
    qrels = load_qrels(doc)
    k, exp = doc["k"], doc["expected"]
    got = {
        "precision_at_k": ce.precision_at_k(sim, qrels, k=k),
        "precision_at_1": ce.precision_at_k(sim, qrels, k=1),
        "hit_at_k": ce.hit_at_k(sim, qrels, k=k),
        "mrr_at_k": ce.mrr_at_k(sim, qrels, k=k),
        "mrr_all": ce.mrr_at_k(sim, qrels, k=None),
        "ndcg_at_k": ce.ndcg_at_k(sim, qrels, k=k),
        "ndcg_linear": ce.ndcg_at_k(sim, qrels, k=k, gain="linear"),
        "err_at_k": ce.err_at_k(sim, qrels, k=k),
        "err_maxrel4": ce.err_at_k(sim, qrels, k=k, max_rel=4.0),
        "q_measure_at_k": ce.q_measure_at_k(sim, qrels, k=k),
    }
    for key, val in got.items():
        assert val == pytest.approx(exp[key], rel=0, abs=1e-12), key
    assert [int(i) for i in ce.rank_concepts(sim)[0]] == exp["rank_row0"]
    if "generated_qrels" in doc:
        gen = ce._generate_qrels([tuple(x) for x in doc["queries"]], [tuple(x) for x in doc["slogans"]])
        assert gen == load_qrels(doc, "generated_qrels")


@pytest.mark.parametrize("style", ["sparse_graded", "dense_generated", "identity", "some_unjudged"])
def test_metrics_over_all_queries_at_once_equal_the_oracles_loops(style, capsys):
    """The six metrics are evaluated for every query at once (running products and sums in rank order); the oracle's
    restatement - pinned by the reference's own function bodies (tests/golden/metrics_*.json) - walks one query and one rank
    at a time, as compare_embeddings.py:95-371 does.  Seeded random matrices and judgements: the same floats, bit for bit."""
    from oracle import oracle
    for seed in range(40):
        rng = np.random.default_rng([seed, len(style)])
        nq, n = int(rng.integers(1, 40)), int(rng.integers(3, 400))
        sim = rng.standard_normal((nq, n)).astype(np.float32)
        qrels = {}
        for q in range(nq):
            gold = int(rng.integers(n))
            if style == "sparse_graded":
                d = {gold: 1.0}
                for j in rng.choice(n, size=min(n, int(rng.integers(0, 6))), replace=False):
                    d.setdefault(int(j), float(rng.choice([0.5, 0.25, 2.0, 3.0, 0])))
            elif style == "dense_generated":          # what _generate_qrels makes (every doc a key), plus the exact doc
                d = {j: (0.5 if rng.random() < 0.05 else 0) for j in range(n)}
                d[gold] = 1
            elif style == "identity":
                d = {gold: 1}
            else:
                d = {gold: 1.0}
                d.setdefault(int(rng.integers(n)), 0.5)
            qrels[q] = d
        graded = {q: (v if q % 3 else {}) for q, v in qrels.items()} if style == "some_unjudged" else qrels
        shared = ce._SharedRanking(sim)
        ce._top(shared, min(n, 37))                    # one selection: the smaller k below read its head
        for k in (1, 3, 10, min(n, 37)):
            for mine, theirs, rels, kw in (
                    (ce.precision_at_k, oracle.precision_at_k, qrels, dict(k=k)), (ce.hit_at_k, oracle.hit_at_k, qrels, dict(k=k)),
                    (ce.mrr_at_k, oracle.mrr_at_k, qrels, dict(k=k)), (ce.mrr_at_k, oracle.mrr_at_k, qrels, dict(k=None)),
                    (ce.ndcg_at_k, oracle.ndcg_at_k, graded, dict(k=k)), (ce.ndcg_at_k, oracle.ndcg_at_k, graded, dict(k=k, gain="linear")),
                    (ce.err_at_k, oracle.err_at_k, graded, dict(k=k)), (ce.err_at_k, oracle.err_at_k, graded, dict(k=k, max_rel=1.0)),
                    (ce.q_measure_at_k, oracle.q_measure_at_k, graded, dict(k=k)),
                    (ce.q_measure_at_k, oracle.q_measure_at_k, graded, dict(k=k, max_rel=4.0))):
                assert mine(shared, rels, **kw) == theirs(sim, rels, **kw), (seed, mine.__name__, kw)
    capsys.readouterr()                                # "TOO SMALL" lines of the unjudged queries (the reference prints them too)


def test_metric_signatures_and_defaults():
    import inspect
    want = {"precision_at_k": ["sim_matrix", "qrels", "k"], "hit_at_k": ["sim_matrix", "qrels", "k"],
            "mrr_at_k": ["sim_matrix", "qrels", "k"], "ndcg_at_k": ["ranked", "qrels", "k", "gain"],
            "err_at_k": ["ranked", "qrels", "k", "max_rel"], "q_measure_at_k": ["ranked", "qrels", "k", "max_rel"],
            "evaluate_retrieval": ["model", "theorems", "queries", "qrels", "top_k_report"],
            "compare_embeddings": ["model", "latex_texts", "concept_texts", "top_k"],
            "rank_concepts": ["sim_matrix"], "load_model": ["model_name"]}
    for name, params in want.items():
        assert list(inspect.signature(getattr(ce, name)).parameters) == params, name
    assert inspect.signature(ce.precision_at_k).parameters["k"].default == 5
    assert inspect.signature(ce.mrr_at_k).parameters["k"].default is None
    assert inspect.signature(ce.ndcg_at_k).parameters["k"].default == 10
    assert inspect.signature(ce.evaluate_retrieval).parameters["top_k_report"].default == 3
    with pytest.raises(StopIteration):
        ce.precision_at_k(np.zeros((1, 3), np.float32), {0: {0: 0.5}}, k=1)
    with pytest.raises(ValueError):
        ce._dcg_from_rels(np.array([1.0]), gain="nope")


def test_text_assembly_matches_reference_strings():
    for case in load_json("text_to_embed.json"):
        paper = dict(case["paper"])
        paper["theorems"] = [case["theorem"]]
        rec = ace.theorem_records(paper)[0]
        assert rec["global_context"] == case["global_context"]
        assert rec["text_to_embed"] == case["text_to_embed"]
        assert list(rec) == ["paper_title", "paper_url", "authors", "citations", "primary_math_tag", "year", "source",
                             "journal_published", "type", "content", "global_context", "text_to_embed"]
    assert (ace.MODEL_NAME, ace.PARSED_PAPERS_DIR, ace.OUTPUT_DIR) == (
        "math-similarity/Bert-MLM_arXiv-MP-class_zbMath", "./app_papers", "./app_embeds")


def test_topk_helper_on_adversarial_fixtures():
    adv = load_json("adversarial.json")
    t = adv["ties6"]
    vals, idx = util.topk(np.array(t["scores"], np.float32), t["k"])
    assert idx.tolist() == [1, 2, 4, 0]
    assert sorted(vals.tolist()) == sorted(np.array(t["scores"], np.float32)[t["torch_topk"]].tolist())
    z = np.zeros(1000, np.float32)
    z[[7, 500, 900]] = 1.0
    assert util.topk(z, 3)[1].tolist() == [7, 500, 900]
    n = np.array([0.3, np.nan, 0.7, 0.1], np.float32)
    assert util.topk(n, 2)[1].tolist() == [2, 0]
    vals, idx = util.topk(np.random.default_rng(0).standard_normal((5, 100)).astype(np.float32), 7)
    assert idx.shape == (5, 7) and (np.diff(vals, axis=1) <= 0).all()


@pytest.fixture(scope="module")
def tiny_encoder():
    return SentenceEncoder(num_layers=1, device="cpu", allow_random_init=True)


def test_encoder_call_surface(tiny_encoder):
    m = tiny_encoder
    texts = ["A tree on $n$ vertices has $n-1$ edges.", "Let $X$ be a scheme.", ""]
    e = m.encode(texts, normalize_embeddings=True, convert_to_numpy=True, show_progress_bar=False, batch_size=2)
    assert isinstance(e, np.ndarray) and e.shape == (3, 768) and e.dtype == np.float32
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
    one = m.encode(texts[0], convert_to_tensor=True)
    assert tuple(one.shape) == (768,) and str(one.dtype) == "torch.float32"
    raw = m.encode(texts)                       # not normalised by default (app_create_embeddings.py:81)
    assert not np.allclose(np.linalg.norm(raw, axis=1), 1.0, atol=1e-3)
    assert np.allclose(raw[0] / np.linalg.norm(raw[0]), e[0], atol=1e-5)
    again = m.encode(list(reversed(texts)), normalize_embeddings=True)[::-1]
    assert np.allclose(again, e, atol=1e-5)       # batching / length sorting does not change rows
    assert m.encode([]).shape == (0, 768)
    assert m.get_sentence_embedding_dimension() == 768 and not m.pretrained
    assert callable(m.similarity) and m.pipeline.similarity_fn_name == "cosine"      # experiments/first_experiment.py:195


def test_hashing_tokenizer_is_stable():
    tok = HashingTokenizer()
    a = tok.token_ids(r"Let $\alpha \in X$ be 42.")
    assert a[0] == 101 and a[-1] == 102 and a == tok.token_ids(r"let $\ALPHA \in x$ be 42.")
    assert tok.token_ids("x " * 2000).__len__() == 512
    enc = tok(["a b c", "a"])
    assert enc["input_ids"].shape == (2, 5) and enc["attention_mask"].sum().item() == 8


def test_embed_texts_mirrors(monkeypatch, tiny_encoder):
    from theoremsearch_amd import embeddings, generate_embeddings
    embeddings._get_embedder.cache_clear()
    monkeypatch.setattr(embeddings, "SentenceEncoder", lambda name: tiny_encoder)
    out = embeddings.embed_texts(["a", "b c"])
    assert isinstance(out, list) and isinstance(out[0], list) and isinstance(out[0][0], float) and len(out[0]) == 768
    assert abs(sum(v * v for v in out[1]) - 1.0) < 1e-5
    embeddings._get_embedder.cache_clear()
    assert generate_embeddings.EMBEDDERS == {"qwen": "Qwen/Qwen3-Embedding-0.6B", "gemma": "google/embeddinggemma-300m"}
    out2 = generate_embeddings.embed_texts(tiny_encoder, ["a", "b c"], batch_size=16)
    assert np.allclose(np.array(out2), np.array(out), atol=1e-6)


def test_create_embedding_library_writes_reference_files(tmp_path, monkeypatch, tiny_encoder):
    papers = tmp_path / "app_papers"
    papers.mkdir()
    cases = load_json("text_to_embed.json")
    doc = dict(cases[0]["paper"])
    (papers / "p1.json").write_text(json.dumps(doc))
    monkeypatch.setattr(ace, "PARSED_PAPERS_DIR", str(papers))
    monkeypatch.setattr(ace, "OUTPUT_DIR", str(tmp_path / "app_embeds"))
    ace.create_embedding_library(model=tiny_encoder)
    emb, data = ace.load_embedding_library(str(tmp_path / "app_embeds"))
    assert tuple(emb.shape) == (len(doc["theorems"]), 768) and str(emb.dtype) == "torch.float32"
    assert [d["text_to_embed"] for d in data] == [c["text_to_embed"] for c in cases[: len(doc["theorems"])]]
    assert pickle.load(open(tmp_path / "app_embeds" / "theorems_data.pkl", "rb")) == data
    assert ace.load_embedding_library(str(tmp_path / "nope")) == (None, None)


def test_partition_and_pool():
    for n, w in ((10, 3), (10_000_000, 8), (7, 8), (0, 2)):
        cuts = [distributed.shard_bounds(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
    assert pgvector.pool_size(3) == 50 and pgvector.pool_size(20) == 200


def test_pgvector_text_rows_parse_like_vector_in():
    """ts_parse_pgvector_text (host code of libtsearch): COPY-style rows -> fp32 matrix, one strtof per value; rows cut
    by the end of a chunk are left for the next call; malformed rows raise."""
    from theoremsearch_amd import TSearchError
    rng = np.random.default_rng(0)
    m = rng.standard_normal((37, 12)).astype(np.float32)
    m[0, 0], m[1, 1], m[2, 2], m[3, 3] = 1e-30, -3.5e12, 0.0, 16777217.0
    txt = "".join(f"{100 + i}\t[{', '.join(repr(float(v)) for v in row)}]\n" for i, row in enumerate(m))
    out, used = pgvector.parse_vectors(txt, 12)
    assert out.dtype == np.float32 and np.array_equal(out, m) and used == len(txt) - 1
    # fed in pieces that cut rows anywhere
    got, tail = [], b""
    raw = txt.encode()
    for lo in range(0, len(raw), 97):
        buf = tail + raw[lo:lo + 97]
        rows, used = pgvector.parse_vectors(buf, 12)
        got.append(rows)
        tail = buf[used:]
    assert np.array_equal(np.concatenate(got), m)
    # max_rows, exponent forms, no spaces
    two, _ = pgvector.parse_vectors("[1,2e0,-3E-1][4,5,6][7,8,9]", 3, max_rows=2)
    assert np.array_equal(two, np.array([[1, 2, -0.3], [4, 5, 6]], np.float32))
    for bad in ("[1,2]", "[1,2,3,4]", "[1,x,3]"):
        with pytest.raises(TSearchError):
            pgvector.parse_vectors(bad, 3)
    empty, used = pgvector.parse_vectors("no vectors here", 3)
    assert empty.shape == (0, 3) and used == 0


def _parse_vectors_restated(buf: bytes, d: int, max_rows: int):
    """What ts_parse_pgvector_text promises, said again in Python: ``(rows, consumed)`` or None where it refuses the text."""
    ws, n, pos, done, rows = b" \t\n\r\x0b\x0c", len(buf), 0, 0, []
    while len(rows) < max_rows:
        i = buf.find(b"[", pos)
        if i < 0:
            break
        p, vals, closed = i + 1, [], False
        while p < n:
            while p < n and (buf[p] in ws or buf[p] == 0x2C):
                p += 1
            if p < n and buf[p] == 0x5D:
                closed, p = True, p + 1
                break
            t0 = p
            while p < n and buf[p] not in (0x2C, 0x5D) and buf[p] not in ws and p - t0 < 63:
                p += 1
            if p >= n:
                break                                   # cut off by the end of the buffer: left for the next call
            tok = buf[t0:p]
            if len(tok) == 63 and buf[p] not in (0x2C, 0x5D) and buf[p] not in ws:
                return None
            if any(c not in b"0123456789+-.eE" for c in tok):
                return None
            try:
                v = float(tok)
            except ValueError:
                return None
            with np.errstate(over="ignore"):
                v = np.float32(v)
            if not np.isfinite(v) or len(vals) >= d:
                return None
            vals.append(v)
        if not closed:
            break
        if len(vals) != d:
            return None
        rows.append(vals)
        pos = done = p
    return np.array(rows, dtype=np.float32).reshape(len(rows), d), done


def test_pgvector_parser_agrees_with_a_restatement_on_random_text():
    """Differential test of the C parser on seeded random text - well-formed rows, rows cut anywhere, stray brackets, signs,
    exponents, over-long literals, NUL bytes: the same rows and the same resume offset, or a refusal where the restatement
    refuses.  (TS_FUZZ_CASES raises the case count; run once under a host AddressSanitizer build, profiles/HISTORY.md.)"""
    import ctypes as C
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    rng = np.random.default_rng(20261005)
    alphabet = [b"[", b"]", b",", b" ", b"\n", b"\t", b"1", b"23", b"0.5", b"-", b"+", b".", b"e", b"E-3", b"7e2", b"x", b"\x00", b"9" * 70,
                b"1e50", b"nan", b"[1,2,3]", b"[4.25, -1e-3,0]", b"17\t"]
    cases = int(os.environ.get("TS_FUZZ_CASES", "400"))
    refused = parsed = 0
    for _ in range(cases):
        d = int(rng.integers(1, 5))
        if rng.random() < 0.5:                          # mostly well-formed rows of this d, then damaged
            rows = [b"%d\t[" % i + b",".join(repr(float(np.float32(v))).encode() for v in rng.standard_normal(d)) + b"]\n"
                    for i in range(int(rng.integers(0, 6)))]
            buf = bytearray(b"".join(rows))
            for _ in range(int(rng.integers(0, 3))):
                if buf:
                    at = int(rng.integers(len(buf)))
                    buf[at:at + int(rng.integers(0, 2))] = alphabet[int(rng.integers(len(alphabet)))]
            buf = bytes(buf[: int(rng.integers(0, len(buf) + 1))]) if rng.random() < 0.5 else bytes(buf)
        else:
            buf = b"".join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), size=int(rng.integers(0, 40))))
        max_rows = int(rng.integers(0, 8))
        out = np.full((max_rows + 1, d), np.float32(-77.0))           # one row more than allowed: must stay untouched
        nrows, used = C.c_int64(-1), C.c_int64(-1)
        rc = lib.ts_parse_pgvector_text(buf, len(buf), d, _ffi.as_ptr(out), max_rows, C.byref(nrows), C.byref(used))
        want = _parse_vectors_restated(buf, d, max_rows)
        if want is None:
            assert rc != 0, buf
            refused += 1
        else:
            assert rc == 0, (buf, _ffi.last_error() if hasattr(_ffi, "last_error") else rc)
            assert nrows.value == want[0].shape[0] and used.value == want[1], (buf, nrows.value, used.value, want)
            assert np.array_equal(out[: nrows.value], want[0]), buf
            parsed += 1
        assert np.all(out[max_rows] == np.float32(-77.0)), buf
    assert refused > cases // 20 and parsed > cases // 4           # the generator reaches both outcomes


def test_showcase_loaders(tmp_path, monkeypatch):
    """app_showcase_model.load_model / load_embedding_library (app_showcase_model.py:32-58): the model or the error that
    stopped it (the app displays it), the library or (None, None)."""
    import pickle
    import torch
    from theoremsearch_amd import app_showcase_model as asm
    monkeypatch.delenv("TS_ALLOW_RANDOM_ENCODER", raising=False)
    monkeypatch.delenv("TS_MODEL_DIR", raising=False)
    with pytest.raises(Exception):
        asm.load_model()                                                # no checkpoint offline
    assert asm.load_embedding_library(str(tmp_path)) == (None, None)
    emb = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    torch.save(emb, tmp_path / "corpus_embeddings.pt")
    with open(tmp_path / "theorems_data.pkl", "wb") as f:
        pickle.dump([{"type": "theorem"}] * 3, f)
    got, data = asm.load_embedding_library(str(tmp_path))
    assert torch.equal(got, emb) and len(data) == 3
    assert asm.EMBEDDING_LIBRARY_DIR == "./app_embeds"


def test_ndcg_says_too_small_where_the_reference_does(capsys):
    """compare_embeddings.py:201-203: an empty relevance list (a query without qrels: its ideal ranking is empty) prints
    "TOO SMALL" and scores 0 - part of the printed report the mirror promises to reproduce."""
    import numpy as np
    from theoremsearch_amd import compare_embeddings as ce
    sim = np.random.default_rng(0).standard_normal((2, 6)).astype(np.float32)
    val = ce.ndcg_at_k(sim, {0: {1: 2.0}, 1: {}}, k=3)
    out = capsys.readouterr().out
    assert out.count("TOO SMALL") == 1 and 0.0 <= val <= 1.0
