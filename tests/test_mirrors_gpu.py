"""The reference-shaped host modules on a real GPU: every search below goes through libtsearch.so."""
import json
import datetime
import os
import socket

import numpy as np
import pytest

from conftest import load_json
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def encoder():
    from theoremsearch_amd.encoder import SentenceEncoder
    import torch
    # random-init BERT-base-shaped stand-in on cuda:0, in the opt-in bf16 (the tests below were written for its kernels; the fp32
    # default - what the reference runs - is covered by the dtype=torch.float32 legs, tests/test_fulldepth_gpu.py and the long-text test)
    return SentenceEncoder(num_layers=2, allow_random_init=True, dtype=torch.bfloat16)


def test_cos_sim_and_semantic_search_match_oracle():
    from theoremsearch_amd import util
    q, c = oracle.inputs(2000, 9, 768, 61, "cos")
    got = util.cos_sim(q, c)
    assert np.max(np.abs(got - oracle.cos_sim(q, c))) <= 1e-5
    assert util.cos_sim(q[0], c[:5]).shape == (1, 5)                      # 1-D promoted like util.cos_sim
    scores, idx = util.semantic_search(q, c, top_k=5)
    truth = oracle.scores_fp64(*oracle.prepared_inputs(q, c, "cos", "f32"))
    oracle.check_topk_against_truth(truth, idx, scores, 5)


def test_encoder_similarity_is_the_cosine_matrix(encoder):
    """SentenceEncoder.similarity = SentenceTransformer.similarity as experiments/first_experiment.py:195,205 call it: the cosine
    matrix of two embedding sets (1-D promoted), a torch tensor; a checkpoint that names the dot product gets that."""
    import torch
    a, b = oracle.inputs(300, 5, 768, 17, "cos")
    got = encoder.similarity(a, b)
    assert isinstance(got, torch.Tensor) and tuple(got.shape) == (5, 300) and got.dtype == torch.float32
    assert np.max(np.abs(got.numpy() - oracle.cos_sim(a, b))) <= 1e-5
    assert tuple(encoder.similarity(a[0], torch.from_numpy(b[:7])).shape) == (1, 7)
    emb = encoder.encode(["a tree on n vertices has n-1 edges", "every bounded sequence has a convergent subsequence"])
    self_sim = encoder.similarity(emb, emb).numpy()
    assert np.allclose(np.diag(self_sim), 1.0, atol=1e-5)
    old = encoder.pipeline.similarity_fn_name
    try:
        encoder.pipeline.similarity_fn_name = "dot"
        assert np.max(np.abs(encoder.similarity(a, b).numpy() - a.astype(np.float64) @ b.astype(np.float64).T)) <= 1e-4
        encoder.pipeline.similarity_fn_name = "manhattan"
        with pytest.raises(NotImplementedError):
            encoder.similarity(a, b)
    finally:
        encoder.pipeline.similarity_fn_name = old


def test_search_and_display_through_libtsearch_returns_the_hits_the_reference_displays():
    """The showcase app's search lines (app_showcase_model.py:92-129) through the filtered search of libtsearch: the hits the
    reference's own function displayed for every sidebar state of tests/golden/showcase.json (rows and :.4f similarities),
    with the matrix and with an index kept across calls."""
    import json
    from callsites_common import StubModel, results_from_calls, results_of
    from theoremsearch_amd import TheoremIndex, app_showcase_model
    case = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "showcase.json")))
    model = StubModel(case["seed"], case["d"])
    data = case["theorems_data"]
    db = model.encode([t["text_to_embed"] for t in data])
    with TheoremIndex.from_embeddings(db, metric="cos") as ix:
        for name, state in case["states"].items():
            f = dict(state["filters"], citation_range=tuple(state["filters"]["citation_range"]))
            if f["year_range"] is not None:
                f["year_range"] = tuple(f["year_range"])
            want = results_from_calls(state["calls"], data)
            for corpus in (db, ix):
                assert results_of(app_showcase_model.search_and_display(case["query"], model, data, corpus, f), data) == want, name


def test_fused_pooling_epilogue_matches_torch():
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    for dtype in (torch.float32, torch.bfloat16):
        # widths of the vector form (d / 8 or d / 4 vectors per row: 48 ... 256 -> 5 ... 1 token groups) and one it does not
        # take (d = 100: not a multiple of the bf16 vector -> the general kernel); a long sequence takes the general kernel too
        for d, S in ((768, 19), (1024, 19), (384, 19), (100, 7), (256, 1100)):
            hidden = torch.randn((37, S, d), generator=g).to(dtype).cuda()
            lens = torch.randint(1, S + 1, (37,), generator=g)
            mask = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64).cuda()
            hf, mf = hidden.double(), mask.unsqueeze(-1).double()          # fp64 reference of the same inputs
            refs = {0: (hf * mf).sum(1) / mf.sum(1).clamp(min=1e-9),
                    1: hf[torch.arange(37, device="cuda"), mask.sum(1) - 1], 2: hf[:, 0]}
            for pooling, ref in refs.items():
                for normalize in (0, 1):
                    want = (torch.nn.functional.normalize(ref, p=2, dim=1) if normalize else ref).float()
                    out = torch.empty((37, d), dtype=torch.float32, device="cuda")
                    _ffi.check(lib.ts_pool_normalize(0, C.c_void_p(hidden.data_ptr()), 1 if dtype == torch.bfloat16 else 0,
                                                     C.c_void_p(mask.data_ptr()), 37, S, d, pooling, normalize,
                                                     C.c_void_p(out.data_ptr()), 0, d,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                    torch.cuda.synchronize()
                    assert torch.allclose(out, want, atol=1e-5, rtol=1e-5), (dtype, d, pooling, normalize, (out - want).abs().max().item())
                    outb = torch.empty((37, d), dtype=torch.bfloat16, device="cuda")
                    _ffi.check(lib.ts_pool_normalize(0, C.c_void_p(hidden.data_ptr()), 1 if dtype == torch.bfloat16 else 0,
                                                     C.c_void_p(mask.data_ptr()), 37, S, d, pooling, normalize,
                                                     C.c_void_p(outb.data_ptr()), 1, d,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                    torch.cuda.synchronize()
                    assert torch.allclose(outb.float(), want, atol=1e-5, rtol=2 ** -7), (dtype, d, pooling, normalize)   # one bf16 rounding


def test_pgvector_adapter_matches_sql_semantics():
    import theoremsearch_amd as ts
    from theoremsearch_amd import pgvector
    rng = np.random.default_rng(71)
    e = oracle.l2_normalize(rng.standard_normal((5000, 1024)).astype(np.float32))     # qwen-sized rows, stored normalised
    qv = oracle.l2_normalize(rng.standard_normal((1, 1024)).astype(np.float32))[0]
    cit = [None if i % 7 == 0 else int(i % 300) for i in range(5000)]
    with ts.TheoremIndex.from_embeddings(e, dtype="f32", metric="ip") as ix:
        rows = pgvector.search(ix, qv, 10)
        want_i, want_sim = oracle.pgvector_search(qv, e, 10)
        assert [r["row"] for r in rows] == want_i.tolist()
        assert np.allclose([r["similarity"] for r in rows], want_sim, atol=1e-5)     # 1 + <e, q>
        # and the sequential fp32 scan of the C restatement of pgvector's "<#>" (oracle/pgvector_ip.c)
        import subprocess
        from conftest import ROOT
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        c_rows, c_sim = oracle.pgvector_c_search(qv, e, 10)
        truth = oracle.scores_fp64(qv[None, :], e)
        oracle.check_topk_against_truth(truth, np.array([[r["row"] for r in rows]]), None, 10)
        oracle.check_topk_against_truth(truth, c_rows[None, :], None, 10)
        assert np.allclose([r["similarity"] for r in rows], c_sim, atol=1e-5)
        rows_w = pgvector.search(ix, qv, 5, citation_weight=0.02, citations=cit)
        pool_i, pool_sim = oracle.pgvector_search(qv, e, oracle.pool_size(5))
        ri, rs, rw = oracle.citation_weighted_rerank(pool_i, pool_sim, [cit[i] for i in pool_i], 0.02, 5)
        assert [r["row"] for r in rows_w] == ri.tolist()
        assert np.allclose([r["score"] for r in rows_w], rw, atol=1e-5)


def test_encoder_to_index_without_host_hop(encoder):
    import theoremsearch_amd as ts
    from theoremsearch_amd import generate_embeddings as ge
    texts = [f"Theorem {i}: every finite group of order {i} has property P_{i % 5}." for i in range(300)]
    with ts.TheoremIndex(len(texts), 768, dtype="f32", metric="cos") as ix:
        ge.embed_into_index(encoder, ix, texts[:128], 0, batch_size=16)
        ge.embed_into_index(encoder, ix, texts[128:], 128, batch_size=16)
        stored = ix.download()
        host = np.array(ge.embed_texts(encoder, texts, batch_size=16), dtype=np.float32)
        assert np.allclose(stored, host, atol=2e-2)       # bf16 encoder forward: same rows up to batch-shape noise
        assert np.allclose(np.linalg.norm(stored, axis=1), 1.0, atol=1e-5)
        scores, idx = ix.search(stored[:40], 1)
        assert idx[:, 0].tolist() == list(range(40)) and np.allclose(scores[:, 0], 1.0, atol=1e-5)


def test_generate_embeddings_driver_upserts_pages(encoder):
    import theoremsearch_amd as ts
    from theoremsearch_amd import generate_embeddings as ge
    rows = [{"slogan_id": i, "slogan": f"Slogan {i}: a bound for the {i}-th eigenvalue."} for i in range(200)]
    pages = [rows[i:i + 128] for i in range(0, 200, 128)]                     # page size 128 (__main__.py:75)
    with ts.TheoremIndex(200, 768, dtype="f32", metric="cos") as ix:
        assert ge.generate_embeddings(pages, "gemma", ix, embedder=encoder) == 200
        assert ge.generate_embeddings(pages, "gemma", ix, embedder=encoder) == 0   # NOT EXISTS filter: nothing left
        first = ix.download()
        changed = [dict(rows[5], slogan="A completely different statement.")]
        assert ge.generate_embeddings([changed], "gemma", ix, embedder=encoder, overwrite=True) == 1
        second = ix.download()
        assert not np.allclose(first[5], second[5], atol=1e-3) and np.array_equal(first[6:], second[6:])
        q = np.array(ge.embed_texts(encoder, [rows[77]["slogan"]]), dtype=np.float32)
        scores, idx = ix.search(q, 1)
        assert idx[0, 0] == 77


def test_concurrent_searches_on_one_handle():
    # Streamlit runs every session on its own thread against one shared library (streamlit_app.py:52)
    import threading
    import theoremsearch_amd as ts
    q, c = oracle.inputs(60_000, 64, 768, 91, "ip")
    want_s, want_i = oracle.search(q, c, 5, "ip", "bf16")
    errors = []
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        def worker(t):
            try:
                for rep in range(5):
                    lo = (7 * t + rep) % 60
                    s, i = ix.search(q[lo:lo + 1 + t], 5)
                    assert np.array_equal(i, want_i[lo:lo + 1 + t]), (t, rep)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
        [t.start() for t in threads]
        [t.join() for t in threads]
    assert not errors, errors


def test_embedding_library_round_trip(tmp_path, monkeypatch, encoder):
    from theoremsearch_amd import app_create_embeddings as ace
    papers = tmp_path / "app_papers"
    papers.mkdir()
    for p in range(6):
        doc = {"title": f"Paper {p}", "url": f"http://example.org/{p}", "authors": ["A"], "citations": p,
               "primary_math_tag": "math.AG", "year": 2020 + p, "source": "arXiv", "journal_published": bool(p % 2),
               "global_notations": f"Notation {p}", "global_definitions": "", "global_assumptions": "",
               "theorems": [{"type": "theorem", "content": f"Statement {p}.{t} about $X_{t}$."} for t in range(20)]}
        (papers / f"p{p}.json").write_text(json.dumps(doc))
    monkeypatch.setattr(ace, "PARSED_PAPERS_DIR", str(papers))
    monkeypatch.setattr(ace, "OUTPUT_DIR", str(tmp_path / "app_embeds"))
    ace.create_embedding_library(model=encoder)
    emb, data = ace.load_embedding_library(str(tmp_path / "app_embeds"))
    assert tuple(emb.shape) == (120, 768) and len(data) == 120
    ix, data2 = ace.load_embedding_index(str(tmp_path / "app_embeds"))
    with ix:
        qe = encoder.encode(data[17]["text_to_embed"], convert_to_tensor=True)          # app_showcase_model.py:92
        scores, idx = ix.search(qe, 5)
        c = emb.numpy()
        truth = oracle.scores_fp64(*oracle.prepared_inputs(qe.cpu().numpy(), c, "cos", "f32"))
        oracle.check_topk_against_truth(truth, idx, scores, 5, score_tol=1e-4)
        assert data2[int(idx[0, 0])]["paper_title"] == data[17]["paper_title"]


def test_evaluate_retrieval_report(capsys, encoder):
    from theoremsearch_amd import compare_embeddings as ce
    theorems = [(f"Slogan {i}: the moduli stack M_{i} is smooth.", f"p{i % 9}") for i in range(200)]
    queries = [(theorems[i][0], theorems[i][1]) for i in range(0, 200, 10)]
    qrels = ce._generate_qrels(queries, theorems)
    for qi in range(len(queries)):
        qrels[qi][qi * 10] = 1
    ce.evaluate_retrieval(encoder, theorems, queries, qrels, 5)
    out = capsys.readouterr().out
    assert "Cos-sim matrix dim (20, 200)" in out and "P@1 | 1.0" in out and "H@5 | 1.0" in out and "nDCG@5 |" in out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gpu_rank(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))       # both ranks share the one GPU of the box
    try:
        from theoremsearch_amd.distributed import ShardedSearcher, shard_bounds
        q, c = oracle.inputs(30_000, 12, 768, 81, "ip")
        lo, hi = shard_bounds(len(c), world, rank)
        searcher = ShardedSearcher.from_local_rows(c[lo:hi], len(c), dtype="bf16", metric="ip")
        scores, idx = searcher.search(q, 10)
        truth = oracle.scores_fp64(*oracle.prepared_inputs(q, c, "ip", "bf16"))
        stats = oracle.check_topk_against_truth(truth, idx, scores, 10)
        # rank of a document over the sharded corpus = its position in the whole ranking
        rows = np.array([int(idx[0, 0]), int(idx[1, 9]), 3, len(c) - 1, len(c) // 2, len(c) // 2 - 1] + [7] * 6)
        got_ranks = searcher.rank_of(q, rows)
        want_ranks = oracle.rank_of(truth, rows)
        close = np.array([np.sum(np.abs(truth[i] - truth[i, r]) <= 1e-6) - 1 for i, r in enumerate(rows)])
        ranks_ok = bool(np.all(np.abs(got_ranks - want_ranks) <= close)) and got_ranks[0] == 0 and got_ranks[1] == 9
        # metadata filter over the whole corpus, applied slice by slice
        mask = np.random.default_rng(5).random(len(c)) < 0.3
        ms, mi = searcher.search(q, 10, mask=mask)
        keep = np.flatnonzero(mask)
        mstats = oracle.check_topk_against_truth(truth[:, keep], np.searchsorted(keep, mi), ms, 10)
        ret[rank] = stats["recall"] if (ranks_ok and mask[mi].all() and mstats["recall"] == 1.0) else -1.0
    finally:
        dist.destroy_process_group()


def test_sharded_hip_search_two_ranks():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_rank, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert dict(ret) == {0: 1.0, 1: 1.0}


def test_index_from_a_pgvector_copy_stream_searches_like_the_matrix():
    """COPY-text rows -> index (pgvector.index_from_copy_stream) answers exactly like an index built from the matrix."""
    import theoremsearch_amd as ts
    from theoremsearch_amd import pgvector
    n, d = 3000, 768
    q, c = oracle.inputs(n, 3, d, 17, "ip")
    txt = "".join(f"{i}\t[{','.join(repr(float(v)) for v in row)}]\n" for i, row in enumerate(c)).encode()
    chunks = [txt[lo:lo + 1_000_003] for lo in range(0, len(txt), 1_000_003)]
    with pgvector.index_from_copy_stream(chunks, n, d) as ix, ts.TheoremIndex.from_embeddings(c, metric="ip") as ref:
        assert np.array_equal(ix.download(), ref.download())
        s1, i1 = ix.search(q, 10)
        s2, i2 = ref.search(q, 10)
        assert np.array_equal(i1, i2) and np.array_equal(s1, s2)


def test_two_threads_share_one_index():
    """a handle is thread-safe (internal mutex; ctypes releases the GIL)."""
    import threading
    import theoremsearch_amd as ts
    n, d = 40000, 768
    q, c = oracle.inputs(n, 64, d, 23, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        want = [ix.search(q[:1], 10), ix.search(q, 10)]
        errors = []

        def worker(kind):
            try:
                for _ in range(25):
                    s, i = ix.search(q[:1], 10) if kind == 0 else ix.search(q, 10)
                    if not (np.array_equal(i, want[kind][1]) and np.array_equal(s, want[kind][0])):
                        errors.append(kind)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(k,)) for k in (0, 1, 0, 1)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors


def test_bench_prints_one_json_line_with_the_contract_keys():
    """bench.py's stdout is exactly one JSON line carrying the driver's keys plus roofline and cpu_baseline."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rows", "300000", "--nq", "16", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=280, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in doc, key
    assert doc["n_gpus"] == 1 and doc["steps"] == 2 and doc["warmup"] == 1 and doc["higher_is_better"] is True
    assert doc["unit"] == "queries/s" and doc["dtype"] == "bf16" and doc["data"] == "synthetic" and doc["vs_baseline"] is None
    assert "workload" in doc["config"] and "model" not in doc["config"]
    assert doc["value"] == pytest.approx(16 / (doc["ms_per_step"] * 1e-3), rel=1e-3)
    roof = doc["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-4) and roof["achieved"] > 0
    assert roof["hbm_frac"] > 0.05 and roof["kernel_ms"] < 1.0, roof     # 300k x 768 bf16 rows: a fraction of a millisecond
    assert 0 < roof["mfma_frac"] < 1 and doc["parity"]["violations"] == 0 and doc["parity"]["queries_checked"] == 16
    cpu = doc["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]
    assert doc["recall_at_10"] == 1.0


def test_pgvector_search_with_a_where_mask():
    """ORDER BY <#> LIMIT k behind a WHERE clause: pgvector.search(mask=sql_filter_mask(...)) returns what the oracle's
    pgvector scan returns over exactly the rows the clause keeps - plain and citation-weighted."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from filters_common import make_sql_rows, sql_filter_states
    import theoremsearch_amd as ts
    from theoremsearch_amd import filters as flt
    from theoremsearch_amd import pgvector
    n, d = 5000, 768
    q, e = oracle.inputs(n, 1, d, 29, "ip")
    rows = make_sql_rows(n)
    cit = [r["citations"] for r in rows]
    with ts.TheoremIndex.from_embeddings(e, metric="ip") as ix:
        for name in ("arxiv_only", "preprint_known_citations", "everything"):
            f = sql_filter_states(top_k=7)[name]
            mask = flt.sql_filter_mask(rows, f)
            keep = np.flatnonzero(mask)
            got = pgvector.search(ix, q[0], 7, mask=mask)
            want_i, want_sim = oracle.pgvector_search(q[0], e[keep], 7)
            assert [g["row"] for g in got] == [int(keep[i]) for i in want_i], name
            assert np.allclose([g["similarity"] for g in got], want_sim, atol=1e-5)
            got_w = pgvector.search(ix, q[0], 5, citation_weight=0.05, citations=cit, mask=mask)
            pool_i, pool_sim = oracle.pgvector_search(q[0], e[keep], oracle.pool_size(5))
            ri, rs, rw = oracle.citation_weighted_rerank(keep[pool_i], pool_sim, [cit[int(keep[i])] for i in pool_i], 0.05, 5)
            assert [g["row"] for g in got_w] == [int(i) for i in ri], name
            assert np.allclose([g["score"] for g in got_w], rw, atol=1e-5)


def test_view_handles_search_the_same_rows_concurrently():
    """ts_index_view: a second handle on the same rows (own stream, scratch, lock); two threads, one per handle, get
    the answers of the owning handle; a view is read-only."""
    import threading
    import theoremsearch_amd as ts
    n, d = 50000, 768
    q, c = oracle.inputs(n, 48, d, 37, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos", row_offset=77) as ix:
        want = ix.search(q, 10)
        with ix.view() as v:
            assert (v.n, v.d, v.row_offset) == (ix.n, ix.d, 77)
            bad = []

            def worker(h):
                for _ in range(20):
                    s, i = h.search(q, 10)
                    if not (np.array_equal(i, want[1]) and np.array_equal(s, want[0])):
                        bad.append(1)

            threads = [threading.Thread(target=worker, args=(h,)) for h in (ix, v)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert not bad
            with pytest.raises(ts.TSearchError):
                v.upload(c[:1], 0)


# ---- call-site bodies of the reference, executed by oracle/gen_golden.py -> tests/golden/callsites.json ----------------
def test_callsite_mirrors_reproduce_the_reference_output_through_the_hip_path(capsys):
    """compare_embeddings / evaluate_retrieval through libtsearch print exactly what the reference's own function bodies
    printed on the same model outputs (exact score ties included); search_theorems returns the hits the reference displayed."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from callsites_common import StubModel, results_from_calls, results_of, run_compare, run_evaluate
    from conftest import load_json
    from theoremsearch_amd import app_scratchpad, compare_embeddings as ce
    cases = load_json("callsites.json")["cases"]
    assert run_compare(ce, cases["compare_embeddings"], capsys) == cases["compare_embeddings"]["stdout"]
    assert run_evaluate(ce, cases["evaluate_retrieval"], capsys) == cases["evaluate_retrieval"]["stdout"]
    case = cases["search_theorems"]
    model = StubModel(case["seed"], case["d"])
    data = case["theorems_data"]
    db = model.encode([t["text_to_embed"] for t in data])
    want = results_from_calls(case["calls"], data)                               # the five hits the reference displayed
    assert results_of(app_scratchpad.search_theorems(case["query"], model, data, db), data) == want   # the matrix, as the app passes it
    import theoremsearch_amd as ts
    with ts.TheoremIndex.from_embeddings(db, metric="cos") as ix:               # or an index kept across calls
        assert results_of(app_scratchpad.search_theorems(case["query"], model, data, ix), data) == want


def test_encode_multi_process_replicas_on_the_gpu(encoder):
    """ec2/generate_embeddings/embeddings.py:32-38 fans a page out over devices; here two replicas share the one GPU of
    the box (one process per replica): rows come back in input order and equal the single-process embeddings (both run
    the same seeded weights in bf16; the fused pooling kernel is deterministic)."""
    from theoremsearch_amd import generate_embeddings as ge
    texts = [f"Theorem {i}: every tree on $n_{i}$ vertices has $n_{i} - 1$ edges " + "and more " * (i % 7) for i in range(61)]
    want = encoder.encode(texts, batch_size=16, normalize_embeddings=True)
    pool = encoder.start_multi_process_pool(["cuda:0", "cuda:0"])
    try:
        got = encoder.encode_multi_process(texts, pool=pool, batch_size=16, normalize_embeddings=True)
    finally:
        encoder.stop_multi_process_pool(pool)
    assert got.shape == want.shape == (61, 768)
    assert np.allclose(got, want, atol=2e-3)           # bf16 forward: batch composition differs between the two splits
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-4)
    # the mirror of embed_texts takes the multi-process branch for a page of at least batch_size texts
    out = ge.embed_texts(encoder, texts, batch_size=16)
    assert isinstance(out, list) and len(out) == 61 and np.allclose(np.array(out, dtype=np.float32), want, atol=2e-3)


def test_generate_embeddings_appends_new_slogan_ids(encoder):
    """The growing form of the upsert loop (ec2/generate_embeddings/__main__.py:85-99): a slogan_id the slot map does not
    know is an INSERT (appended behind the last row), a known one is skipped, or re-embedded in place with overwrite."""
    import theoremsearch_amd as ts
    from theoremsearch_amd import generate_embeddings as ge
    rows = [{"slogan_id": 1000 + 7 * i, "slogan": f"Slogan {i}: a tree on {i} vertices has {i - 1} edges."} for i in range(150)]
    pages = [rows[i:i + 64] for i in range(0, 150, 64)]
    slots = {}
    with ts.TheoremIndex(0, 768, dtype="bf16", metric="ip") as ix:
        assert ge.generate_embeddings(pages[:2], "gemma", ix, embedder=encoder, slots=slots) == 128
        assert ix.n == 128 and slots[1000] == 0 and slots[1000 + 7 * 127] == 127
        assert ge.generate_embeddings(pages, "gemma", ix, embedder=encoder, slots=slots) == 22       # only the new ids
        assert ix.n == 150 and len(slots) == 150
        want = encoder.encode([r["slogan"] for r in rows], normalize_embeddings=True, batch_size=16)
        scores, idx = ix.search(want[[3, 140]], 1)
        assert idx[:, 0].tolist() == [slots[rows[3]["slogan_id"]], slots[rows[140]["slogan_id"]]] and (scores[:, 0] > 0.99).all()
        before = ix.download(5, 1).copy()
        rows[5]["slogan"] = "Completely different text about schemes."
        assert ge.generate_embeddings([rows[:8]], "gemma", ix, embedder=encoder, slots=slots, overwrite=True) == 8
        assert ix.n == 150 and not np.array_equal(ix.download(5, 1), before)


def test_add_layernorm_kernel_matches_torch_in_fp64():
    """ts_add_layernorm = LayerNorm(a + b) * gamma + beta with the sum, mean and variance in fp32, against the same
    expression in fp64 on the same (rounded) inputs: fp32 and bf16, the encoder widths, a ragged row count, in place."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(4)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for d in (384, 768, 1024):
            a = (torch.randn((1003, d), generator=g) * 2.0 + 0.3).to(dtype).cuda()
            b = torch.randn((1003, d), generator=g).to(dtype).cuda()
            gamma = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dtype).cuda()
            beta = (0.1 * torch.randn(d, generator=g)).to(dtype).cuda()
            want = torch.nn.functional.layer_norm(a.double() + b.double(), (d,), gamma.double(), beta.double(), 1e-12)
            out = torch.empty_like(a)
            args = (C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), 1e-12, 1003, d, 1 if dtype == torch.bfloat16 else 0)
            _ffi.check(lib.ts_add_layernorm(0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), *args, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            assert torch.allclose(out.double(), want, atol=tol, rtol=tol), (dtype, d, (out.double() - want).abs().max().item())
            _ffi.check(lib.ts_add_layernorm(0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), *args, C.c_void_p(a.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))       # in place over `a`
            torch.cuda.synchronize()
            assert torch.equal(a, out)
    with pytest.raises(_ffi.TSearchError):
        _ffi.check(lib.ts_add_layernorm(0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(gamma.data_ptr()),
                                        C.c_void_p(beta.data_ptr()), 1e-12, 4, 10, 1, C.c_void_p(out.data_ptr()), None))


def test_embed_layernorm_kernel_matches_the_modules_own(encoder):
    """ts_embed_layernorm = BertEmbeddings (word + token type + position, LayerNorm) against the same expression in fp64 on the
    same tables: fp32 and bf16, token types given and absent, a ragged token count, ids at both ends of the tables."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(9)
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for d in (384, 768, 1024):
            V, P, T, B, S = 1000, 64, 2, 7, 19
            word = torch.randn((V, d), generator=g).to(dtype).cuda()
            pos = torch.randn((P, d), generator=g).to(dtype).cuda()
            typ = torch.randn((T, d), generator=g).to(dtype).cuda()
            gamma = (1.0 + 0.1 * torch.randn(d, generator=g)).to(dtype).cuda()
            beta = (0.1 * torch.randn(d, generator=g)).to(dtype).cuda()
            ids = torch.randint(0, V, (B, S), generator=g)
            ids[0, 0], ids[0, 1] = 0, V - 1
            ids = ids.cuda()
            for tt in (None, torch.randint(0, T, (B, S), generator=g).cuda()):
                x = word.double()[ids] + (typ.double()[tt] if tt is not None else typ.double()[0]) + pos.double()[torch.arange(S)][None]
                want = torch.nn.functional.layer_norm(x, (d,), gamma.double(), beta.double(), 1e-12)
                out = torch.empty((B, S, d), dtype=dtype, device="cuda")
                _ffi.check(lib.ts_embed_layernorm(0, C.c_void_p(ids.data_ptr()), C.c_void_p(tt.data_ptr()) if tt is not None else None,
                                                  C.c_void_p(word.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(typ.data_ptr()),
                                                  V, P, T, C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()), 1e-12, B * S, S, d,
                                                  1 if dtype == torch.bfloat16 else 0, C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                torch.cuda.synchronize()
                assert torch.allclose(out.double(), want, atol=tol, rtol=tol), (dtype, d, (out.double() - want).abs().max().item())
    with pytest.raises(_ffi.TSearchError):
        _ffi.check(lib.ts_embed_layernorm(0, C.c_void_p(ids.data_ptr()), None, C.c_void_p(word.data_ptr()), C.c_void_p(pos.data_ptr()),
                                          C.c_void_p(typ.data_ptr()), V, P, T, C.c_void_p(gamma.data_ptr()), C.c_void_p(beta.data_ptr()),
                                          1e-12, 4, 2, 10, 1, C.c_void_p(out.data_ptr()), None))
    # the module of the encoder itself: the fused forward's input layer against BertEmbeddings
    enc = {k: v.cuda() for k, v in encoder._tokenize(["Let $G$ be a finite group.", "Every bounded sequence has a convergent subsequence."]).items()}
    with torch.inference_mode():
        want = encoder.model.embeddings(input_ids=enc["input_ids"], token_type_ids=enc.get("token_type_ids")).float()
        got = encoder._fused._embed(enc["input_ids"], enc.get("token_type_ids")).float()
    assert torch.allclose(got, want, atol=4e-2, rtol=4e-2), (got - want).abs().max().item()


def test_short_sequence_attention_kernel_matches_fp64():
    """ts_attention_short = softmax(Q K^T / 8 + key mask) V per (sequence, head) from the fused projection's layout, against the
    same expression in fp64 on the same bf16 inputs: every tile count (1 .. 64 tokens: all score tiles at once; 65 .. 128: one
    query tile at a time; ragged lengths), with and without a key
    mask (every sequence keeps at least its first token), a head count that does not fill the last workgroup; and against
    torch's scaled_dot_product_attention; longer sequences and other head sizes are refused."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(11)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B, S, H in ((3, 1, 2), (5, 7, 3), (4, 16, 12), (9, 19, 5), (6, 32, 12), (3, 33, 2), (2, 48, 3), (5, 61, 7), (2, 64, 12),
                    (3, 65, 5), (2, 80, 12), (5, 81, 3), (2, 96, 7), (3, 100, 2), (2, 112, 6), (4, 127, 3), (3, 128, 12)):
        qkv = (torch.randn((B, S, 3, H, 64), generator=g) * 1.5).to(torch.bfloat16).cuda()
        lens = torch.randint(1, S + 1, (B,), generator=g)
        km = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64).cuda()
        for mask in (None, km):
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).double() for i in range(3))            # [B][H][S][64]
            sc = q @ k.transpose(-1, -2) / 8.0
            if mask is not None:
                sc = sc.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
            want = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, S, H * 64)
            out = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device="cuda")
            _ffi.check(lib.ts_attention_short(0, C.c_void_p(qkv.data_ptr()), C.c_void_p(mask.data_ptr()) if mask is not None else None,
                                             B, S, H, 64, C.c_void_p(out.data_ptr()), st))
            torch.cuda.synchronize()
            err = (out.double() - want).abs().max().item()
            assert err <= 3e-2, (B, S, H, mask is not None, err)                               # bf16 probabilities and output
            bias = None if mask is None else torch.zeros((B, 1, 1, S), dtype=torch.bfloat16, device="cuda").masked_fill_(
                mask[:, None, None, :] == 0, float("-inf"))
            ref = torch.nn.functional.scaled_dot_product_attention(*(qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)), attn_mask=bias)
            ref = ref.transpose(1, 2).reshape(B, S, H * 64)
            assert (out.float() - ref.float()).abs().max().item() <= 4e-2
    for S, hd in ((129, 64), (16, 32)):
        with pytest.raises(_ffi.TSearchError):
            _ffi.check(lib.ts_attention_short(0, C.c_void_p(qkv.data_ptr()), None, 1, S, 1, hd, C.c_void_p(out.data_ptr()), st))


def test_gqa_causal_attention_kernel_matches_fp64():
    """ts_attention_gqa = softmax(Q K^T / sqrt(128) + causal + key mask) V per (sequence, query head), query head h over key /
    value head h / (hq / hkv), from the stacked projection's layout [tokens][(hq + 2 hkv) * 128] (Qwen3Attention, the production
    embedder: streamlit_app.py:55), against the same expression in fp64 on the same bf16 inputs: every tile count up to 128
    tokens (up to 64: all score tiles at once; 65 .. 128: one query tile at a time), ragged lengths, padding on the LEFT (the Qwen tokenizer's side: the padding rows of a causal sequence have no allowed
    key and come back as zeros) and on the right, causal and not, 16 / 8 and 4 / 4 heads; longer sequences and other head sizes
    are refused."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(12)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B, S, HQ, HKV in ((3, 1, 2, 1), (5, 7, 4, 2), (4, 16, 16, 8), (9, 19, 6, 2), (6, 32, 16, 8), (3, 33, 4, 4), (2, 48, 6, 3),
                          (5, 61, 2, 2), (2, 64, 16, 8), (3, 65, 4, 2), (2, 80, 16, 8), (5, 81, 2, 1), (2, 96, 6, 3), (3, 100, 16, 8),
                          (2, 112, 4, 4), (4, 127, 2, 1), (3, 128, 16, 8), (2, 96, 8, 2), (3, 71, 12, 3)):
        qkv = (torch.randn((B, S, (HQ + 2 * HKV), 128), generator=g) * 1.2).to(torch.bfloat16).cuda()
        lens = torch.randint(1, S + 1, (B,), generator=g)
        right = (torch.arange(S)[None, :] < lens[:, None]).to(torch.int64).cuda()
        left = (torch.arange(S)[None, :] >= (S - lens)[:, None]).to(torch.int64).cuda()
        rep = HQ // HKV
        q = qkv[:, :, :HQ].permute(0, 2, 1, 3).double()                                           # [B][HQ][S][128]
        k = qkv[:, :, HQ:HQ + HKV].permute(0, 2, 1, 3).double().repeat_interleave(rep, dim=1)
        v = qkv[:, :, HQ + HKV:].permute(0, 2, 1, 3).double().repeat_interleave(rep, dim=1)
        for causal in (True, False):
            for mask in (None, right, left):
                sc = q @ k.transpose(-1, -2) / (128.0 ** 0.5)
                allow = torch.ones((B, 1, S, S), dtype=torch.bool, device="cuda")
                if causal:
                    allow = allow & torch.ones((S, S), dtype=torch.bool, device="cuda").tril_()[None, None]
                if mask is not None:
                    allow = allow & (mask[:, None, None, :] != 0)
                sc = sc.masked_fill(~allow, float("-inf"))
                p = torch.softmax(sc, dim=-1)
                p = torch.where(allow.any(dim=-1, keepdim=True), p, torch.zeros_like(p))          # a row without a key: zeros
                want = (p @ v).permute(0, 2, 1, 3).reshape(B, S, HQ * 128)
                out = torch.empty((B, S, HQ * 128), dtype=torch.bfloat16, device="cuda")
                _ffi.check(lib.ts_attention_gqa(0, C.c_void_p(qkv.data_ptr()), C.c_void_p(mask.data_ptr()) if mask is not None else None,
                                               B, S, HQ, HKV, 128, 1 if causal else 0, C.c_void_p(out.data_ptr()), st))
                torch.cuda.synchronize()
                err = (out.double() - want).abs().max().item()
                assert err <= 3e-2, (B, S, HQ, HKV, causal, mask is not None, err)                # bf16 probabilities and output
    for S, hd in ((129, 128), (16, 64)):
        with pytest.raises(_ffi.TSearchError):
            _ffi.check(lib.ts_attention_gqa(0, C.c_void_p(qkv.data_ptr()), None, 1, S, 2, 1, hd, 1, C.c_void_p(out.data_ptr()), st))


def test_fused_bert_forward_matches_the_models_own(encoder):
    """FusedBertForward (QKV as one GEMM, add + LayerNorm as one kernel) against the model's own forward on the same bf16
    weights: hidden states of the real tokens within bf16 noise, sentence embeddings within 2e-2 and cosine > 0.9995;
    fp32 weights: within 2e-4."""
    import torch
    from theoremsearch_amd.encoder import SentenceEncoder
    assert encoder._fused is not None
    texts = [f"Let $f_{i}$ be a continuous map of a compact space, number {i}. " * (1 + i % 3) for i in range(37)]
    enc = {k: v.cuda() for k, v in encoder._tokenize(texts).items()}
    with torch.inference_mode():
        want = encoder.model(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"]).last_hidden_state.float()
        got = encoder.forward_hidden(enc["input_ids"], enc["attention_mask"]).float()
    real = enc["attention_mask"].bool()
    assert (want - got)[real].abs().max().item() < 0.15 and (want - got)[real].abs().mean().item() < 0.01
    same = [texts[0]] * 5                                    # no padding: the attention runs without a mask
    es = {k: v.cuda() for k, v in encoder._tokenize(same).items()}
    assert bool(es["attention_mask"].all())
    with torch.inference_mode():
        w_ = encoder.model(input_ids=es["input_ids"], attention_mask=es["attention_mask"]).last_hidden_state.float()
        g_ = encoder.forward_hidden(es["input_ids"], es["attention_mask"], no_padding=True).float()
    assert (w_ - g_).abs().max().item() < 0.15 and (w_ - g_).abs().mean().item() < 0.01
    fused = encoder.encode(texts, normalize_embeddings=True, convert_to_numpy=True)
    os.environ["TS_ENCODER_FUSED"] = "0"
    try:
        plain_enc = SentenceEncoder(num_layers=2, allow_random_init=True, dtype=torch.bfloat16)
    finally:
        del os.environ["TS_ENCODER_FUSED"]
    assert plain_enc._fused is None
    plain = plain_enc.encode(texts, normalize_embeddings=True, convert_to_numpy=True)
    assert np.abs(fused - plain).max() < 2e-2 and np.min(np.sum(fused * plain, axis=1)) > 0.9995
    f32 = SentenceEncoder(num_layers=2, allow_random_init=True, dtype=torch.float32)
    assert f32._fused is not None
    e32 = {k: v.cuda() for k, v in f32._tokenize(texts[:9]).items()}
    with torch.inference_mode():
        w32 = f32.model(input_ids=e32["input_ids"], attention_mask=e32["attention_mask"]).last_hidden_state
        g32 = f32.forward_hidden(e32["input_ids"], e32["attention_mask"])
    assert (w32 - g32)[e32["attention_mask"].bool()].abs().max().item() < 2e-4


def test_bench_under_torch_distributed_run_with_two_ranks_sharing_the_gpu():
    """The driver's launch form for N > 1 (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N), rehearsed
    with two ranks on the box's one GPU (--share-gpu: gloo carries the exchange through host memory, RCCL refuses two ranks
    per GPU): shard bounds, the bare timed region + bracketed sustained leg, max-over-ranks timing, the merged parity check
    and the single JSON line of rank 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--rows", "600000",
           "--nq", "64", "--steps", "3", "--warmup", "1", "--sustained-steps", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["steps"] == 3 and doc["recall_at_10"] == 1.0 and doc["parity"]["violations"] == 0
    assert doc["config"]["rows"] == 600000 and "x2" in doc["config"]["parallelism"]
    assert doc["sustained"]["steps"] == 4 and doc["roofline"]["kernel_ms"] > 0 and doc["cpu_baseline"] is None


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 4` bare (no torch.distributed.run around it, WORLD_SIZE unset), the command shape the driver
    uses for its 2 / 4 / 8-GPU runs: the process starts the ranks as child processes itself, relays rank 0's single JSON line
    and its exit code; the line names what the communicator saw (backend, world, one device per rank), what one all-gather of
    the packed top-k costs, and every round of the exchange-placement trial.  FOUR ranks share the box's GPU here: with this
    test's own process that is five on the card, and a box allows six - the eight-rank launch is rehearsed without a device in
    tests/test_distributed_cpu.py (`--rehearse-launch`)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--share-gpu", "--rows", "800000", "--nq", "64",
           "--steps", "3", "--warmup", "1", "--sustained-steps", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 4 and doc["steps"] == 3 and doc["recall_at_10"] == 1.0 and doc["parity"]["violations"] == 0
    assert doc["parity"]["queries_checked"] == 64 and doc["config"]["rows"] == 800000 and "x4" in doc["config"]["parallelism"]
    ex = doc["exchange"]
    assert ex["world"] == 4 and ex["backend"] == "gloo" and ex["native"] is False and ex["launched_by"] == "bench.py"
    assert ex["allgather_us"] > 0 and len(ex["devices"]) == 4 and sorted(d_[0] for d_ in ex["devices"]) == [0, 1, 2, 3]
    assert ex["distinct_gpus"] == 1                      # the rehearsal: every rank on the box's one GPU
    trial = ex["placement_trial"]
    assert trial["rounds"] >= 3 and trial["rounds_discarded"] == 1
    assert len(trial["overlap_ms_per_step"]) == trial["rounds"] == len(trial["inline_ms_per_step"])
    assert ex["placement"].startswith("overlap") == (not trial["median_inline_ms"] < 0.98 * trial["median_overlap_ms"])
    # a sharded run times bare and takes its kernel time from a bracketed leg of its own
    assert doc["sustained"]["kernel_brackets"] is False and doc["sustained"]["bracket_leg"]["kernel_ms"] > 0
    assert doc["roofline"]["kernel_ms"] == doc["sustained"]["bracket_leg"]["kernel_ms"]
    # a rank that fails takes the job down with a non-zero exit code and no JSON line
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "600000", "--nq", "64", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert bad.returncode != 0 and not bad.stdout.strip(), (bad.returncode, bad.stdout)   # rank 1 has no GPU of its own here


# ---- the decoder-style encoder of the production app (Qwen3-Embedding shape): kernels around its GEMMs --------------------
def test_add_rmsnorm_kernel_matches_the_module_chain():
    """ts_add_rmsnorm against Qwen3DecoderLayer's own chain on the same inputs - `residual + x` rounded to the storage type,
    Qwen3RMSNorm (fp32 inside, rounded, times the weight) - bit for bit in bf16 and to fp32 noise in fp32; against fp64; without
    the addend (a plain norm) and in place over the residual."""
    import ctypes as C
    import torch
    from transformers.models.qwen3.modeling_qwen3 import Qwen3RMSNorm
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(14)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, code, tol in ((torch.bfloat16, 1, 2e-2), (torch.float32, 0, 2e-5)):
        for d in (1024, 768, 2048 if dtype == torch.bfloat16 else 512):
            rows = 777
            a = (torch.randn((rows, d), generator=g) * 1.7).to(dtype).cuda()
            b = torch.randn((rows, d), generator=g).to(dtype).cuda()
            norm = Qwen3RMSNorm(d, eps=1e-6).to("cuda", dtype=dtype)
            with torch.no_grad():
                norm.weight.copy_((1.0 + 0.2 * torch.randn(d, generator=g)).to(dtype))
                want_sum = a + b
                want = norm(want_sum)
                plain = norm(a)
            out_sum, out = torch.empty_like(a), torch.empty_like(a)
            _ffi.check(lib.ts_add_rmsnorm(0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(norm.weight.data_ptr()), 1e-6,
                                          rows, d, code, C.c_void_p(out_sum.data_ptr()), C.c_void_p(out.data_ptr()), st))
            out_plain = torch.empty_like(a)
            _ffi.check(lib.ts_add_rmsnorm(0, C.c_void_p(a.data_ptr()), None, C.c_void_p(norm.weight.data_ptr()), 1e-6, rows, d, code, None,
                                          C.c_void_p(out_plain.data_ptr()), st))
            torch.cuda.synchronize()
            assert torch.equal(out_sum, want_sum), (dtype, d)
            if dtype == torch.bfloat16:      # the module's roundings, followed step by step: at most one bf16 ulp where an fp32 sum differs
                assert (out.float() - want.float()).abs().max().item() <= 2 ** -6 * want.float().abs().max().item()
                assert (out != want).float().mean().item() < 1e-3 and (out_plain != plain).float().mean().item() < 1e-3
            else:
                assert torch.allclose(out, want, atol=1e-5, rtol=1e-5) and torch.allclose(out_plain, plain, atol=1e-5, rtol=1e-5)
            s64 = want_sum.double()
            ref = s64 * torch.rsqrt((s64 * s64).mean(-1, keepdim=True) + 1e-6) * norm.weight.double()
            assert torch.allclose(out.double(), ref, atol=tol, rtol=tol)
            _ffi.check(lib.ts_add_rmsnorm(0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(norm.weight.data_ptr()), 1e-6,
                                          rows, d, code, C.c_void_p(a.data_ptr()), C.c_void_p(out_plain.data_ptr()), st))   # residual in place
            torch.cuda.synchronize()
            assert torch.equal(a, want_sum) and torch.equal(out_plain, out)
    with pytest.raises(_ffi.TSearchError):
        _ffi.check(lib.ts_add_rmsnorm(0, C.c_void_p(a.data_ptr()), None, C.c_void_p(norm.weight.data_ptr()), 1e-6, 4, 10, 1, None,
                                      C.c_void_p(out.data_ptr()), None))


def test_qk_norm_rope_and_swiglu_kernels_match_the_modules():
    """ts_qk_norm_rope against Qwen3Attention's own q_norm / k_norm + apply_rotary_pos_emb (values untouched), ts_swiglu
    against Qwen3MLP's act_fn(gate) * up, on the same tensors."""
    import ctypes as C
    import torch
    from transformers.models.qwen3.modeling_qwen3 import Qwen3RMSNorm, apply_rotary_pos_emb
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(15)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    B, S, hq, hkv, hd = 5, 37, 16, 8, 128
    for dtype, code in ((torch.bfloat16, 1), (torch.float32, 0)):
        qkv = (torch.randn((B, S, (hq + 2 * hkv) * hd), generator=g) * 1.3).to(dtype).cuda()
        qn, kn = Qwen3RMSNorm(hd, eps=1e-6).to("cuda", dtype=dtype), Qwen3RMSNorm(hd, eps=1e-6).to("cuda", dtype=dtype)
        with torch.no_grad():
            qn.weight.copy_((1.0 + 0.3 * torch.randn(hd, generator=g)).to(dtype))
            kn.weight.copy_((1.0 + 0.3 * torch.randn(hd, generator=g)).to(dtype))
        inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
        emb = torch.cat((ang, ang), dim=-1)
        cos, sin = emb.cos().to(dtype).cuda(), emb.sin().to(dtype).cuda()           # [S x 128], cast as Qwen3RotaryEmbedding does
        with torch.no_grad():
            q = qn(qkv[..., :hq * hd].view(B, S, hq, hd)).transpose(1, 2)
            k = kn(qkv[..., hq * hd:(hq + hkv) * hd].view(B, S, hkv, hd)).transpose(1, 2)
            wq, wk = apply_rotary_pos_emb(q, k, cos[None], sin[None])
        got = qkv.clone()
        _ffi.check(lib.ts_qk_norm_rope(0, C.c_void_p(got.data_ptr()), C.c_void_p(qn.weight.data_ptr()), C.c_void_p(kn.weight.data_ptr()),
                                       C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), 1e-6, B * S, S, hq, hkv, hd, code, st))
        torch.cuda.synchronize()
        gq = got[..., :hq * hd].view(B, S, hq, hd).transpose(1, 2)
        gk = got[..., hq * hd:(hq + hkv) * hd].view(B, S, hkv, hd).transpose(1, 2)
        assert torch.equal(got[..., (hq + hkv) * hd:], qkv[..., (hq + hkv) * hd:])                 # the values
        tol = 2 ** -6 if dtype == torch.bfloat16 else 1e-5
        for have, want in ((gq, wq), (gk, wk)):
            assert (have.float() - want.float()).abs().max().item() <= tol * max(1.0, want.float().abs().max().item())
            if dtype == torch.bfloat16:
                assert (have != want).float().mean().item() < 2e-3                                   # the same roundings, step by step
        with pytest.raises(_ffi.TSearchError) as e:
            _ffi.check(lib.ts_qk_norm_rope(0, C.c_void_p(got.data_ptr()), C.c_void_p(qn.weight.data_ptr()), C.c_void_p(kn.weight.data_ptr()),
                                           C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), 1e-6, B * S, S, hq, hkv, 64, code, st))
        assert e.value.code == -5
        rows, inter = 1001, 3072
        gu = (torch.randn((rows, 2 * inter), generator=g) * 2.0).to(dtype).cuda()
        want = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
        out = torch.empty((rows, inter), dtype=dtype, device="cuda")
        _ffi.check(lib.ts_swiglu(0, C.c_void_p(gu.data_ptr()), rows, inter, code, C.c_void_p(out.data_ptr()), st))
        torch.cuda.synchronize()
        assert (out.float() - want.float()).abs().max().item() <= tol * max(1.0, want.float().abs().max().item())
        if dtype == torch.bfloat16:
            assert (out != want).float().mean().item() < 2e-3


def test_fused_qwen3_forward_matches_the_models_own():
    """FusedQwen3Forward (stacked projections + ts_add_rmsnorm + ts_qk_norm_rope + ts_swiglu) against Qwen3Model's own forward on
    the same random-init weights: hidden states to bf16 noise, sentence embeddings (last-token pooling) to cosine > 0.9995; padded
    batches (causal + padding mask) and unpadded ones (causal only); fp32 to 2e-4; TS_ENCODER_FUSED=0 keeps the model's own."""
    import torch
    from theoremsearch_amd.encoder import FusedQwen3Forward, SentenceEncoder
    name = "Qwen/Qwen3-Embedding-0.6B"
    enc = SentenceEncoder(name, num_layers=3, allow_random_init=True, dtype=torch.bfloat16)
    assert isinstance(enc._fused, FusedQwen3Forward) and enc.pooling == "lasttoken" and enc.embedding_dim == 1024
    texts = [f"lemma {i}: every finite group of order {i} " + "is solvable " * (i % 5) for i in range(40)]
    tok = {k: v.cuda() for k, v in enc._tokenize(texts).items()}
    assert not bool(tok["attention_mask"].all())                                  # ragged lengths: the padded path
    with torch.inference_mode():
        want = enc.model(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"]).last_hidden_state.float()
        got = enc.forward_hidden(tok["input_ids"], tok["attention_mask"]).float()
    real = tok["attention_mask"].bool()
    assert (want - got)[real].abs().max().item() < 0.25 and (want - got)[real].abs().mean().item() < 0.02
    same = tok["input_ids"][:, :6].contiguous()                                   # every row full: causal only, no mask tensor
    ones = torch.ones_like(same)
    with torch.inference_mode():
        w2 = enc.model(input_ids=same, attention_mask=ones).last_hidden_state.float()
        g2 = enc.forward_hidden(same, ones, no_padding=True).float()
    assert (w2 - g2).abs().max().item() < 0.25 and (w2 - g2).abs().mean().item() < 0.02
    fused = enc.encode(texts, normalize_embeddings=True, convert_to_numpy=True)
    os.environ["TS_ENCODER_FUSED"] = "0"
    try:
        plain_enc = SentenceEncoder(name, num_layers=3, allow_random_init=True, dtype=torch.bfloat16)
    finally:
        del os.environ["TS_ENCODER_FUSED"]
    assert plain_enc._fused is None
    plain = plain_enc.encode(texts, normalize_embeddings=True, convert_to_numpy=True)
    assert np.min(np.sum(fused * plain, axis=1)) > 0.9995
    f32 = SentenceEncoder(name, num_layers=2, allow_random_init=True, dtype=torch.float32)
    assert isinstance(f32._fused, FusedQwen3Forward)
    e32 = {k: v.cuda() for k, v in f32._tokenize(texts[:9]).items()}
    with torch.inference_mode():
        w32 = f32.model(input_ids=e32["input_ids"], attention_mask=e32["attention_mask"]).last_hidden_state
        g32 = f32.forward_hidden(e32["input_ids"], e32["attention_mask"])
    assert (w32 - g32)[e32["attention_mask"].bool()].abs().max().item() < 5e-4


# ---- round 4: the reference's second embedder, google/embeddinggemma-300m (Gemma3TextModel, bidirectional) -------------------------
def test_gemma3_kernels_match_the_modules():
    """ts_gemma_norm against Gemma3DecoderLayer's post-sublayer norm + residual add + the norm that follows (Gemma3RMSNorm:
    v * rsqrt(mean(v^2) + eps) * (1 + w) in fp32, rounded once), ts_gemma_qk_norm_rope against Gemma3Attention's q_norm / k_norm +
    apply_rotary_pos_emb over heads of 256 (values untouched), ts_geglu against Gemma3MLP's gelu_tanh(gate) * up - each on the
    same tensors as the torch modules, bf16 and fp32."""
    import ctypes as C
    import torch
    from transformers.models.gemma3.modeling_gemma3 import Gemma3RMSNorm, apply_rotary_pos_emb
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    g = torch.Generator(device="cpu").manual_seed(16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, code in ((torch.bfloat16, 1), (torch.float32, 0)):
        tol = 2 ** -6 if dtype == torch.bfloat16 else 2e-5
        for d in (768, 1152 if dtype == torch.bfloat16 else 1024, 640):
            rows = 333
            x = (torch.randn((rows, d), generator=g) * 3.0).to(dtype).cuda()
            y = (torch.randn((rows, d), generator=g) * 0.7).to(dtype).cuda()
            post, nxt = Gemma3RMSNorm(d, eps=1e-6).to("cuda", dtype=dtype), Gemma3RMSNorm(d, eps=1e-6).to("cuda", dtype=dtype)
            with torch.no_grad():
                post.weight.copy_((0.3 * torch.randn(d, generator=g)).to(dtype))
                nxt.weight.copy_((0.3 * torch.randn(d, generator=g)).to(dtype))
                want_sum = x + post(y)
                want = nxt(want_sum)
                plain = nxt(x)
            out_sum, out, out_plain = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            _ffi.check(lib.ts_gemma_norm(0, C.c_void_p(y.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(post.weight.data_ptr()),
                                         C.c_void_p(nxt.weight.data_ptr()), 1e-6, rows, d, code, C.c_void_p(out_sum.data_ptr()),
                                         C.c_void_p(out.data_ptr()), st))
            _ffi.check(lib.ts_gemma_norm(0, None, C.c_void_p(x.data_ptr()), None, C.c_void_p(nxt.weight.data_ptr()), 1e-6, rows, d, code,
                                         None, C.c_void_p(out_plain.data_ptr()), st))
            torch.cuda.synchronize()
            for have, ref in ((out_sum, want_sum), (out, want), (out_plain, plain)):
                assert (have.float() - ref.float()).abs().max().item() <= tol * max(1.0, ref.float().abs().max().item()), (dtype, d)
                if dtype == torch.bfloat16:
                    assert (have != ref).float().mean().item() < 2e-3, (dtype, d)          # the same roundings, step by step
        B, S, hq, hkv, hd = 5, 37, 3, 1, 256
        qkv = (torch.randn((B, S, (hq + 2 * hkv) * hd), generator=g) * 1.3).to(dtype).cuda()
        qn, kn = Gemma3RMSNorm(hd, eps=1e-6).to("cuda", dtype=dtype), Gemma3RMSNorm(hd, eps=1e-6).to("cuda", dtype=dtype)
        with torch.no_grad():
            qn.weight.copy_((0.3 * torch.randn(hd, generator=g)).to(dtype))
            kn.weight.copy_((0.3 * torch.randn(hd, generator=g)).to(dtype))
        inv = 1.0 / (1e4 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
        emb = torch.cat((ang, ang), dim=-1)
        cos, sin = emb.cos().to(dtype).cuda(), emb.sin().to(dtype).cuda()
        with torch.no_grad():
            q = qn(qkv[..., :hq * hd].view(B, S, hq, hd).transpose(1, 2))
            k = kn(qkv[..., hq * hd:(hq + hkv) * hd].view(B, S, hkv, hd).transpose(1, 2))
            wq, wk = apply_rotary_pos_emb(q, k, cos[None], sin[None])
        got = qkv.clone()
        _ffi.check(lib.ts_gemma_qk_norm_rope(0, C.c_void_p(got.data_ptr()), C.c_void_p(qn.weight.data_ptr()), C.c_void_p(kn.weight.data_ptr()),
                                             C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), 1e-6, B * S, S, hq, hkv, hd, code, st))
        torch.cuda.synchronize()
        gq = got[..., :hq * hd].view(B, S, hq, hd).transpose(1, 2)
        gk = got[..., hq * hd:(hq + hkv) * hd].view(B, S, hkv, hd).transpose(1, 2)
        assert torch.equal(got[..., (hq + hkv) * hd:], qkv[..., (hq + hkv) * hd:])                 # the values
        for have, want in ((gq, wq), (gk, wk)):
            assert (have.float() - want.float()).abs().max().item() <= tol * max(1.0, want.float().abs().max().item())
            if dtype == torch.bfloat16:
                assert (have != want).float().mean().item() < 2e-3
        with pytest.raises(_ffi.TSearchError) as e:
            _ffi.check(lib.ts_gemma_qk_norm_rope(0, C.c_void_p(got.data_ptr()), C.c_void_p(qn.weight.data_ptr()), C.c_void_p(kn.weight.data_ptr()),
                                                 C.c_void_p(cos.data_ptr()), C.c_void_p(sin.data_ptr()), 1e-6, B * S, S, hq, hkv, 128, code, st))
        assert e.value.code == -5
        rows, inter = 1001, 1152
        gu = (torch.randn((rows, 2 * inter), generator=g) * 2.0).to(dtype).cuda()
        want = torch.nn.functional.gelu(gu[:, :inter], approximate="tanh") * gu[:, inter:]
        out = torch.empty((rows, inter), dtype=dtype, device="cuda")
        _ffi.check(lib.ts_geglu(0, C.c_void_p(gu.data_ptr()), rows, inter, code, C.c_void_p(out.data_ptr()), st))
        torch.cuda.synchronize()
        assert (out.float() - want.float()).abs().max().item() <= tol * max(1.0, want.float().abs().max().item())
        if dtype == torch.bfloat16:
            assert (out != want).float().mean().item() < 5e-3


def test_fused_gemma3_forward_matches_the_models_own():
    """FusedGemma3Forward (stacked projections + ts_gemma_norm + ts_gemma_qk_norm_rope + ts_geglu) against Gemma3TextModel's own
    forward (bidirectional attention) on the same random-init weights: hidden states of the real tokens to bf16 noise, sentence
    embeddings (mean pooling, the two Dense modules, Normalize) to cosine > 0.9995; padded and unpadded batches; fp32 to 5e-4;
    TS_ENCODER_FUSED=0 keeps the model's own."""
    import torch
    from theoremsearch_amd.encoder import FusedGemma3Forward, SentenceEncoder
    name = "google/embeddinggemma-300m"
    enc = SentenceEncoder(name, num_layers=7, allow_random_init=True, dtype=torch.bfloat16)   # seven layers: sliding and full attention ones
    assert isinstance(enc._fused, FusedGemma3Forward) and enc.pooling == "mean" and enc.embedding_dim == 768
    assert len(set(enc.model.config.layer_types)) == 2
    texts = [f"lemma {i}: every finite group of order {i} " + "is solvable " * (i % 5) for i in range(40)]
    tok = {k: v.cuda() for k, v in enc._tokenize(texts).items()}
    assert not bool(tok["attention_mask"].all())
    with torch.inference_mode():
        want = enc.model(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"]).last_hidden_state.float()
        got = enc.forward_hidden(tok["input_ids"], tok["attention_mask"]).float()
    real = tok["attention_mask"].bool()
    scale = want[real].abs().max().item()
    assert (want - got)[real].abs().max().item() < 0.06 * scale and (want - got)[real].abs().mean().item() < 0.01 * scale
    same = tok["input_ids"][:, :6].contiguous()
    ones = torch.ones_like(same)
    with torch.inference_mode():
        w2 = enc.model(input_ids=same, attention_mask=ones).last_hidden_state.float()
        g2 = enc.forward_hidden(same, ones, no_padding=True).float()
    assert (w2 - g2).abs().max().item() < 0.06 * w2.abs().max().item()
    fused = enc.encode(texts, convert_to_numpy=True)
    assert np.allclose(np.sum(fused * fused, axis=1), 1.0, atol=1e-3)               # the pipeline's Normalize module
    os.environ["TS_ENCODER_FUSED"] = "0"
    try:
        plain_enc = SentenceEncoder(name, num_layers=7, allow_random_init=True, dtype=torch.bfloat16)
    finally:
        del os.environ["TS_ENCODER_FUSED"]
    assert plain_enc._fused is None
    plain = plain_enc.encode(texts, convert_to_numpy=True)
    assert np.min(np.sum(fused * plain, axis=1)) > 0.9995
    f32 = SentenceEncoder(name, num_layers=2, allow_random_init=True, dtype=torch.float32)
    assert isinstance(f32._fused, FusedGemma3Forward)
    e32 = {k: v.cuda() for k, v in f32._tokenize(texts[:9]).items()}
    with torch.inference_mode():
        w32 = f32.model(input_ids=e32["input_ids"], attention_mask=e32["attention_mask"]).last_hidden_state
        g32 = f32.forward_hidden(e32["input_ids"], e32["attention_mask"])
    assert (w32 - g32)[e32["attention_mask"].bool()].abs().max().item() < 5e-4 * max(1.0, w32.abs().max().item())


# ---- round 5: fp32-class GEMMs of the encoder on the bf16 matrix pipe ---------------------------------------------------------
def test_split_pieces_kernel_and_the_three_piece_linear():
    """ts_split_pieces: hi = bf16(x) exactly as torch rounds, lo = bf16(x - hi), laid out [hi | lo | hi] (activations) and
    [hi | hi | lo] (weights); hi + lo carries x to 2^-16 of its magnitude.  pieces_linear (ONE bf16 GEMM with fp32 accumulation
    over the three-fold depth) against fp64: within 2e-5 of |x||w| per entry - where the library's fp32 GEMM lands within
    ~1e-6 and a plain bf16 GEMM within ~4e-3 - with and without a bias, on a 3-D input."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    from theoremsearch_amd.fused_forward import pieces_linear, split_pieces
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn(300, 768, generator=g) * torch.logspace(-3, 2, 300)[:, None]).cuda()
    for pattern in (0, 1):
        p = split_pieces(x, pattern).float()
        hi = x.bfloat16().float()
        lo = (x - hi).bfloat16().float()
        assert torch.equal(p[:, :768], hi)
        assert torch.equal(p[:, 768:1536], lo if pattern == 0 else hi)
        assert torch.equal(p[:, 1536:], hi if pattern == 0 else lo)
        assert ((hi + lo - x).abs() <= x.abs() * 2.0 ** -16).all()
    with pytest.raises(_ffi.TSearchError):
        _ffi.check(_ffi.load().ts_split_pieces(0, C.c_void_p(x.data_ptr()), 300, 766, 0, C.c_void_p(x.data_ptr()), None))
    w = (torch.randn(512, 768, generator=g) * 0.05).cuda()
    b = torch.randn(512, generator=g).cuda()
    x3d = torch.randn(5, 60, 768, generator=g).cuda()
    ref = x3d.double() @ w.double().t()
    scale = x3d.double().norm(dim=-1, keepdim=True) * w.double().norm(dim=1)
    got = pieces_linear(x3d, split_pieces(w, 1))
    assert got.shape == (5, 60, 512) and got.dtype == torch.float32
    assert ((got.double() - ref).abs() / scale).max().item() < 2e-5
    got_b = pieces_linear(x3d, split_pieces(w, 1), b)
    assert ((got_b.double() - ref - b.double()).abs() / scale).max().item() < 2e-5
    plain = torch.nn.functional.linear(x3d.bfloat16(), w.bfloat16()).double()
    assert ((plain - ref).abs() / scale).max().item() > 20 * ((got.double() - ref).abs() / scale).max().item()


def test_float_attention_kernel_matches_fp64():
    """ts_attention_float (fp32 in / out, exact-fp32 matrix instructions) against softmax(Q K^T * scale + masks) V in fp64 for the
    three families' head shapes - BERT 12 x 64, Qwen3 16 / 8 x 128 causal grouped-query, Gemma3 3 / 1 x 256 with its own scale -
    at 5 ... 128 tokens, with and without a key mask; the optional pieces are exactly ts_split_pieces of the output; shapes the
    kernel does not serve are refused."""
    import ctypes as C
    import torch
    from theoremsearch_amd import _ffi
    from theoremsearch_amd.fused_forward import attention_float, split_pieces
    g = torch.Generator(device="cpu").manual_seed(77)
    for hq, hkv, hd, causal, scale in ((12, 12, 64, False, 64 ** -0.5), (16, 8, 128, True, 128 ** -0.5), (3, 1, 256, False, 0.0625)):
        for B, S in ((3, 5), (2, 16), (3, 33), (2, 128)) + (((2, 200), (1, 512)) if hd == 64 else ((2, 256),) if hd == 128 else ()):
            qkv = (torch.randn(B, S, (hq + 2 * hkv) * hd, generator=g) * 1.5).cuda()
            q = qkv[..., :hq * hd].view(B, S, hq, hd).double()
            k = qkv[..., hq * hd:(hq + hkv) * hd].view(B, S, hkv, hd).double().repeat_interleave(hq // hkv, dim=2)
            v = qkv[..., (hq + hkv) * hd:].view(B, S, hkv, hd).double().repeat_interleave(hq // hkv, dim=2)
            for use_mask in (False, True):
                mask = None
                if use_mask:
                    mask = torch.ones(B, S, dtype=torch.int64)
                    mask[0, S // 2 + 1:] = 0                                   # a padded sequence
                    mask = mask.cuda()
                s_ = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
                if causal:
                    s_ = s_.masked_fill(~torch.ones(S, S, dtype=torch.bool, device="cuda").tril_()[None, None], float("-inf"))
                if mask is not None:
                    s_ = s_.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
                want = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s_, dim=-1), v).reshape(B, S, hq * hd)
                got, pieces = attention_float(qkv, mask, B, S, hq, hkv, hd, causal, scale, want_pieces=True)
                torch.cuda.synchronize()
                real = torch.ones(B, S, dtype=torch.bool, device="cuda") if mask is None else mask.bool()
                err = (got.double() - want)[real].abs().max().item()
                assert err < 2e-5, (hq, hkv, hd, causal, B, S, use_mask, err)
                assert torch.equal(pieces.view(B, S, -1)[real], split_pieces(got.view(B * S, -1), 0).view(B, S, -1)[real])
    out = torch.empty(1, 4, 64, device="cuda")
    x = torch.randn(1, 600, 3 * 64, device="cuda")
    for S, hd in ((600, 64), (4, 32)):
        with pytest.raises(_ffi.TSearchError):
            _ffi.check(_ffi.load().ts_attention_float(0, C.c_void_p(x.data_ptr()), None, None, 1, S, 1, 1, hd, 0, 0.125,
                                                      C.c_void_p(out.data_ptr()), None, None))
    # only the pieces (the fp32-class forward reads nothing else): the same pieces, no fp32 context
    qkv = (torch.randn(2, 33, 3 * 12 * 64, generator=g) * 1.5).cuda()
    both = attention_float(qkv, None, 2, 33, 12, 12, 64, False, 0.125, want_pieces=True)
    only = attention_float(qkv, None, 2, 33, 12, 12, 64, False, 0.125, want_pieces=True, want_context=False)
    assert only[0] is None and torch.equal(only[1], both[1])
    # the stacked projection's bias added on the way in (the GEMM in front then runs without one): the same answer as on qkv + bias
    qkv = torch.randn(2, 40, 3 * 12 * 64, generator=g).cuda()
    bias = torch.randn(3 * 12 * 64, generator=g).cuda()
    want = attention_float(qkv + bias, None, 2, 40, 12, 12, 64, False, 0.125)[0]
    got = attention_float(qkv, None, 2, 40, 12, 12, 64, False, 0.125, bias=bias)[0]
    assert (got - want).abs().max().item() < 1e-5


def test_long_corpus_texts_take_the_float_attention_in_the_fused_fp32_forward():
    """Corpus texts are global context + statement (app_create_embeddings.py:48-70): hundreds of tokens, up to BERT's 512 positions.
    In fp32 - the reference's storage - the fused BERT forward serves them with ts_attention_float up to 512 tokens (Qwen3 heads:
    256): ragged batches of 150-400 tokens against the model's own forward, and against torch's attention (TS_ENCODER_ATTENTION=0)."""
    import torch
    from theoremsearch_amd.encoder import SentenceEncoder
    texts = [("Let $X_%d$ be a compact Hausdorff space and $f$ a continuous map. " % i) * (12 + 3 * (i % 7)) for i in range(9)]
    for name, layers in (("math-similarity/Bert-MLM_arXiv-MP-class_zbMath", 3), ("Qwen/Qwen3-Embedding-0.6B", 2)):
        enc = SentenceEncoder(name, num_layers=layers, allow_random_init=True, dtype=torch.float32)
        assert enc._fused is not None
        if "Qwen" in name:
            texts_ = [t[: len(t) * 2 // 5] for t in texts]                # up to ~230 tokens: the Qwen3 heads' limit is 256
        else:
            texts_ = texts
        tok = {k: v.cuda() for k, v in enc._tokenize(texts_).items()}
        S = tok["input_ids"].shape[1]
        assert 128 < S <= (256 if "Qwen" in name else 512), S
        with torch.inference_mode():
            want = enc.model(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"]).last_hidden_state
            got = enc.forward_hidden(tok["input_ids"], tok["attention_mask"])
            os.environ["TS_ENCODER_ATTENTION"] = "0"
            try:
                torch_way = enc.forward_hidden(tok["input_ids"], tok["attention_mask"])
            finally:
                del os.environ["TS_ENCODER_ATTENTION"]
        real = tok["attention_mask"].bool()
        scale = want[real].abs().max().item()
        assert (want - got)[real].abs().max().item() < 1e-4 * scale, (name, S)
        assert (torch_way - got)[real].abs().max().item() < 1e-4 * scale, (name, S)
