"""The oracle against the committed golden vectors (CPU only).

metrics_*.json were produced by the reference's own metric functions
(compare_embeddings.py:47-371); search_*.npz by the published algorithm of
sentence_transformers.util.cos_sim + torch.topk / np.argsort run with torch-CPU
(oracle/gen_golden.py).
"""
import numpy as np
import pytest

from conftest import (load_json, load_qrels, load_search_case, metric_cases, search_cases)
from oracle import oracle


@pytest.mark.parametrize("name", metric_cases())
def test_metrics_match_reference_outputs(name):
    doc = load_json(f"metrics_{name}.json")
    sim = np.array(doc["sim_matrix"], dtype=np.float32)
    qrels = load_qrels(doc)
    k = doc["k"]
    exp = doc["expected"]
    got = {
        "precision_at_k": oracle.precision_at_k(sim, qrels, k=k),
        "precision_at_1": oracle.precision_at_k(sim, qrels, k=1),
        "hit_at_k": oracle.hit_at_k(sim, qrels, k=k),
        "mrr_at_k": oracle.mrr_at_k(sim, qrels, k=k),
        "mrr_all": oracle.mrr_at_k(sim, qrels, k=None),
        "ndcg_at_k": oracle.ndcg_at_k(sim, qrels, k=k),
        "ndcg_linear": oracle.ndcg_at_k(sim, qrels, k=k, gain="linear"),
        "err_at_k": oracle.err_at_k(sim, qrels, k=k),
        "err_maxrel4": oracle.err_at_k(sim, qrels, k=k, max_rel=4.0),
        "q_measure_at_k": oracle.q_measure_at_k(sim, qrels, k=k),
    }
    for key, val in got.items():
        assert val == pytest.approx(exp[key], rel=0, abs=1e-15), key
    assert [int(i) for i in oracle.rank_concepts(sim)[0]] == exp["rank_row0"]
    if "generated_qrels" in doc:
        gen = oracle.generate_qrels([tuple(x) for x in doc["queries"]], [tuple(x) for x in doc["slogans"]])
        assert gen == load_qrels(doc, "generated_qrels")


def test_metrics_raise_without_grade_one():
    # compare_embeddings.py:111 - next() over an empty generator
    sim = np.random.default_rng(0).standard_normal((2, 5)).astype(np.float32)
    qrels = {0: {0: 0.5}, 1: {1: 0.5}}
    with pytest.raises(StopIteration):
        oracle.precision_at_k(sim, qrels, k=3)


@pytest.mark.parametrize("name", search_cases())
def test_search_matches_reference_formulation(name):
    case = load_search_case(name)
    q, c = oracle.golden_inputs(case["N"], case["B"], case["d"], case["seed"], case["metric"])
    k = case["k"]
    vals, idx = oracle.search(q, c, k, metric=case["metric"], dtype=case["dtype"])
    kk = min(k, case["N"])
    # fp64 truth of the golden run: ranks whose neighbours are > 1e-6 away are pinned
    ts = case["truth_scores"]
    ti = case["truth_idx"]
    gaps = ts[:, :-1] - ts[:, 1:]
    for b in range(case["B"]):
        for r in range(kk):
            lo = gaps[b, r - 1] if r > 0 else np.inf
            hi = gaps[b, r] if r < gaps.shape[1] else np.inf
            if lo > 1e-6 and hi > 1e-6:
                assert idx[b, r] == ti[b, r], (name, b, r)
                # and the reference primitives agree there too
                assert case["torch_topk_idx"][b, r] == ti[b, r]
                assert case["argsort_idx"][b, r] == ti[b, r]
    # scores: within 1e-5 of the reference's fp32 values at the same index
    assert np.allclose(np.sort(vals, axis=1), np.sort(case["torch_topk_scores"], axis=1), atol=1e-5, rtol=0)
    # the oracle's own fp64 check agrees with its answer
    qp, cp = oracle.prepared_inputs(q, c, case["metric"], case["dtype"])
    stats = oracle.check_topk_against_truth(oracle.scores_fp64(qp, cp), idx, vals, k)
    assert stats["recall"] == 1.0


def test_adversarial_primitives():
    adv = load_json("adversarial.json")
    t = adv["ties6"]
    s = np.array(t["scores"], dtype=np.float32)
    vals, idx = oracle.topk_canonical(s, t["k"])
    assert idx[0].tolist() == [1, 2, 4, 0]                 # score desc, index asc
    for key in ("argsort_neg", "argsort_rev", "torch_topk"):
        ref = t[key]
        assert sorted(s[ref].tolist()) == sorted(vals[0].tolist())     # same score multiset
        assert set(ref[:3]) == set(idx[0, :3].tolist())                # the three 0.9s
    z = np.zeros(adv["sparse_ones"]["n"], dtype=np.float32)
    z[adv["sparse_ones"]["ones_at"]] = 1.0
    _, idx = oracle.topk_canonical(z, 3)
    assert idx[0].tolist() == [7, 500, 900]
    assert set(adv["sparse_ones"]["torch_topk"]) == {7, 500, 900}
    n = np.array([np.nan if v is None else v for v in adv["nan"]["scores"]], dtype=np.float32)
    vals, idx = oracle.topk_canonical(n, 2)
    assert idx[0].tolist() == [2, 0]                        # NaN is never selected ...
    assert adv["nan"]["argsort_neg"] == [2, 0]              # ... like np.argsort(-x), unlike torch.topk
    vals, idx = oracle.topk_canonical(n, 4)
    assert idx[0].tolist() == [2, 0, 3, -1] and vals[0, 3] == -np.inf
    zr = adv["zero_row"]
    got = oracle.cos_sim(np.array(zr["a"], np.float32), np.array(zr["b"], np.float32))
    assert np.allclose(got, np.array(zr["cos_sim"]), atol=1e-6)
    assert got[0, 0] == 0.0 and not np.isnan(got).any()
    d = adv["duplicates"]
    S = oracle.cos_sim(np.array(d["query"], np.float32), np.array(d["corpus"], np.float32))
    assert np.allclose(S[0], np.array(d["cos_sim"]), atol=1e-6)
    _, idx = oracle.topk_canonical(S, 3)
    assert set(idx[0, :2].tolist()) == {3, 11} == set(d["torch_topk"][:2])
    assert idx[0, 2] == d["torch_topk"][2]


def test_k_larger_than_n_pads():
    s = np.array([[0.1, 0.3, 0.2]], dtype=np.float32)
    vals, idx = oracle.topk_canonical(s, 5)
    assert idx.tolist() == [[1, 2, 0, -1, -1]]
    assert np.isneginf(vals[0, 3:]).all()


def test_text_to_embed_matches_reference_strings():
    for case in load_json("text_to_embed.json"):
        assert oracle.global_context(case["paper"]) == case["global_context"]
        assert oracle.text_to_embed(case["paper"], case["theorem"]) == case["text_to_embed"]


def test_bf16_rounding_matches_torch():
    torch = pytest.importorskip("torch")
    x = np.random.default_rng(3).standard_normal(100000).astype(np.float32)
    x[:4] = [np.inf, -np.inf, 0.0, -0.0]
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(oracle.f32_to_bf16_bits(x), want)
    nan = np.array([np.nan], dtype=np.float32)
    assert np.isnan(oracle.round_to_bf16(nan)).all()


def test_pgvector_adapter_semantics():
    rng = np.random.default_rng(5)
    e = oracle.l2_normalize(rng.standard_normal((500, 32)).astype(np.float32))
    qv = oracle.l2_normalize(rng.standard_normal((1, 32)).astype(np.float32))[0]
    idx, sim = oracle.pgvector_search(qv, e, 7)
    ip = e @ qv
    assert idx.tolist() == np.argsort(-ip, kind="stable")[:7].tolist()
    assert np.allclose(sim, 1.0 + ip[idx], atol=1e-6)        # 1 - (<#>) = 1 + <e,q>  (streamlit_app.py:275)
    cit = [None, 0, 5, 100, 1, 20, 3]
    ridx, rsim, w = oracle.citation_weighted_rerank(idx, sim, cit, 0.05, 3)
    manual = sim + 0.05 * np.array([0, 0, np.log(5), np.log(100), 0, np.log(20), np.log(3)])
    assert ridx.tolist() == idx[np.argsort(-manual, kind="stable")[:3]].tolist()
    assert oracle.pool_size(3) == 50 and oracle.pool_size(20) == 200


def test_c_pgvector_scan_agrees_with_numpy_oracle():
    import os
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    rng = np.random.default_rng(6)
    e = oracle.l2_normalize(rng.standard_normal((20000, 768)).astype(np.float32))
    for seed in range(3):
        qv = oracle.l2_normalize(np.random.default_rng(seed).standard_normal((1, 768)).astype(np.float32))[0]
        rows_c, sim_c = oracle.pgvector_c_search(qv, e, 10)
        rows_np, sim_np = oracle.pgvector_search(qv, e, 10)
        truth = oracle.scores_fp64(qv[None, :], e)
        # sequential fp32 accumulation vs BLAS order: same ranking wherever the fp64 gaps are clear
        oracle.check_topk_against_truth(truth, rows_c[None, :], (sim_c - 1.0)[None, :].astype(np.float32), 10)
        oracle.check_topk_against_truth(truth, rows_np[None, :], (sim_np - 1.0)[None, :].astype(np.float32), 10)
    # ties and padding
    e2 = np.zeros((5, 4), np.float32)
    e2[[1, 3]] = [1, 0, 0, 0]
    rows, sim = oracle.pgvector_c_search(np.array([1, 0, 0, 0], np.float32), e2, 7)
    assert rows.tolist() == [1, 3, 0, 2, 4, -1, -1] and sim[:2].tolist() == [2.0, 2.0]


def test_bench_generator_agrees_with_the_oracle_restatement():
    """bench.py draws its corpus from /synthetic.py; the checkers regenerate rows through the oracle's own
    normalisation and bf16 rounding.  The two must agree bit for bit."""
    import synthetic
    for bf16 in (False, True):
        a = synthetic.synth_chunk(3, 20000, 768, bf16=bf16)
        b = oracle.synth_chunk(3, 20000, 768, bf16=bf16)
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert np.array_equal(synthetic.synth_queries(0, 7), oracle.synth_queries(0, 7))
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    assert np.array_equal(synthetic.bf16_bits(x), oracle.f32_to_bf16_bits(x))
    assert np.array_equal(synthetic.bf16_bits_to_f32(synthetic.bf16_bits(x)), oracle.round_to_bf16(x))


def test_chunked_truth_applies_the_same_protocol_as_the_matrix_form():
    """oracle.ChunkedTruth (full-size checks, bench.py's parity leg) against check_topk_against_truth on a corpus small
    enough for both: same statistics on a right answer (ties and a masked chunk included), same verdict on wrong ones."""
    rng = np.random.default_rng(3)
    q = rng.standard_normal((9, 48)).astype(np.float32)
    c = rng.standard_normal((6000, 48)).astype(np.float32)
    c[4100] = c[77]                                             # an exact tie that straddles two chunks
    truth = oracle.scores_fp64(q, c)
    scores, idx = oracle.search(q, c, 10)
    want = oracle.check_topk_against_truth(truth, idx, scores, 10)

    def run(idx_, scores_, allowed=None):
        ct = oracle.ChunkedTruth(q, idx_, 10)
        for r0 in range(0, 6000, 1700):
            ct.add(c[r0:r0 + 1700], r0, None if allowed is None else allowed[r0:r0 + 1700])
        return ct.check(scores_)

    assert run(idx, scores) == want and want["recall"] == 1.0
    # a swapped pair at a pinned rank, a duplicated index, a row outside the top-k, a score off by 1e-4: all rejected
    bad = idx.copy(); bad[2, [3, 4]] = bad[2, [4, 3]]
    dup = idx.copy(); dup[1, 9] = dup[1, 0]
    out = idx.copy(); out[5, 9] = int(np.argsort(truth[5])[0])
    off = scores.copy(); off[7, 2] += 1e-4
    for i_, s_ in ((bad, scores), (dup, scores), (out, scores), (idx, off)):
        with pytest.raises(AssertionError):
            run(i_, s_)
    # restricted to allowed rows: equals the matrix form over exactly those rows
    allowed = rng.random(6000) < 0.4
    keep = np.flatnonzero(allowed)
    ms, mi_local = oracle.search(q, c[keep], 10)
    mi = keep[mi_local]
    assert run(mi, ms, allowed) == oracle.check_topk_against_truth(truth[:, keep], mi_local, ms, 10)
    # sharded accumulation: two accumulators merged = one
    a, b = oracle.ChunkedTruth(q, idx, 10), oracle.ChunkedTruth(q, idx, 10)
    a.add(c[:3000], 0)
    b.add(c[3000:], 3000)
    a.merge(b.best_s, b.best_i, b.got_s, b.n)
    assert a.check(scores) == want


def test_packed_result_blocks_round_trip():
    from theoremsearch_amd.distributed import pack_results, packed_bytes, packed_idx_off, unpack_results
    rng = np.random.default_rng(1)
    for nq, k in ((1, 1), (3, 5), (256, 10), (7, 200)):
        parts = [(rng.standard_normal((nq, k)).astype(np.float32), rng.integers(-1, 10**10, (nq, k))) for _ in range(3)]
        blob = np.concatenate([pack_results(s, i) for s, i in parts])
        assert blob.size == 3 * packed_bytes(nq, k) and packed_idx_off(nq, k) % 8 == 0
        s, i = unpack_results(blob, 3, nq, k)
        for p in range(3):
            assert np.array_equal(s[p], parts[p][0]) and np.array_equal(i[p], parts[p][1])
