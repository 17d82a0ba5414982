"""ts_search_cpu - the library's ONE host-only entry (BASELINE.json configs[0], the reference's CPU-runnable
plumbing case: util.cos_sim + argsort over ~1k theorems, compare_embeddings.py:24-31,55-92) - against the same committed
golden vectors, ragged shapes and adversarial fixtures the HIP path is held to (tests/test_search_gpu.py), on the CPU box.
The way in is explicit: TheoremIndex(..., device=-1).  Nothing falls back to it: every other device number still raises
without a GPU (checked at the end)."""
import numpy as np
import pytest

from conftest import gpu_available, load_json, load_search_case, search_cases
from oracle import oracle

GAP = 1e-6
SCORE_TOL = 1e-5


@pytest.fixture(scope="module")
def ts():
    import theoremsearch_amd as ts
    return ts


def check(q, c, metric, dtype, k, scores, idx):
    qp, cp = oracle.prepared_inputs(q, c, metric, dtype)
    stats = oracle.check_topk_against_truth(oracle.scores_fp64(qp, cp), idx, scores, k, gap=GAP, score_tol=SCORE_TOL)
    assert stats["recall"] == 1.0
    return stats


@pytest.mark.parametrize("name", [n for n in search_cases() if load_search_case(n)["N"] * load_search_case(n)["B"] <= 4096 * 256])
def test_golden_cases_on_the_host(ts, name):
    case = load_search_case(name)
    q, c = oracle.golden_inputs(case["N"], case["B"], case["d"], case["seed"], case["metric"])
    k = case["k"]
    with ts.TheoremIndex.from_embeddings(c, dtype=case["dtype"], metric=case["metric"], device=-1) as ix:
        scores, idx = ix.search(q, k)
    stats = check(q, c, case["metric"], case["dtype"], k, scores, idx)
    ts_, ti = case["truth_scores"], case["truth_idx"]             # the golden run (torch formulation of the reference)
    gaps = ts_[:, :-1] - ts_[:, 1:]
    for b in range(case["B"]):
        for r in range(min(k, case["N"])):
            lo = gaps[b, r - 1] if r > 0 else np.inf
            hi = gaps[b, r] if r < gaps.shape[1] else np.inf
            if lo > GAP and hi > GAP:
                assert idx[b, r] == case["torch_topk_idx"][b, r] == ti[b, r], (name, b, r)
    assert stats["pinned"] > 0.9 * stats["positions"]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 1, 1), (5, 8, 3, 10), (31, 40, 2, 5), (257, 768, 5, 256), (1000, 1024, 9, 7),
                                      (4097, 100, 4, 64), (3001, 384, 6, 10)])
def test_ragged_shapes_on_the_host(ts, dtype, n, d, nq, k):
    rng = np.random.default_rng(n * 7 + d)
    c = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos", device=-1) as ix:
        scores, idx = ix.search(q, k)
    assert scores.shape == (nq, k) and idx.shape == (nq, k)
    check(q, c, "cos", dtype, k, scores, idx)
    if k > n:
        assert (idx[:, n:] == -1).all() and np.isneginf(scores[:, n:]).all()


def test_adversarial_fixtures_on_the_host(ts):
    adv = load_json("adversarial.json")
    dup = adv["duplicates"]
    c, qv = np.array(dup["corpus"], np.float32), np.array(dup["query"], np.float32)
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos", device=-1) as ix:
        scores, idx = ix.search(qv, 3)
    assert set(idx[0, :2].tolist()) == {3, 11} == set(dup["torch_topk"][:2]) and idx[0, 2] == dup["torch_topk"][2]
    assert np.allclose(scores[0], np.sort(np.array(dup["cos_sim"], np.float32))[::-1][:3], atol=SCORE_TOL)
    zr = adv["zero_row"]
    with ts.TheoremIndex.from_embeddings(np.array(zr["b"], np.float32), dtype="f32", metric="cos", device=-1) as ix:
        scores, idx = ix.search(np.array(zr["a"], np.float32), 3)
    assert idx[0].tolist() == [1, 0, 2]
    # exact ties resolve to the lowest row; NaN rows never rank
    rng = np.random.default_rng(8)
    base = rng.standard_normal((50, 768), dtype=np.float32)
    c3 = np.concatenate([base, base, base], axis=0)
    q = base[:4] + 0.01 * rng.standard_normal((4, 768), dtype=np.float32)
    for dtype in ("f32", "bf16"):
        with ts.TheoremIndex.from_embeddings(c3, dtype=dtype, metric="cos", device=-1) as ix:
            scores, idx = ix.search(q, 6)
        assert idx[:, :3].tolist() == [[b, b + 50, b + 100] for b in range(4)]
        check(q, c3, "cos", dtype, 6, scores, idx)
    cn = rng.standard_normal((300, 768), dtype=np.float32)
    cn[7, 5] = np.nan
    cn[200, :] = np.nan
    qn = rng.standard_normal((2, 768), dtype=np.float32)
    with ts.TheoremIndex.from_embeddings(cn, dtype="f32", metric="ip", device=-1) as ix:
        scores, idx = ix.search(qn, 256)
    assert 7 not in idx and 200 not in idx and not np.isnan(scores).any()
    assert set(idx[0].tolist()) == set(oracle.search(qn, cn, 256, "ip", "f32")[1][0].tolist())


def test_nothing_falls_back_to_the_host_entry(ts):
    """device=-1 is the only way in: the host index has upload / search / close and nothing else, and without a GPU every
    other device number still raises (the library has no device entry that computes on the host)."""
    from theoremsearch_amd import _ffi
    c = np.random.default_rng(1).standard_normal((100, 16), dtype=np.float32)
    ix = ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos", device=-1, row_offset=1000)
    s, i = ix.search(c[:2], 1)
    assert i[:, 0].tolist() == [1000, 1001]                        # global ids, as on the device
    with pytest.raises(ValueError):
        ix.search(c[:2], 1, algo="mfma")
    with pytest.raises(ValueError):
        ix.search(c[:2], 1, mask=np.ones(100, bool))
    ix.close()
    if not gpu_available():
        with pytest.raises(_ffi.TSearchError):
            ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos", device=0)
