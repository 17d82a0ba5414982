"""Parity of the HIP search path with the CPU oracle and the committed golden vectors.
Everything here goes through the C ABI (theoremsearch_amd._ffi -> libtsearch.so)."""
import json

import numpy as np
import pytest

from conftest import load_json, load_search_case, search_cases
from oracle import oracle

pytestmark = pytest.mark.gpu

GAP = 1e-6        # fp64 gap below which two ranks count as tied
SCORE_TOL = 1e-5  # north_star: cosine scores within 1e-5 fp32


@pytest.fixture(scope="module")
def ts():
    import theoremsearch_amd as ts
    from theoremsearch_amd import _ffi
    assert _ffi.device_count() > 0, "GPU tests need a HIP device"
    return ts


def algos_for(dtype, d):
    if dtype == "bf16" and d in (384, 512, 768, 1024):
        return ["scan", "mfma"]
    return ["scan", "mfma"] if (dtype == "f32" and d in (384, 512, 768, 1024)) else ["scan"]     # fp32 MFMA (exact fp32)


def check(q, c, metric, dtype, k, scores, idx):
    qp, cp = oracle.prepared_inputs(q, c, metric, dtype)
    truth = oracle.scores_fp64(qp, cp)
    stats = oracle.check_topk_against_truth(truth, idx, scores, k, gap=GAP, score_tol=SCORE_TOL)
    assert stats["recall"] == 1.0
    return stats


@pytest.mark.parametrize("name", search_cases())
def test_golden_cases(ts, name):
    case = load_search_case(name)
    q, c = oracle.golden_inputs(case["N"], case["B"], case["d"], case["seed"], case["metric"])
    k = case["k"]
    with ts.TheoremIndex.from_embeddings(c, dtype=case["dtype"], metric=case["metric"]) as ix:
        for algo in algos_for(case["dtype"], case["d"]):
            scores, idx = ix.search(q, k, algo=algo)
            stats = check(q, c, case["metric"], case["dtype"], k, scores, idx)
            # the golden run (torch formulation of the reference) at pinned ranks
            ts_, ti = case["truth_scores"], case["truth_idx"]
            gaps = ts_[:, :-1] - ts_[:, 1:]
            kk = min(k, case["N"])
            for b in range(case["B"]):
                for r in range(kk):
                    lo = gaps[b, r - 1] if r > 0 else np.inf
                    hi = gaps[b, r] if r < gaps.shape[1] else np.inf
                    if lo > GAP and hi > GAP:
                        assert idx[b, r] == case["torch_topk_idx"][b, r] == ti[b, r], (name, algo, b, r)
            assert stats["pinned"] > 0.9 * stats["positions"]


def test_index_rows_match_oracle_preparation(ts):
    q, c = oracle.inputs(3000, 1, 768, 21, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        got = ix.download()
    want = oracle.l2_normalize(c)
    assert got.shape == want.shape
    assert np.mean(got == want) > 0.99999 and np.allclose(got, want, rtol=0, atol=1e-7)
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        gotb = ix.download()
    wantb = oracle.f32_to_bf16_bits(want)
    assert np.mean(gotb == wantb) > 0.9999
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:     # plain RNE of the given values
        assert np.array_equal(ix.download(), oracle.f32_to_bf16_bits(c))
    with ts.TheoremIndex.from_embeddings(oracle.f32_to_bf16_bits(c), dtype="bf16", metric="ip") as ix:  # stored as given
        assert np.array_equal(ix.download(), oracle.f32_to_bf16_bits(c))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n,d,nq,k", [(1, 8, 1, 1), (5, 8, 3, 10), (31, 40, 2, 5), (257, 768, 5, 256),
                                      (1000, 1024, 9, 7), (4097, 100, 4, 64), (600, 768, 300, 3),
                                      (3001, 384, 6, 10), (2003, 384, 5, 200), (2500, 512, 7, 70), (900, 500, 3, 5)])
def test_ragged_shapes(ts, dtype, n, d, nq, k):
    rng = np.random.default_rng(n * 7 + d)
    c = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos") as ix:
        for algo in algos_for(dtype, d):
            scores, idx = ix.search(q, k, algo=algo)
            assert scores.shape == (nq, k) and idx.shape == (nq, k)
            check(q, c, "cos", dtype, k, scores, idx)
            if k > n:
                assert (idx[:, n:] == -1).all() and np.isneginf(scores[:, n:]).all()


def test_adversarial_fixtures(ts):
    adv = load_json("adversarial.json")
    dup = adv["duplicates"]
    c = np.array(dup["corpus"], np.float32)
    qv = np.array(dup["query"], np.float32)
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        scores, idx = ix.search(qv, 3)
        assert set(idx[0, :2].tolist()) == {3, 11} == set(dup["torch_topk"][:2])
        assert idx[0, 2] == dup["torch_topk"][2]
        assert np.allclose(scores[0], np.sort(np.array(dup["cos_sim"], np.float32))[::-1][:3], atol=SCORE_TOL)
        full = ix.scores(qv)
        assert np.allclose(full[0], np.array(dup["cos_sim"]), atol=SCORE_TOL)
    zr = adv["zero_row"]
    with ts.TheoremIndex.from_embeddings(np.array(zr["b"], np.float32), dtype="f32", metric="cos") as ix:
        full = ix.scores(np.array(zr["a"], np.float32))
        assert np.allclose(full, np.array(zr["cos_sim"]), atol=1e-6) and full[0, 0] == 0.0
        scores, idx = ix.search(np.array(zr["a"], np.float32), 3)
        assert idx[0].tolist() == [1, 0, 2]


def test_exact_ties_resolve_to_lowest_index(ts):
    rng = np.random.default_rng(8)
    base = rng.standard_normal((50, 768), dtype=np.float32)
    c = np.concatenate([base, base, base], axis=0)            # every row three times
    q = base[:4] + 0.01 * rng.standard_normal((4, 768), dtype=np.float32)
    for dtype in ("f32", "bf16"):
        with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos") as ix:
            for algo in algos_for(dtype, 768):
                scores, idx = ix.search(q, 6, algo=algo)
                want_s, want_i = oracle.search(q, c, 6, "cos", dtype)
                # duplicates have bit-identical scores on the device: canonical order is exact
                assert idx[:, :3].tolist() == [[b, b + 50, b + 100] for b in range(4)], (dtype, algo)
                assert scores[:, 0].tolist() == scores[:, 1].tolist() == scores[:, 2].tolist()
                check(q, c, "cos", dtype, 6, scores, idx)


def test_nan_rows_are_never_returned(ts):
    rng = np.random.default_rng(9)
    c = rng.standard_normal((300, 768), dtype=np.float32)
    c[7, 5] = np.nan
    c[200, :] = np.nan
    q = rng.standard_normal((2, 768), dtype=np.float32)
    for dtype in ("f32", "bf16"):
        with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
            for algo in algos_for(dtype, 768):
                scores, idx = ix.search(q, 300, algo=algo) if False else ix.search(q, 256, algo=algo)
                assert 7 not in idx and 200 not in idx
                assert not np.isnan(scores).any()
                want_s, want_i = oracle.search(q, c, 256, "ip", dtype)
                assert set(idx[0].tolist()) == set(want_i[0].tolist())


def test_mfma_and_scan_agree_and_levels_run(ts):
    q, c = oracle.inputs(150_000, 40, 768, 31, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        s1, i1, st = ix.search(q, 10, algo="mfma", return_stats=True)
        s2, i2 = ix.search(q, 10, algo="scan")
        assert st["algo"] == 2 and st["levels"] >= 2 and st["fallback_queries"] == 0
        assert 0 < st["candidates"] < 40 * 8192
        check(q, c, "ip", "bf16", 10, s1, i1)
        check(q, c, "ip", "bf16", 10, s2, i2)
        assert np.mean(i1 == i2) > 0.999
        s3, i3 = ix.search(q, 200, algo="mfma")
        check(q, c, "ip", "bf16", 200, s3, i3)
        s4, i4, st4 = ix.search(q[:1], 5, return_stats=True)      # a single query goes through the scan ...
        check(q[:1], c, "ip", "bf16", 5, s4, i4)
        s5, i5, st5 = ix.search(q[:9], 5, return_stats=True)      # ... a batch through the MFMA path
        check(q[:9], c, "ip", "bf16", 5, s5, i5)
        assert (st4["algo"], st5["algo"]) == (1, 2)


@pytest.mark.parametrize("knobs", [{"TS_MFMA_AHEAD": 2}, {"TS_MFMA_RUN": 4, "TS_MFMA_STAT": 0}, {"TS_MFMA_GRID": 64},
                                   {"TS_MFMA_GRID": 200, "TS_MFMA_AHEAD": 1}, {"TS_MFMA_SHAPE": 32, "TS_MFMA_AHEAD": 1}])
@pytest.mark.parametrize("n,d,nq", [(310_000, 768, 256), (120_000, 1024, 150)])
def test_general_units_of_the_tile_loop_give_the_same_answer(ts, knobs, n, d, nq):
    """The tile loop of the 16x16 kernel has a branch-free steady part (default ring depth, single-tile runs) and general
    units for the rest: a shallower DMA ring, sampled runs of tiles and other grids (tiles per workgroup) drive the
    general units and the hand-over between the two at other places; the answers may not change."""
    q, c = oracle.inputs(n, nq, d, 4242 + nq, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        s0, i0, st0 = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st0["fallback_queries"] == 0, st0
        check(q, c, "ip", "bf16", 10, s0, i0)
        for k_, v_ in knobs.items():
            ix.set_option(k_, v_)
        s1, i1, st1 = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st1["fallback_queries"] == 0, st1
        assert np.array_equal(i1, i0) and np.array_equal(s1, s0)


def test_mfma_d1024_and_query_blocks(ts):
    # Qwen-sized rows (vector(1024), rds_schema.sql:50-56): 128 queries per launch, so 300 queries = 3 blocks
    q, c = oracle.inputs(50_000, 300, 1024, 33, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        s1, i1, st = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st["algo"] == 2 and st["fallback_queries"] == 0
        check(q, c, "ip", "bf16", 10, s1, i1)
        s2, i2 = ix.search(q[:130], 10, algo="scan")
        assert np.mean(i1[:130] == i2) > 0.999
    # d = 768: 128 < nq <= 256 runs two query groups per wave, nq <= 128 one
    q, c = oracle.inputs(40_000, 200, 768, 34, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        for nq in (128, 129, 200):
            s, i = ix.search(q[:nq], 7, algo="mfma")
            check(q[:nq], c, "cos", "bf16", 7, s, i)


@pytest.mark.parametrize("nq", [193, 256])
def test_paired_full_pass_at_d1024_answers_like_two_launches(ts, nq):
    """bf16 x 1024 (the production table, rds_schema.sql:50-56) holds 192 queries per workgroup; 193 .. 256 queries run as ONE
    launch of workgroup pairs (each half of a pair multiplies 128 of the queries against the same tiles) instead of two
    launches of 128.  Two forms: TS_MFMA_PAIR=1, two query blocks per wave over the whole row - the sum of a score in the
    order of the unpaired launches, so the same answers bit for bit; and the default, the k-split (2 x 2 waves: query column x
    half of every unit's k-steps, the two partial sums of a score meet through LDS) - the same products in another order of
    the fp32 sum: against the oracle, scores within an ulp or two of the unpaired ones, and its own answers bit for bit
    with a row mask, on other grids, as the tile shares move over repeated searches, and for device queries read in place."""
    import torch
    n = 330_000
    q, c = oracle.inputs(n, nq, 1024, 7100 + nq, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        ix.set_option("TS_MFMA_PAIR", 0)
        s0, i0, st0 = ix.search(q, 10, algo="mfma", return_stats=True)
        check(q, c, "ip", "bf16", 10, s0, i0)
        ix.set_option("TS_MFMA_PAIR", 1)
        s1, i1, st1 = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st1["fallback_queries"] == 0 and st1["algo"] == 2, st1
        assert np.array_equal(i1, i0) and np.array_equal(s1, s0)
        ix.set_option("TS_MFMA_PAIR", None)                    # the default: the k-split form
        sk, ik, stk = ix.search(q, 10, algo="mfma", return_stats=True)
        assert stk["fallback_queries"] == 0 and stk["algo"] == 2, stk
        check(q, c, "ip", "bf16", 10, sk, ik)
        assert np.abs(sk - s0).max() < 3e-7 and np.mean(ik == i0) > 0.995
        i0, s0 = ik, sk
        for rep in range(4):                                   # the feedback partition moves the pairs' tile shares
            s1, i1, st1 = ix.search(q, 10, algo="mfma", return_stats=True)
            assert st1["fallback_queries"] == 0 and st1["algo"] == 2, st1
            assert np.array_equal(i1, i0) and np.array_equal(s1, s0), rep
        for grid in (64, 208):                                 # 208 = 13 groups of 16; 64: 32 pairs
            ix.set_option("TS_MFMA_GRID", grid)
            s2, i2 = ix.search(q, 10, algo="mfma")
            assert np.array_equal(i2, i0) and np.array_equal(s2, s0), grid
        ix.set_option("TS_MFMA_GRID", 200)                     # not a multiple of 16: no pairs, two launches
        s2, i2 = ix.search(q, 10, algo="mfma")
        assert np.abs(s2 - s0).max() < 3e-7 and np.mean(i2 == i0) > 0.995
        ix.set_option("TS_MFMA_GRID", None)
        mask = np.random.default_rng(3).random(n) < 0.4
        sm, im = ix.search(q, 10, mask=mask, algo="mfma")
        keep = np.flatnonzero(mask)
        qp, cp = oracle.prepared_inputs(q, c, "ip", "bf16")
        stats = oracle.check_topk_against_truth(oracle.scores_fp64(qp, cp[keep]), np.searchsorted(keep, im), sm, 10)
        assert stats["recall"] == 1.0 and mask[im].all()
        if nq == 256:                                          # the storage form, a whole launch's worth: read in place
            qd = torch.from_numpy(q).cuda().to(torch.bfloat16)
            out_s = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
            out_i = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
            st = torch.cuda.Stream()
            ix.search_device(qd.data_ptr(), "bf16", nq, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream, algo="mfma")
            st.synchronize()
            want_s, want_i = ix.search(oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(q)), 10, algo="mfma")
            assert np.array_equal(out_i.cpu().numpy(), want_i) and np.array_equal(out_s.cpu().numpy(), want_s)


@pytest.mark.parametrize("n,grid", [(700, 16), (3_000, 16), (3_000, 32), (9_001, 256), (40_000, 48), (131_072, 256), (131_072, 1024)])
def test_k_split_pairs_on_short_and_odd_tile_ranges(ts, n, grid):
    """The k-split form tests a tile one tile late (the partner's half arrives during the next tile) and keeps its sums in two
    register halves by tile parity: tile ranges of 0, 1, 2, 3 ... tiles per pair, odd and even, a ragged last tile, ranges that
    end inside the steady part of the loop and ranges that never reach it, more workgroups than CUs (the pairs' position words are one
    per workgroup of the grid) - against the oracle and the other form."""
    q, c = oracle.inputs(n, 256, 1024, 9100 + n + grid, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        ix.set_option("TS_MFMA_GRID", grid)
        sk, ik, st = ix.search(q, 10, algo="mfma", return_stats=True)
        check(q, c, "ip", "bf16", 10, sk, ik)
        ix.set_option("TS_MFMA_PAIR", 1)
        s1, i1 = ix.search(q, 10, algo="mfma")
        check(q, c, "ip", "bf16", 10, s1, i1)
        assert np.abs(sk - s1).max() < 3e-7 and np.mean(ik == i1) > 0.995
        ix.set_option("TS_MFMA_PAIR", None)
        ix.set_option("TS_MFMA_PAIR_LAG", 0)                   # the pairs unpaced: the same sums, bit for bit
        s0, i0 = ix.search(q, 10, algo="mfma")
        assert np.array_equal(i0, ik) and np.array_equal(s0, sk)
        ix.set_option("TS_MFMA_PAIR_LAG", None)
        for k in (1, 100):
            s, i = ix.search(q[:200], k, algo="mfma")
            check(q[:200], c, "ip", "bf16", k, s, i)


@pytest.mark.parametrize("kind", ["outliers", "light_tail", "shifted"])
def test_estimated_threshold_is_verified_not_trusted(ts, kind):
    # The full pass's threshold is extrapolated from one sample (Gaussian tail).  Distributions that break the
    # estimate must still give the exact answer: too many candidates or too few both end in the exact re-run.
    rng = np.random.default_rng(44)
    n, d = 120_000, 768
    c = rng.standard_normal((n, d), dtype=np.float32) * np.float32(1 / np.sqrt(d))
    q = rng.standard_normal((24, d), dtype=np.float32) * np.float32(1 / np.sqrt(d))
    if kind == "outliers":            # 1 % of the rows 20x longer: sample variance inflated, heavy tail
        c[rng.choice(n, n // 100, replace=False)] *= np.float32(20.0)
    elif kind == "light_tail":        # scores bounded: rows are signed unit vectors of 4 coordinates
        c = np.zeros((n, d), np.float32)
        cols = rng.integers(0, d, size=(n, 4))
        c[np.arange(n)[:, None], cols] = rng.choice([-0.5, 0.5], size=(n, 4)).astype(np.float32)
    else:                             # every score shifted far from zero by a common component
        c += q.mean(axis=0) * np.float32(30.0)
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        scores, idx, st = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st["algo"] == 2
        check(q, c, "ip", "bf16", 10, scores, idx)


@pytest.mark.parametrize("d", [384, 512])
def test_mfma_narrow_widths(ts, d):
    """d = 384 / 512 (MiniLM-class models) on the MFMA path: one and two query groups per wave, partial last tile,
    more queries than one launch holds, large k."""
    q, c = oracle.inputs(70_013, 300, d, 90 + d, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        for nq, k in ((5, 10), (128, 10), (129, 3), (256, 50), (300, 256)):
            s, i, st = ix.search(q[:nq], k, algo="mfma", return_stats=True)
            assert st["algo"] == 2
            check(q[:nq], c, "cos", "bf16", k, s, i)
        sa, ia = ix.search(q[:40], 10)              # auto picks the MFMA path for a batch
        ss, is_ = ix.search(q[:40], 10, algo="scan")
        assert np.mean(ia == is_) > 0.999
        # round 3: these widths run the 16x16x32 kernel (d = 384 as one unit of 12 k-steps per tile, d = 512 as two of 8);
        # the 32x32x16 kernel is the A/B partner - bit-identical answers (bf16 products are exact, fp32 sums of the same set)
        for nq, k in ((64, 10), (200, 10), (256, 50)):
            s16, i16 = ix.search(q[:nq], k, algo="mfma")
            ix.set_option("TS_MFMA_SHAPE", 32)
            s32, i32 = ix.search(q[:nq], k, algo="mfma")
            ix.set_option("TS_MFMA_SHAPE", None)
            assert np.array_equal(i16, i32) and np.allclose(s16, s32, atol=2e-7), (d, nq, k)
        ix.set_option("TS_MFMA_STAT", 0)            # the guaranteed chain at these widths runs the 32x32 kernel
        sg, ig = ix.search(q[:100], 10, algo="mfma")
        ix.set_option("TS_MFMA_STAT", None)
        check(q[:100], c, "cos", "bf16", 10, sg, ig)


@pytest.mark.parametrize("d", [384, 512])
def test_fp32_batches_at_narrow_widths_run_on_the_fp32_mfma_path(ts, d):
    """fp32 x 384 / 512 (MiniLM-class models in the reference's fp32): round 2 served a batch as ceil(Q / 4) scan passes;
    now the F32 mode of the 16x16 kernel (v_mfma_f32_16x16x4_f32, exact fp32)."""
    q, c = oracle.inputs(60_007, 150, d, 40 + d, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        for nq, k in ((5, 10), (64, 10), (65, 3), (150, 100)):
            s, i, st = ix.search(q[:nq], k, return_stats=True)
            assert st["algo"] == 2 and st["fallback_queries"] == 0, (d, nq, st)
            check(q[:nq], c, "cos", "f32", k, s, i)
        ix.set_option("TS_MFMA_STAT", 0)            # no sparse level of this kernel at these widths: the batch takes the scan
        s, i, st = ix.search(q[:20], 10, return_stats=True)
        ix.set_option("TS_MFMA_STAT", None)
        assert st["algo"] == 1
        check(q[:20], c, "cos", "f32", 10, s, i)


def test_clustered_corpus_is_served_without_reruns(ts):
    """5 % of the rows form a cluster that every query is close to (related theorems).  The Gaussian estimate lands
    inside the cluster, and the only other bound - the k-th best of the sample - admits k * N / sample rows, which
    for a large corpus is more than the candidate buffer holds: every query would be re-run by the exact scan.  The
    exponential tail fit of the sample's order statistics holds the candidate count near its target instead.
    Scaled down to a test by a small sample (TS_MFMA_FIRST_ROWS) and k = 50: with the fit no re-runs, without it
    all queries overflow; the answers are exact and identical either way."""
    rng = np.random.default_rng(71)
    n, d, nq, k = 300_000, 768, 16, 50
    c = rng.standard_normal((n, d), dtype=np.float32) * np.float32(1 / np.sqrt(d))
    u = rng.standard_normal(d).astype(np.float32)
    u /= np.linalg.norm(u)
    members = rng.choice(n, n // 20, replace=False)
    c[members] += (0.5 + 0.1 * rng.standard_normal(members.size)).astype(np.float32)[:, None] * u
    q = u + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.2 / np.sqrt(d))
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        ix.set_option("TS_MFMA_FIRST_ROWS", 2048)
        scores, idx, st = ix.search(q, k, algo="mfma", return_stats=True)
        check(q, c, "ip", "bf16", k, scores, idx)
        assert st["fallback_queries"] == 0
        assert k * nq <= st["candidates"] < nq * 4096
        ix.set_option("TS_MFMA_TAIL_FIT", 0)
        s0, i0, st0 = ix.search(q, k, algo="mfma", return_stats=True)
        ix.set_option("TS_MFMA_TAIL_FIT", None)
        check(q, c, "ip", "bf16", k, s0, i0)                # exact either way (through the re-run)
        assert st0["fallback_queries"] > 0 and st0["candidates"] > 3 * st["candidates"], (st, st0)


def test_candidate_overflow_falls_back_exactly(ts):
    # 20,000 copies of one row that matches query 0: far more ties above any threshold than the
    # candidate buffer holds -> that query must be re-run by the exact scan
    rng = np.random.default_rng(10)
    c = rng.standard_normal((60_000, 768), dtype=np.float32) * np.float32(0.05)
    hot = rng.standard_normal(768).astype(np.float32)
    rows = rng.choice(60_000, size=20_000, replace=False)
    c[rows] = hot
    q = np.stack([hot, rng.standard_normal(768).astype(np.float32)])
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        scores, idx, st = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st["fallback_queries"] >= 1
        assert idx[0].tolist() == sorted(rows.tolist())[:10]
        check(q, c, "cos", "bf16", 10, scores, idx)


def test_score_matrix_matches_cos_sim(ts):
    q, c = oracle.inputs(1500, 70, 768, 41, "cos")
    want = oracle.cos_sim(q, c)
    for dtype, tol in (("f32", SCORE_TOL),):
        with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos") as ix:
            got = ix.scores(q)
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= tol
            # ranking-by-full-argsort agrees wherever the fp64 gaps are clear (compare_embeddings.py:105)
            truth = oracle.scores_fp64(*oracle.prepared_inputs(q, c, "cos", dtype))
            top = np.argsort(-got, axis=1)[:, :5]
            oracle.check_topk_against_truth(truth, top, None, 5, gap=GAP)


def test_merge_topk_matches_numpy(ts):
    rng = np.random.default_rng(12)
    nparts, nq, k = 8, 33, 10
    scores = rng.standard_normal((nparts, nq, k)).astype(np.float32)
    idx = rng.permutation(nparts * nq * k).reshape(nparts, nq, k).astype(np.int64) + (1 << 33)
    scores[0, 0, :3] = scores[1, 0, 0]            # ties across parts
    idx[3, 1, 5:] = -1                            # padding entries
    scores[3, 1, 5:] = -np.inf
    got_s, got_i = ts.merge_topk(scores, idx, k)
    for b in range(nq):
        s = scores[:, b, :].reshape(-1)
        i = idx[:, b, :].reshape(-1)
        keep = i >= 0
        order = np.lexsort((i[keep], -s[keep]))[:k]
        assert got_i[b].tolist() == i[keep][order].tolist()
        assert got_s[b].tolist() == s[keep][order].tolist()


def test_sharded_search_equals_whole(ts):
    # linearity of the top-k reduction: merging per-shard answers gives the whole-index answer,
    # bit for bit when every part runs the same kernel (same accumulation order per row)
    q, c = oracle.inputs(40_000, 16, 768, 51, "ip")
    k = 10
    bounds = [0, 9_999, 20_000, 33_333, 40_000]
    for algo in ("mfma", "scan"):
        with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as whole:
            ws, wi = whole.search(q, k, algo=algo)
        parts_s, parts_i = [], []
        for a, b in zip(bounds[:-1], bounds[1:]):
            with ts.TheoremIndex.from_embeddings(c[a:b], dtype="bf16", metric="ip", row_offset=a) as shard:
                s, i = shard.search(q, k, algo=algo)
                parts_s.append(s)
                parts_i.append(i)
        ms, mi = ts.merge_topk(np.stack(parts_s), np.stack(parts_i), k)
        assert np.array_equal(mi, wi) and np.array_equal(ms, ws), algo


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_filtered_search_equals_search_of_the_allowed_rows(ts, dtype):
    """ts_search_filtered: the k best rows whose mask bit is set = an unfiltered oracle search over exactly
    those rows (ids mapped back); masks from dense to empty, n not a multiple of 32."""
    n, d, nq, k = 20011, 768, 6, 10
    q, c = oracle.inputs(n, nq, d, 77, "cos")
    rng = np.random.default_rng(8)
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos") as ix:
        for frac in (1.0, 0.5, 0.03, 0.0004, 0.0):
            mask = rng.random(n) < frac
            if frac == 0.0004:
                mask[-1] = True  # last row, last partial mask word
            allowed = np.flatnonzero(mask)
            scores, idx = ix.search(q, k, mask=mask)
            m = min(k, allowed.size)
            assert (idx[:, m:] == -1).all() and np.isneginf(scores[:, m:]).all()
            if m == 0:
                continue
            assert mask[idx[:, :m]].all()
            qp, cp = oracle.prepared_inputs(q, c[allowed], "cos", dtype)
            truth = oracle.scores_fp64(qp, cp)
            local = np.searchsorted(allowed, idx[:, :m])
            pad = np.full((nq, k - m), -1, dtype=np.int64)
            stats = oracle.check_topk_against_truth(truth, np.concatenate([local, pad], 1), scores, k, gap=GAP,
                                                    score_tol=SCORE_TOL)
            assert stats["recall"] == 1.0
        # unfiltered search afterwards is unaffected by the previous mask
        s0, i0 = ix.search(q, k, algo="scan")
        check(q, c, "cos", dtype, k, s0, i0)


@pytest.mark.parametrize("d", [768, 1024])
def test_filtered_batches_run_on_the_mfma_path(ts, d):
    """A batch behind a host mask that keeps >= 10 % of the rows goes through the MFMA kernel (bit tested in its append
    path, threshold estimates made for the allowed rows); sparser masks through the scan.  Either way: the k best
    ALLOWED rows, exactly."""
    n, nq, k = 60_013, 40, 10
    q, c = oracle.inputs(n, nq, d, 91, "cos")
    rng = np.random.default_rng(6)
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as ix:
        for frac in (0.9, 0.5, 0.15, 0.05, 0.0002):
            mask = rng.random(n) < frac
            mask[-1] = True
            allowed = np.flatnonzero(mask)
            scores, idx = ix.search(q, k, mask=mask)
            m = min(k, allowed.size)
            assert (idx[:, m:] == -1).all() and mask[idx[:, :m]].all()
            qp, cp = oracle.prepared_inputs(q, c[allowed], "cos", "bf16")
            truth = oracle.scores_fp64(qp, cp)
            local = np.full_like(idx, -1)
            local[:, :m] = np.searchsorted(allowed, idx[:, :m])
            stats = oracle.check_topk_against_truth(truth, local, scores, k, gap=GAP, score_tol=SCORE_TOL)
            assert stats["recall"] == 1.0
        s_un, i_un = ix.search(q, k)                 # the next unfiltered batch is unaffected
        check(q, c, "cos", "bf16", k, s_un, i_un)


def test_subset_index_returns_the_ids_of_the_parent(ts):
    """ts_index_subset: a batch search of the copy = the filtered scan of the parent = the oracle over the allowed
    rows, with the parent's global ids (row_offset included); MFMA path and scan path of the copy."""
    n, d, nq, k = 30011, 768, 40, 10
    q, c = oracle.inputs(n, nq, d, 88, "cos")
    rng = np.random.default_rng(2)
    mask = rng.random(n) < 0.6
    mask[0] = mask[-1] = True
    allowed = np.flatnonzero(mask)
    off = 1000
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos", row_offset=off) as ix:
        with ix.subset(mask) as sub:
            assert sub.n == allowed.size
            qp, cp = oracle.prepared_inputs(q, c[allowed], "cos", "bf16")
            truth = oracle.scores_fp64(qp, cp)
            for algo in ("mfma", "scan"):
                scores, idx = sub.search(q, k, algo=algo)
                assert mask[idx - off].all()
                local = np.searchsorted(allowed, idx - off)
                stats = oracle.check_topk_against_truth(truth, local, scores, k, gap=GAP, score_tol=SCORE_TOL)
                assert stats["recall"] == 1.0
            # same answer as the masked scan of the parent (same arithmetic: bitwise)
            s_par, i_par = ix.search(q[:4], k, mask=mask)
            s_sub, i_sub = sub.search(q[:4], k, algo="scan")
            assert np.array_equal(i_par, i_sub) and np.array_equal(s_par, s_sub)
            with pytest.raises(ts.TSearchError):
                sub.upload(c[:1], 0)
            with pytest.raises(ts.TSearchError):
                sub.subset(np.arange(3))
        with ix.subset(np.array([off + 5, off + 9])) as two:
            s2, i2 = two.search(q[:2], 5)
            assert set(i2[:, :2].ravel()) == {off + 5, off + 9} and (i2[:, 2:] == -1).all()
        with ix.subset(np.zeros(n, bool)) as none:
            s0, i0 = none.search(q[:2], 3)
            assert (i0 == -1).all()
        with pytest.raises(ts.TSearchError):
            ix.subset(np.array([off + 9, off + 5]))   # not ascending
        with pytest.raises(ts.TSearchError):
            ix.subset(np.array([5]))                  # below row_offset


def test_showcase_filters_return_topk_where_the_reference_pool_runs_dry(ts):
    """filters.search_filtered against the restated app loop (app_showcase_model.py:93-129): same rows whenever the
    top-200 pool held top_k matches; when it ran dry, ours continues with the next best matching rows."""
    from filters_common import filter_states, make_theorems
    from theoremsearch_amd import filters as flt
    n, d = 6000, 768
    q, c = oracle.inputs(n, 1, d, 31, "cos")
    data = make_theorems(n)
    qp, cp = oracle.prepared_inputs(q, c, "cos", "f32")
    cos = oracle.scores_fp64(qp, cp)[0]
    saw_dry = False
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        for name, f in filter_states(top_k=10).items():
            got = flt.search_filtered(ix, q, data, f)
            ref_rows, dry = oracle.showcase_filter_search(cos, data, f)
            full_rows, _ = oracle.showcase_filter_search(cos, data, f, pool=n)
            got_ids = [next(i for i in full_rows if data[i] is g["info"]) for g in got]
            assert got_ids == full_rows, name
            assert got_ids[: len(ref_rows)] == ref_rows, name
            if dry and len(full_rows) > len(ref_rows):
                saw_dry = True
            for g, i in zip(got, got_ids):
                assert abs(g["similarity"] - cos[i]) <= SCORE_TOL
    assert saw_dry


@pytest.mark.parametrize("dtype,d,n", [("f32", 768, 20011), ("bf16", 768, 20011), ("bf16", 1024, 5003), ("f32", 384, 3001),
                                       ("bf16", 384, 3001), ("f32", 512, 2001), ("bf16", 512, 2001), ("f32", 100, 999)])
def test_rank_of_matches_position_in_the_full_ranking(ts, dtype, d, n):
    """ts_rank_of = position of the row in the full ranking of the fp64 truth, wherever the truth separates the
    target from its neighbours by more than GAP; consistent with ts_search (rank r <=> idx[r] == row)."""
    nq = 9
    q, c = oracle.inputs(n, nq, d, 41, "cos")
    qp, cp = oracle.prepared_inputs(q, c, "cos", dtype)
    truth = oracle.scores_fp64(qp, cp)
    rng = np.random.default_rng(12)
    order = np.argsort(-truth, axis=1, kind="stable")
    # targets: the best row, the worst row, a top-10 row, random rows, one outside the index
    targets = np.array([order[0, 0], order[1, -1], order[2, 7]] + [int(x) for x in rng.integers(0, n, nq - 4)] + [n + 5])
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="cos") as ix:
        ranks, scores = ix.rank_of(q, targets)
        exp = oracle.rank_of(truth, targets)
        assert ranks[-1] == -1 and np.isnan(scores[-1]) and exp[-1] == -1
        s_idx = ix.search(q, 10, algo="scan")[1]
        for b in range(nq - 1):
            t = targets[b]
            assert abs(scores[b] - truth[b, t]) <= SCORE_TOL
            close = np.sum(np.abs(truth[b] - truth[b, t]) <= GAP) - 1   # rows the truth cannot separate from the target
            assert abs(int(ranks[b]) - int(exp[b])) <= close, (b, ranks[b], exp[b])
            if ranks[b] < 10:
                assert s_idx[b, ranks[b]] == t
        assert ranks[0] == 0 and ranks[1] == n - 1 and ranks[2] == 7


def test_count_above_summed_over_shards_is_the_rank(ts):
    """ts_count_above: per-shard counts of rows that rank before (score, global id) add up to ts_rank_of of the whole
    index - including an exact tie that straddles the shard boundary."""
    n, d, nq = 9001, 768, 6
    q, c = oracle.inputs(n, nq, d, 61, "cos")
    c[10] = c[n - 3]                                   # duplicate rows on different shards
    rows = np.array([10, n - 3, 0, n - 1, 4500, 4499])
    cuts = [0, 3000, 4500, n]
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos") as whole:
        want, want_scores = whole.rank_of(q, rows)
        shards = [ts.TheoremIndex.from_embeddings(c[a:b], dtype="bf16", metric="cos", row_offset=a) for a, b in zip(cuts, cuts[1:])]
        try:
            total = np.zeros(nq, np.int64)
            for sh in shards:
                cnt = sh.count_above(q, want_scores, rows)
                assert (cnt >= 0).all()
                total += cnt
            assert np.array_equal(total, want)
            # one query, both copies of the duplicated row: the higher id ranks directly behind the lower one
            q2 = np.repeat(q[:1], 2, axis=0)
            r2, s2 = whole.rank_of(q2, [10, n - 3])
            assert s2[0] == s2[1] and r2[1] == r2[0] + 1
            t2 = sum(sh.count_above(q2, s2, [10, n - 3]) for sh in shards)
            assert np.array_equal(t2, r2)
            nan_cnt = shards[0].count_above(q[:1], [np.nan], [5])
            assert nan_cnt[0] == -1
        finally:
            for sh in shards:
                sh.close()


def test_metrics_from_the_index_equal_metrics_from_the_matrix(ts):
    """IndexRanking (top-k searches + counting pass) through the six metric functions = the same functions on the
    full similarity matrix (the reference's formulation, compare_embeddings.py:55-371)."""
    from theoremsearch_amd import compare_embeddings as ce
    n, nq, d = 4000, 24, 768
    q, c = oracle.inputs(n, nq, d, 53, "cos")
    rng = np.random.default_rng(4)
    qrels = {}
    qp, cp = oracle.prepared_inputs(q, c, "cos", "f32")
    truth = oracle.scores_fp64(qp, cp)
    order = np.argsort(-truth, axis=1, kind="stable")
    for i in range(nq):
        exact = int(order[i, [0, 1, 2, 5, 40, 900][i % 6]])      # exact doc at assorted depths of the ranking
        rels = {exact: 1}
        for j in rng.integers(0, n, 6):
            rels.setdefault(int(j), 0.5)
        for j in order[i, 1:4]:
            rels.setdefault(int(j), 0.5)
        qrels[i] = rels
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        sim = ix.scores(q)
        ranking = ce.IndexRanking(ix, q)
        for k in (1, 3, 10):
            for fn in (ce.precision_at_k, ce.hit_at_k, ce.mrr_at_k, ce.ndcg_at_k, ce.err_at_k, ce.q_measure_at_k):
                assert fn(ranking, qrels, k=k) == pytest.approx(fn(sim, qrels, k=k), abs=1e-12), (fn.__name__, k)
        assert ce.mrr_at_k(ranking, qrels, k=None) == pytest.approx(ce.mrr_at_k(sim, qrels, k=None), abs=1e-12)
        assert ce.mrr_at_k(ranking, qrels, k=None) == pytest.approx(oracle.mrr_at_k(truth, qrels, k=None), abs=1e-12)


def test_empty_and_invalid_arguments(ts):
    from theoremsearch_amd import _ffi
    c = np.random.default_rng(1).standard_normal((10, 16), dtype=np.float32)
    with ts.TheoremIndex.from_embeddings(c) as ix:
        s, i = ix.search(np.zeros((0, 16), np.float32), 3)
        assert s.shape == (0, 3)
        with pytest.raises(_ffi.TSearchError):
            ix.search(c[:1], 0)
        with pytest.raises(_ffi.TSearchError):
            ix.search(c[:1], 257)
        with pytest.raises(ValueError):
            ix.search(np.zeros((1, 8), np.float32), 3)
        with pytest.raises(_ffi.TSearchError):
            ix.search(c[:1], 3, algo="mfma")          # not a bf16 / d=768 index
    with ts.TheoremIndex(0, 16) as empty:
        s, i = empty.search(c[:2], 4)
        assert (i == -1).all() and np.isneginf(s).all()


# ---- batched fp32: the exact-fp32 matrix paths (v_mfma_f32_16x16x4_f32, v_mfma_f32_32x32x2_f32) ---------------------
@pytest.mark.parametrize("shape", [16, 32])
@pytest.mark.parametrize("n,nq,k,metric", [(200_000, 130, 10, "cos"), (70_001, 73, 200, "ip"), (16_384, 13, 1, "cos"),
                                           (300_000, 256, 10, "ip"), (120_000, 64, 10, "cos"), (90_000, 193, 33, "ip")])
def test_fp32_batches_run_on_the_fp32_mfma_path(ts, n, nq, k, metric, shape):
    """The reference's evaluation is an fp32 [Q x N] matrix with Q ~ 73 (compare_embeddings.py:61,105): batches of an fp32
    index go through an fp32 matrix kernel (64 / 128 queries per launch on the 16x16x4 form, the default; 128 on the
    32x32x2 kernel) instead of ceil(Q / 4) scan passes - same protocol, same answers as the scan at pinned ranks."""
    q, c = oracle.inputs(n, nq, 768, 1000 + nq, metric)
    c[n // 2] = c[3]                                          # an exact tie
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric=metric) as ix:
        ix.set_option("TS_MFMA_F32", shape)
        scores, idx, st = ix.search(q, k, return_stats=True)             # auto
        assert st["algo"] == 2 and st["fallback_queries"] == 0, st
        check(q, c, metric, "f32", k, scores, idx)
        s2, i2 = ix.search(q, k, algo="mfma")
        assert np.array_equal(i2, idx) and np.array_equal(s2, scores)    # deterministic
        s3, i3 = ix.search(q[:8], k, algo="scan")
        assert np.array_equal(i3, idx[:8])
        assert np.allclose(s3, scores[:8], atol=1e-6)
        # a host mask that keeps a third of the rows: still the matrix path, still exact
        mask = np.random.default_rng(n).random(n) < 0.34
        ms, mi = ix.search(q, k, mask=mask)
        keep = np.flatnonzero(mask)
        qp, cp = oracle.prepared_inputs(q, c[keep], metric, "f32")
        local = np.where(mi >= 0, np.searchsorted(keep, np.maximum(mi, 0)), -1)
        stats = oracle.check_topk_against_truth(oracle.scores_fp64(qp, cp), local, ms, k)
        assert stats["recall"] == 1.0 and mask[mi[mi >= 0]].all()


@pytest.mark.parametrize("n,nq,k,metric", [(150_000, 70, 10, "cos"), (40_000, 129, 50, "ip")])
def test_fp32_batches_at_d1024_run_on_the_fp32_mfma_path(ts, n, nq, k, metric):
    """The Qwen-sized tables (vector(1024), rds_schema.sql:50-56) in fp32: 64 queries per launch on v_mfma_f32_16x16x4_f32."""
    q, c = oracle.inputs(n, nq, 1024, 2000 + nq, metric)
    c[n - 7] = c[11]
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric=metric) as ix:
        scores, idx, st = ix.search(q, k, return_stats=True)
        assert st["algo"] == 2 and st["fallback_queries"] == 0, st
        check(q, c, metric, "f32", k, scores, idx)
        s2, i2 = ix.search(q, k, algo="mfma")
        assert np.array_equal(i2, idx) and np.array_equal(s2, scores)
        s3, i3 = ix.search(q[:5], k, algo="scan")
        assert np.array_equal(i3, idx[:5]) and np.allclose(s3, scores[:5], atol=1e-6)


def test_small_fp32_batches_stay_on_the_scan(ts):
    q, c = oracle.inputs(50_000, 12, 768, 77, "cos")
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
        s4, i4, st = ix.search(q[:4], 10, return_stats=True)          # one scan pass serves four queries
        assert st["algo"] == 1
        s5, i5, st = ix.search(q[:5], 10, return_stats=True)          # five would need two: the matrix kernel is cheaper
        assert st["algo"] == 2
        check(q[:5], c, "cos", "f32", 10, s5, i5)
        assert np.array_equal(i5[:4], i4)
        ix.set_option("TS_MFMA_F32", 0)
        _, _, st = ix.search(np.tile(q, (3, 1)), 10, return_stats=True)
        assert st["algo"] == 1


@pytest.mark.parametrize("dtype,d,n,nq,k", [("bf16", 768, 300_000, 256, 10), ("bf16", 1024, 150_000, 100, 50), ("bf16", 384, 90_000, 70, 10),
                                           ("f32", 768, 120_000, 128, 10), ("f32", 1024, 80_000, 33, 200)])
def test_dense_threshold_sample_and_the_list_form_give_the_same_answers(ts, dtype, d, n, nq, k):
    """Round 3: the threshold sample is a dense score matrix from a kernel of its own (kernels_sample.h) + one select per
    query; round 2's form (the full-pass kernel over the sample, candidates gathered from lane-private lists) stays
    selectable (TS_MFMA_SAMPLE=0).  Both feed the same estimates; the answers are exact and identical either way, also
    behind a row mask (the sample sees the allowed rows only)."""
    q, c = oracle.inputs(n, nq, d, 500 + d + nq, "ip")
    mask = np.random.default_rng(8).random(n) < 0.5
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
        s1, i1, st1 = ix.search(q, k, algo="mfma", return_stats=True)
        m1 = ix.search(q, k, algo="mfma", mask=mask)
        ix.set_option("TS_MFMA_SAMPLE", 0)
        s0, i0, st0 = ix.search(q, k, algo="mfma", return_stats=True)
        m0 = ix.search(q, k, algo="mfma", mask=mask)
        ix.set_option("TS_MFMA_SAMPLE", None)
        assert st1["algo"] == st0["algo"] == 2 and st1["levels"] == st0["levels"]
        assert st1["fallback_queries"] == 0 and st0["fallback_queries"] == 0
        # (d = 384: the list form runs the 32x32 kernel, whose fp32 sums add the same exact products in another order)
        assert np.array_equal(i1, i0) and (np.array_equal(s1, s0) if d != 384 else np.allclose(s1, s0, atol=2e-7))
        assert np.array_equal(m1[1], m0[1]) and np.allclose(m1[0], m0[0], atol=2e-7) and mask[m1[1]].all()
        # the two samples see the same rows: candidate counts of the full pass agree to within the estimates' noise
        # (round 4: the dense form's select is a one-wave kernel with its own reductions and cut - the same thresholds)
        assert 0.9 < st1["candidates"] / max(1, st0["candidates"]) < 1.11, (st1, st0)
        check(q, c, "ip", dtype, k, s1, i1)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_threshold_sample_of_degenerate_score_distributions(ts, dtype):
    """Round 4: the sample's select cuts at mean + z sd and, where that leaves too few or too many survivors, finds the cut
    by bisection on the ordered scores.  The shapes that send it there: every row the same (sd = 0: every sample score
    equal), two values only (a pile of equal scores right at the cut, larger than the sort's capacity), a mask that leaves the
    sample fewer live rows than k, and k = 200 over a plain corpus (hundreds of survivors).  The answers stay exact (ties:
    lowest rows first) whatever the thresholds come out as."""
    n, d, nq = 40_000, 768, 70
    rng = np.random.default_rng(5)
    q = rng.standard_normal((nq, d), dtype=np.float32) * np.float32(1.0 / np.sqrt(d))
    one = rng.standard_normal(d).astype(np.float32) * np.float32(1.0 / np.sqrt(d))
    # (a) every row the same
    c = np.tile(one, (n, 1))
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
        s, i = ix.search(q, 10, algo="mfma")
        assert np.array_equal(i, np.tile(np.arange(10), (nq, 1)))
        check(q, c, "ip", dtype, 10, s, i)
    # (b) two values: a third of the rows score higher than the rest for every query that likes `one`
    other = rng.standard_normal(d).astype(np.float32) * np.float32(1.0 / np.sqrt(d))
    c = np.tile(other, (n, 1))
    c[rng.random(n) < 0.33] = one
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
        s, i = ix.search(q, 10, algo="mfma")
        check(q, c, "ip", dtype, 10, s, i)
    # (c) plain rows: a mask that keeps a tenth of them with k = 200 (a 4,096-row sample sees ~400 live rows), and k = 200 unmasked
    q2, c2 = oracle.inputs(n, nq, d, 97, "ip")
    mask = rng.random(n) < 0.1
    with ts.TheoremIndex.from_embeddings(c2, dtype=dtype, metric="ip") as ix:
        s, i, st = ix.search(q2, 200, algo="mfma", return_stats=True)
        assert st["fallback_queries"] == 0, st
        check(q2, c2, "ip", dtype, 200, s, i)
        s, i = ix.search(q2, 200, mask=mask)
        rows = np.flatnonzero(mask)
        local = np.searchsorted(rows, i)
        assert (rows[local] == i).all()
        check(q2, c2[rows], "ip", dtype, 200, s, local)
