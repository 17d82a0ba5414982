"""Parity at BASELINE.json's FULL sizes, every query through the oracle's protocol (pinned ranks exact, near-tie runs as
sets, scores within 1e-5 of the fp64 truth of the same operands):

  configs[2]  10M x 768 bf16, 256 queries, top-10, algo = auto  (the level plan, estimated thresholds and tail-fit gate
              of the MFMA path exactly as bench.py runs them; no query may need the exact re-run)
  configs[4]  encoder-in-loop over that same 10M index: encoder forward + fused pooling on the device, embeddings handed
              to the search by device pointer, all 256 answers checked
  configs[1]  1M x 768 fp32, one query, top-10                 (the streaming scan)
  configs[3]  50M x 768 bf16 row-sharded 8 ways (ts_shards_*: routed uploads, eight searches, packed per-shard top-10,
              merge), 256 queries: bit-identical to ONE index over the same 50M rows, and checked against the committed
              fp64 truth of the 50M-row corpus.  A one-GPU box holds all eight shards on device 0 (288 GB of HBM: the
              shards and the whole index together are 154 GB); the exchange is then device copies instead of RCCL

plus a fixed-seed slice of the randomised sweep of tests/stress_parity.py, small and --big (1M-2.5M rows, clusters,
masks, k up to 256).  The fp64 truth is accumulated chunk by chunk on the host (oracle.ChunkedTruth) from the same
generator bench.py uses (synthetic.synth_chunk); nothing here reads /root/reference.

Matches the batched form of the reference's evaluation, util.cos_sim(q, s) + np.argsort(-S) per query
(compare_embeddings.py:61,105), at the shapes the north-star names.
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle

sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu

K = 10
D = 768


@pytest.fixture(scope="module")
def ts():
    import theoremsearch_amd as ts
    from theoremsearch_amd import _ffi
    assert _ffi.device_count() > 0, "GPU tests need a HIP device"
    return ts


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def build_index(ts, rows_total, dtype, keep_rows=False, digests=None, also=None):
    """Corpus of bench.py (chunks of synthetic.synth_chunk, stored as given: metric ip on unit rows).  ``keep_rows``: also
    return the chunks' host arrays (the truth pass then does not generate them a second time).  ``digests``: a dict that
    receives the SHA-256 of every chunk as generated here (what a committed truth fixture is matched against).  ``also``:
    a second receiver of every chunk with an ``upload(rows, row0)`` method (the row-sharded form of the same corpus)."""
    import synthetic
    bf16 = dtype == "bf16"
    ch = synthetic.CHUNK_ROWS
    ix = ts.TheoremIndex(rows_total, D, dtype=dtype, metric="ip")
    chunks = list(range((rows_total + ch - 1) // ch))
    kept = {}

    def make(c):
        data = synthetic.synth_chunk(c, ch, D, bf16=bf16)
        hi = min(rows_total, (c + 1) * ch)
        ix.upload(data[: hi - c * ch], c * ch)
        if also is not None:
            also.upload(data[: hi - c * ch], c * ch)
        if digests is not None:
            digests[c] = _digest(data[: hi - c * ch])
        if keep_rows:
            kept[c] = data[: hi - c * ch]
        return c

    with ThreadPoolExecutor(_threads()) as ex:
        list(ex.map(make, chunks))
    return (ix, chunks, kept) if keep_rows else (ix, chunks)


def chunked_truth_check(q_host, dtype, rows_total, chunks, idx, scores, k):
    return chunked_truth_check_many(dtype, rows_total, chunks, [(q_host, idx, scores)], k)[0]


def chunked_truth_check_many(dtype, rows_total, chunks, answers, k, kept=None):
    """``answers``: list of (queries as stored [bf16 bits / f32], idx, scores); one pass over the corpus checks them all.
    ``kept``: the chunks' host arrays from `build_index` (generated again otherwise)."""
    import synthetic
    bf16 = dtype == "bf16"
    ch = synthetic.CHUNK_ROWS
    truths = [oracle.ChunkedTruth(oracle.bf16_bits_to_f32(q) if bf16 else q, idx, k) for q, idx, _ in answers]

    def chunk_scores(c):
        data = kept[c] if kept is not None else synthetic.synth_chunk(c, ch, D, bf16=bf16)[: min(rows_total, (c + 1) * ch) - c * ch]
        vals = (oracle.bf16_bits_to_f32(data) if bf16 else data).astype(np.float64)
        return c, [t.q64 @ vals.T for t in truths]

    with ThreadPoolExecutor(max(1, _threads() // 2)) as ex:
        for c, ss in ex.map(chunk_scores, chunks):
            for t, s in zip(truths, ss):
                t.add_scores(s, c * ch)
    return [t.check(scores, gap=1e-6, score_tol=1e-5) for t, (_, _, scores) in zip(truths, answers)]


def _digest(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cached_truth_check(fixture, q_bits, idx, scores, k, digests, row_values):
    """A committed fp64 truth (tests/golden/fullsize_c3_truth.npz / fullsize_c4_truth.npz, oracle/gen_fullsize_truth.py: the
    k + 64 best rows of each benchmark query over the 10M / 50M x 768 bf16 corpus) in place of a pass over the corpus -
    after the digests of the queries and of EVERY corpus chunk have been matched against the rows generated HERE
    (``digests``: chunk id -> SHA-256, filled by `build_index`).  Returns None when the file does not describe this box's
    inputs (the caller then computes the truth itself).  The fp64 score of a returned row comes from the cached list, or -
    a row outside the k + 64 best - from the row itself (``row_values(r)``: its bf16 bits)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    if int(z["k"]) != k or str(z["query_digest"]) != _digest(q_bits):
        return None
    ids, want = z["chunk_ids"].tolist(), z["chunk_digests"].tolist()
    if len(ids) != len(digests):
        return None
    for c, w in zip(ids, want):
        if digests.get(int(c)) != str(w):
            return None
    t = oracle.ChunkedTruth(oracle.bf16_bits_to_f32(q_bits), idx, k)
    t.best_s, t.best_i, t.n = z["best_s"], z["best_i"], int(z["n"])
    for b in range(idx.shape[0]):
        known = {int(j): float(v) for j, v in zip(t.best_i[b], t.best_s[b])}
        for j in range(idx.shape[1]):
            r = int(idx[b, j])
            if r in known:
                t.got_s[b, j] = known[r]
            elif 0 <= r < t.n:
                row = oracle.bf16_bits_to_f32(row_values(r)).astype(np.float64)
                t.got_s[b, j] = float(t.q64[b] @ row)
    return t.check(scores, gap=1e-6, score_tol=1e-5)


def encoder_in_loop_step(ix, nq, seq_len=32):
    """One step of configs[4]: encoder forward + fused pooling on the device -> search by device pointer.  Returns the
    bf16 bits of the embeddings the index multiplied (what the oracle needs), and the answer."""
    import torch
    from theoremsearch_amd.encoder import SentenceEncoder
    enc = SentenceEncoder(allow_random_init=True, num_layers=2)
    g = torch.Generator(device="cpu").manual_seed(5678)
    tok = torch.randint(1000, 30000, (nq, seq_len), generator=g).cuda()
    tok[:, 0], tok[:, -1] = 101, 102
    mask = torch.ones_like(tok)
    st = torch.cuda.Stream()
    out_s = torch.empty((nq, K), dtype=torch.float32, device="cuda")
    out_i = torch.empty((nq, K), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):                                             # twice: the second step reuses every buffer
        with torch.cuda.stream(st), torch.inference_mode():
            hidden = enc.model(input_ids=tok, attention_mask=mask).last_hidden_state
            emb = enc.pool(hidden, mask, True)
            ix.search_device(emb.data_ptr(), "f32", nq, K, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
    st.synchronize()
    bits = oracle.f32_to_bf16_bits(emb.cpu().numpy())               # the search rounds fp32 queries to bf16 (RNE)
    return bits, out_s.cpu().numpy(), out_i.cpu().numpy()


@pytest.mark.timeout(1500)
def test_config2_10m_bf16_batch256_every_query(ts):
    import synthetic
    rows_total, nq = 10_000_000, 256
    t0 = time.time()
    digests = {}
    ix, chunks, kept = build_index(ts, rows_total, "bf16", keep_rows=True, digests=digests)
    q = synthetic.synth_queries(0, nq, D, bf16=True)
    t1 = time.time()
    try:
        scores, idx, st = ix.search(q, K, algo="auto", return_stats=True)
        assert st["algo"] == 2 and st["levels"] == 2, st           # the MFMA path with one sample level, as bench.py runs it
        assert st["fallback_queries"] == 0, st                      # no query needed the exact re-run
        assert st["candidates"] >= nq * K
        # deterministic: a second search returns the same bits
        s2, i2 = ix.search(q, K, algo="auto")
        assert np.array_equal(idx, i2) and np.array_equal(scores, s2)
        # and the exact scan agrees on a slice of the batch (4 queries = one pass over the corpus)
        s3, i3 = ix.search(q[:4], K, algo="scan")
        assert np.array_equal(i3, idx[:4])
        # configs[4], encoder-in-loop over the SAME 10M index: the sentence encoder's forward (PyTorch-ROCm; random-init
        # stand-in, no weights offline) + the fused pooling kernel produce 256 query embeddings on the device, which
        # are handed to the search by pointer (no host hop) - the loop bench.py --workload c5 times
        enc_q, enc_scores, enc_idx = encoder_in_loop_step(ix, nq)
    finally:
        ix.close()
    t2 = time.time()
    # the benchmark's own queries: the committed fp64 truth when it describes the rows generated here; the encoder's
    # embeddings are this run's own: one pass over the kept rows
    ch = synthetic.CHUNK_ROWS
    stats = cached_truth_check("fullsize_c3_truth.npz", q, idx, scores, K, digests, lambda r: kept[r // ch][r % ch])
    answers = [(enc_q, enc_idx, enc_scores)] + ([] if stats is not None else [(q, idx, scores)])
    checked = chunked_truth_check_many("bf16", rows_total, chunks, answers, K, kept=kept)
    enc_stats = checked[0]
    kept.clear()                                                   # 15 GB of host rows: nothing below needs them
    cached = stats is not None
    if not cached:
        stats = checked[1]
    print(f"[fullsize] build {t1 - t0:.0f}s, search {t2 - t1:.0f}s, truth {time.time() - t2:.0f}s (benchmark queries: "
          f"{'committed fixture' if cached else 'computed here'}), {stats}, encoder-in-loop {enc_stats}")
    assert stats["recall"] == 1.0 and stats["positions"] == nq * K
    assert stats["pinned"] >= 0.99 * stats["positions"]
    assert enc_stats["recall"] == 1.0 and enc_stats["positions"] == nq * K


@pytest.mark.timeout(1500)
def test_config3_50m_8_shards(ts):
    """BASELINE.json configs[3] at its size: 50M x 768 bf16 rows, row-sharded 8 ways behind ts_shards_* (one
    process, one shard and one stream per device, packed per-shard top-k, one exchange, merge), 256 queries, top-10.  The
    reference's only multi-device call is encode_multi_process (ec2/generate_embeddings/embeddings.py:32); its search is one
    sequential scan (streamlit_app.py:253-283), so the sharded answer must be THE answer of the whole corpus:
      * bit-identical (ids and scores) to one TheoremIndex over the same 50M rows,
      * every query through the oracle's protocol against the committed fp64 truth of the 50M-row corpus
        (tests/golden/fullsize_c4_truth.npz, trusted after the SHA-256 of every one of the 200 chunks generated here matches),
      * every shard contributes (the global top-10 of 256 queries over random rows draw from all eight row ranges).
    One GPU holds the eight shards (76.8 GB) beside the whole index (76.8 GB); the exchange is device copies there and
    ncclAllGather on eight devices - the searches, the packed blocks and the merge are the same code."""
    import synthetic
    from theoremsearch_amd.distributed import Shards
    rows_total, nq, G = 50_000_000, 256, 8
    ch = synthetic.CHUNK_ROWS
    t0 = time.time()
    digests = {}
    sh = Shards(rows_total, D, G, dtype="bf16", metric="ip", devices=[0] * G)
    whole = None
    try:
        assert [sh.bounds(g) for g in range(G)] == [(rows_total * g // G, rows_total * (g + 1) // G) for g in range(G)]
        whole, chunks = build_index(ts, rows_total, "bf16", digests=digests, also=sh)
        t1 = time.time()
        q = synthetic.synth_queries(0, nq, D, bf16=True)
        scores, idx = sh.search(q, K)
        s2, i2 = sh.search(q, K)                                   # deterministic
        assert np.array_equal(idx, i2) and np.array_equal(scores, s2)
        want_s, want_i, st = whole.search(q, K, algo="auto", return_stats=True)
        assert st["algo"] == 2 and st["fallback_queries"] == 0, st
    finally:
        sh.close()
        if whole is not None:
            whole.close()
    t2 = time.time()
    assert np.array_equal(idx, want_i), "the sharded search and the one-index search disagree on ids"
    assert np.array_equal(scores, want_s), "the sharded search and the one-index search disagree on score bits"
    owners = np.unique(idx // (rows_total // G))
    assert owners.tolist() == list(range(G)), f"shards that contributed to the answers: {owners}"
    stats = cached_truth_check("fullsize_c4_truth.npz", q, idx, scores, K, digests,
                               lambda r: synthetic.synth_chunk(r // ch, ch, D, bf16=True)[r % ch])
    cached = stats is not None
    if not cached:                                                 # another generator stream on this box: the truth computed here
        stats = chunked_truth_check(q, "bf16", rows_total, chunks, idx, scores, K)
    print(f"[fullsize c4] build {t1 - t0:.0f}s, searches {t2 - t1:.0f}s, truth {time.time() - t2:.0f}s "
          f"({'committed fixture' if cached else 'computed here'}), {stats}")
    assert stats["recall"] == 1.0 and stats["positions"] == nq * K
    assert stats["pinned"] >= 0.99 * stats["positions"]


@pytest.mark.timeout(600)
def test_config1_1m_f32_batch1(ts):
    import synthetic
    rows_total = 1_000_000
    ix, chunks = build_index(ts, rows_total, "f32")
    try:
        for b in range(3):                                        # three different single queries
            q = synthetic.synth_queries(b, 1, D, bf16=False)
            scores, idx, st = ix.search(q, K, return_stats=True)
            assert st["algo"] == 1
            stats = chunked_truth_check(q, "f32", rows_total, chunks, idx, scores, K)
            assert stats["recall"] == 1.0 and stats["pinned"] == K
    finally:
        ix.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed,cases", [(101, 14), (202, 14), (303, 14)])
def test_random_sweep_fixed_seeds(ts, seed, cases):
    """tests/stress_parity.py with fixed seeds: d in {40, 384, 512, 768, 1024}, both dtypes and metrics, 1..300 queries,
    k up to 256, every algorithm, masks, duplicated rows."""
    import stress_parity
    rng = np.random.default_rng(seed)
    for c in range(cases):
        print(stress_parity.one_case(rng, False, c, watchdog=False))


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("seed,cases", [(11, 3), (12, 3)])
def test_random_sweep_big_fixed_seeds(ts, seed, cases):
    """The --big sweep with fixed seeds: bf16 x 768 corpora of 1M-2.5M rows through the MFMA path, half of them with a
    cluster of 4 % of the rows pulled towards the queries, most behind host masks, k up to 256."""
    import stress_parity
    rng = np.random.default_rng(seed)
    for c in range(cases):
        print(stress_parity.one_case(rng, True, c, watchdog=False))
