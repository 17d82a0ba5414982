"""GPU tests of the host-side API added in round 2: index growth (reserve / append), the sharded search behind the C ABI
(ts_shards_*, ts_comm_*), ordering of calls across streams, per-handle options.  Everything goes through libtsearch.so."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ts():
    import theoremsearch_amd as ts
    from theoremsearch_amd import _ffi
    assert _ffi.device_count() > 0, "GPU tests need a HIP device"
    return ts


def check(q, c, metric, dtype, k, scores, idx):
    qp, cp = oracle.prepared_inputs(q, c, metric, dtype)
    stats = oracle.check_topk_against_truth(oracle.scores_fp64(qp, cp), idx, scores, k)
    assert stats["recall"] == 1.0
    return stats


# ---- growth -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,metric", [("f32", "cos"), ("bf16", "ip")])
def test_append_grows_the_index_and_searches_like_a_rebuild(ts, dtype, metric):
    """New slogan_ids of the upsert pipeline (ec2/generate_embeddings/__main__.py:85-99: INSERT ... ON CONFLICT DO
    UPDATE) are appended; the grown index answers bit for bit like one built from all the rows at once."""
    q, c = oracle.inputs(40_000, 9, 768, 5, metric)
    with ts.TheoremIndex.from_embeddings(c[:1000], dtype=dtype, metric=metric) as ix:
        assert ix.append(c[1000:1003]) == 1000            # fits the padding of the first allocation
        assert ix.append(c[1003:20_000]) == 1003          # forces a move to a larger allocation
        ix.reserve(40_000)
        assert ix.append(c[20_000:]) == 20_000
        assert ix.n == 40_000
        scores, idx = ix.search(q, 10)
        with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric=metric) as ref:
            rs, ri = ref.search(q, 10)
            assert np.array_equal(ix.download(), ref.download())
        assert np.array_equal(idx, ri) and np.array_equal(scores, rs)
        check(q, c, metric, dtype, 10, scores, idx)
        # an UPDATE of an existing row is an upload at its slot
        ix.upload(c[7:8], 30_000)
        s2, i2 = ix.search(c[7:8], 2)
        assert set(i2[0].tolist()) == {7, 30_000}


def test_append_is_refused_while_views_exist_and_on_derived_indexes(ts):
    from theoremsearch_amd import _ffi
    q, c = oracle.inputs(2000, 2, 768, 6, "ip")
    with ts.TheoremIndex.from_embeddings(c[:300], metric="ip") as ix:
        v = ix.view()
        with pytest.raises(_ffi.TSearchError) as e:
            ix.append(c[300:2000])                         # needs a move: the view holds the old rows
        assert e.value.code == -5 and ix.n == 300
        with pytest.raises(_ffi.TSearchError):
            v.append(c[300:301])
        with pytest.raises(_ffi.TSearchError) as e:
            ix.close()                                     # the view reads the owner's rows: destroy refused, handle kept
        assert e.value.code == -5
        sv, iv = v.search(q, 3)
        assert np.array_equal(iv, ix.search(q, 3)[1])
        v.close()
        assert ix.append(c[300:2000]) == 300 and ix.n == 2000
        sub = ix.subset(np.arange(0, 2000, 2))
        with pytest.raises(_ffi.TSearchError):
            sub.append(c[:1])
        assert sub.download().shape == (1000, 768)         # reading a subset index is allowed
        sub.close()


def test_append_device_rows_with_row_offset(ts):
    import torch
    q, c = oracle.inputs(5000, 3, 768, 8, "cos")
    with ts.TheoremIndex(0, 768, dtype="bf16", metric="cos", row_offset=1_000_000) as ix:
        dev = torch.from_numpy(c).cuda()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            first = ix.append_device(dev.data_ptr(), "f32", 768, 5000, s.cuda_stream)
        assert first == 1_000_000 and ix.n == 5000
        scores, idx = ix.search(q, 5)                      # a call on another stream: ordered behind the append
        check(q, c, "cos", "bf16", 5, scores, idx - 1_000_000)


def test_attach_device_rows_zero_copy(ts):
    """ts_index_attach_device: the index searches a tensor in place (bf16 rows a torch kernel wrote) - same answer as an
    index that was uploaded; too small an allocation and host pointers are refused."""
    import torch
    from theoremsearch_amd import _ffi
    n, d = 100_000, 768
    q, c = oracle.inputs(n, 40, d, 31, "ip")
    cap = (n + 255) // 256 * 256
    dev = torch.zeros((cap, d), dtype=torch.bfloat16, device="cuda")
    dev[:n] = torch.from_numpy(c).cuda().to(torch.bfloat16)            # RNE, like the library's own rounding
    torch.cuda.synchronize()
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ref, ts.TheoremIndex(n, d, dtype="bf16", metric="ip") as ix:
        want_s, want_i = ref.search(q, 10)
        with pytest.raises(_ffi.TSearchError):
            ix.attach_device(dev.data_ptr(), n)                         # not rounded up to 256 rows
        with pytest.raises(_ffi.TSearchError):
            ix.attach_device(c.ctypes.data, cap)                        # host memory
        ix.attach_device(dev.data_ptr(), cap, keep_alive=dev)
        assert np.array_equal(ix.download(), ref.download())
        for algo in ("mfma", "scan"):
            s, i = ix.search(q, 10, algo=algo)
            assert np.array_equal(i, want_i) and np.allclose(s, want_s, atol=1e-6)
        with pytest.raises(_ffi.TSearchError):
            ix.append(c[:500])                                          # past the attached allocation: it cannot grow


# ---- ordering across streams ---------------------------------------------------------------------------------------
def test_calls_on_different_streams_share_the_scratch_safely(ts):
    """Two searches enqueued back to back on two different streams of one handle, device outputs: the second must not
    overwrite scratch the first still reads (per-handle scratch across streams)."""
    import torch
    q, c = oracle.inputs(120_000, 256, 768, 9, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        want_s, want_i = ix.search(q, 10)
        qa = torch.from_numpy(q).cuda()
        qb = torch.from_numpy(q[::-1].copy()).cuda()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = [(torch.empty((256, 10), dtype=torch.float32, device="cuda"), torch.empty((256, 10), dtype=torch.int64, device="cuda"))
                for _ in range(6)]
        for rep in range(3):
            ix.search_device(qa.data_ptr(), "f32", 256, 10, outs[2 * rep][0].data_ptr(), outs[2 * rep][1].data_ptr(), s1.cuda_stream)
            ix.search_device(qb.data_ptr(), "f32", 256, 10, outs[2 * rep + 1][0].data_ptr(), outs[2 * rep + 1][1].data_ptr(), s2.cuda_stream)
        ix.synchronize()
        torch.cuda.synchronize()
        for rep in range(3):
            assert np.array_equal(outs[2 * rep][1].cpu().numpy(), want_i)
            assert np.array_equal(outs[2 * rep + 1][1].cpu().numpy(), want_i[::-1])
            assert np.array_equal(outs[2 * rep][0].cpu().numpy(), want_s)


def test_upload_reads_rows_produced_on_the_default_stream(ts):
    """embed_into_index's pattern: rows are produced by kernels on torch's default stream (handle 0) and
    uploaded with stream = 0 = the index's own stream, which must be ordered behind them."""
    import torch
    n, d = 200_000, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    base = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    with ts.TheoremIndex(n, d, dtype="f32", metric="ip") as ix:
        for rep in range(3):
            rows = base
            for _ in range(20):                            # a queue of default-stream kernels the upload has to wait for
                rows = rows * 1.0001 + 0.001
            ix.upload_device(rows.data_ptr(), "f32", d, 0, n, torch.cuda.current_stream().cuda_stream)
            want = rows.cpu().numpy()
            ix.synchronize()
            got = ix.download()
            assert np.array_equal(got, want), f"repeat {rep}: the upload read its source before it was written"


def test_options_are_per_handle(ts):
    from theoremsearch_amd import _ffi
    q, c = oracle.inputs(70_000, 40, 768, 4, "ip")
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as a, \
            ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as b:
        a.set_option("TS_MFMA_STAT", 0)                     # the chain of guaranteed bounds: more levels
        _, _, sa = a.search(q, 10, algo="mfma", return_stats=True)
        _, _, sb = b.search(q, 10, algo="mfma", return_stats=True)
        assert sa["levels"] > sb["levels"] == 2
        a.set_option("TS_MFMA_STAT", None)
        _, _, sa = a.search(q, 10, algo="mfma", return_stats=True)
        assert sa["levels"] == 2
        with pytest.raises(_ffi.TSearchError):
            a.set_option("TS_NO_SUCH_KNOB", 1)


# ---- sharded search behind the C ABI --------------------------------------------------------------------------------
@pytest.mark.parametrize("ngpu", [1, 3, 8])
def test_shards_on_one_device_answer_like_the_whole_index(ts, ngpu):
    """ts_shards_*: the corpus row-sharded over `ngpu` shards (all on device 0 here: the exchange uses device copies,
    RCCL refuses two ranks on one GPU), routed uploads, packed per-shard results, merge: bit-identical to one index."""
    from theoremsearch_amd.distributed import Shards
    n, k = 90_001, 10
    q, c = oracle.inputs(n, 33, 768, 12, "ip")
    c[70_000] = c[5]                                        # an exact tie across shards: the lower global id ranks first
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as whole:
        want_s, want_i = whole.search(q, k)
    for threads in ("1", "0"):                              # one host thread per shard (default) / every enqueue from the caller's thread
        os.environ["TS_SHARDS_THREADS"] = threads
        try:
            with Shards(n, 768, ngpu, dtype="bf16", metric="ip", devices=[0] * ngpu) as sh:
                assert not sh.uses_rccl
                assert sh.bounds(0)[0] == 0 and sh.bounds(ngpu - 1)[1] == n
                for r0 in range(0, n, 25_000):              # uploads that straddle shard boundaries
                    sh.upload(c[r0:r0 + 25_000], r0)
                scores, idx = sh.search(q, k)
                for _ in range(3):                          # the workers are persistent: the same answer call after call
                    s2, i2 = sh.search(q, k)
                    assert np.array_equal(i2, idx) and np.array_equal(s2, scores)
        finally:
            del os.environ["TS_SHARDS_THREADS"]
        if threads == "1":
            first = (scores, idx)
        else:
            assert np.array_equal(idx, first[1]) and np.array_equal(scores, first[0])
    # same ids; scores bit-identical while the shards run the same kernel as the whole index (small shards fall under
    # the MFMA path's minimum size and take the scan, whose fp32 summation order differs in the last bit)
    assert np.array_equal(idx, want_i)
    assert np.array_equal(scores, want_s) if n // ngpu >= 16384 else np.allclose(scores, want_s, atol=1e-6)
    check(q, c, "ip", "bf16", k, scores, idx)


def test_comm_of_one_rank_runs_the_rccl_exchange(ts):
    """ts_comm_*: the RCCL communicator inside libtsearch (ncclCommInitRank, ncclAllGather) with world = 1 - all a one-GPU
    box can run - must reproduce the plain search through search -> all-gather -> merge."""
    from theoremsearch_amd import _ffi
    _ffi.prefer_torch_rccl()
    lib = _ffi.load()
    q, c = oracle.inputs(50_000, 17, 768, 13, "cos")
    ident = C.create_string_buffer(128)
    _ffi.check(lib.ts_comm_unique_id(ident, 128))
    comm = C.c_void_p()
    _ffi.check(lib.ts_comm_create(0, 1, 0, ident, 128, C.byref(comm)))
    try:
        w, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        _ffi.check(lib.ts_comm_info(comm, C.byref(w), C.byref(r), C.byref(d)))
        assert (w.value, r.value, d.value) == (1, 0, 0)
        with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="cos", row_offset=777) as ix:
            want_s, want_i = ix.search(q, 10)
            scores = np.empty((17, 10), np.float32)
            idx = np.empty((17, 10), np.int64)
            for _ in range(3):
                _ffi.check(lib.ts_comm_search(comm, ix.handle, _ffi.as_ptr(q), 0, 0, 17, 10, _ffi.as_ptr(scores), _ffi.as_ptr(idx), 0, None))
                assert np.array_equal(idx, want_i) and np.array_equal(scores, want_s)
            assert idx.min() >= 777
    finally:
        lib.ts_comm_destroy(comm)


def test_shards_of_the_production_table_shape_answer_like_the_whole_index(ts):
    """vector(1024) bf16 rows (rds_schema.sql:50-56) and a full launch of 256 queries: every shard runs the k-split paired pass
    over its own tile ranges; a score is the same two half sums wherever its row lives, so ids AND score bits are those of one
    index over all the rows."""
    from theoremsearch_amd.distributed import Shards
    n, k = 200_003, 10
    q, c = oracle.inputs(n, 256, 1024, 15, "ip")
    c[150_000] = c[7]                                       # an exact tie across shards
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as whole:
        want_s, want_i, st = whole.search(q, k, algo="mfma", return_stats=True)
        assert st["algo"] == 2 and st["fallback_queries"] == 0
    for shards in (2, 5):
        with Shards(n, 1024, shards, dtype="bf16", metric="ip", devices=[0] * shards) as sh:
            sh.upload(c, 0)
            scores, idx = sh.search(q, k)
            assert np.array_equal(idx, want_i) and np.array_equal(scores, want_s), shards


@pytest.mark.parametrize("pipeline", [1, 2])
def test_sharded_searcher_device_pipeline_with_changing_queries(ts, pipeline):
    """ShardedSearcher.search_device - the loop bench.py times - with DIFFERENT queries every step (identical queries
    would mask a result block that is read while it is rewritten): search on one stream (pipeline 2: on the streams of
    two handles in turn), exchange + merge on the side stream, double-buffered results."""
    import torch
    from theoremsearch_amd.distributed import ShardedSearcher
    n, nq, k = 150_000, 64, 10
    _, c = oracle.inputs(n, 1, 768, 21, "ip")
    rng = np.random.default_rng(5)
    batches = [rng.standard_normal((nq, 768)).astype(np.float32) for _ in range(6)]
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        want = [ix.search(b, k) for b in batches]
        searcher = ShardedSearcher(index=ix, pipeline=pipeline)
        dev = [torch.from_numpy(b).cuda() for b in batches]
        torch.cuda.synchronize()
        main = torch.cuda.Stream()
        got = []
        for step, qd in enumerate(dev):
            s, i, done = searcher.search_device(qd.data_ptr(), "f32", nq, k, stream=main)
            got.append((s, i, done))
            if step >= 1:                                   # results of the previous step are still valid (double-buffered)
                ps, pi, pdone = got[step - 1]
                pdone.synchronize()
                assert np.array_equal(pi.cpu().numpy(), want[step - 1][1]), f"step {step - 1}"
                assert np.array_equal(ps.cpu().numpy(), want[step - 1][0])
        got[-1][2].synchronize()
        assert np.array_equal(got[-1][1].cpu().numpy(), want[-1][1])
        searcher.close()


@pytest.mark.parametrize("pipeline", [1, 2])
def test_sharded_pipeline_reads_a_query_buffer_that_the_caller_rewrites_every_step(ts, pipeline):
    """ONE device buffer of queries in the index's own form (bf16, a whole launch's worth: read in place, no copy), rewritten
    on the caller's stream before every call - what an encoder replaying into a static output does (bench.py c5).  With two
    searches in flight the search runs on a lane stream of its own: the caller's stream must be ordered behind that read,
    or step i + 1's rewrite lands in the rows step i is still multiplying."""
    import torch
    from theoremsearch_amd.distributed import ShardedSearcher
    n, nq, k = 400_000, 64, 10
    _, c = oracle.inputs(n, 1, 768, 22, "ip")
    rng = np.random.default_rng(6)
    batches = [oracle.bf16_bits_to_f32(oracle.f32_to_bf16_bits(rng.standard_normal((nq, 768)).astype(np.float32))) for _ in range(8)]
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        want = [ix.search(b, k) for b in batches]
        searcher = ShardedSearcher(index=ix, pipeline=pipeline)
        dev = [torch.from_numpy(b).cuda().to(torch.bfloat16) for b in batches]
        qbuf = torch.empty((nq, 768), dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()
        main = torch.cuda.Stream()
        for lo in range(0, len(dev), 4):                    # four result blocks in rotation: four calls, then look
            got = []
            for qd in dev[lo:lo + 4]:
                with torch.cuda.stream(main):
                    qbuf.copy_(qd, non_blocking=True)       # the rewrite: ordered on the caller's stream only
                got.append(searcher.search_device(qbuf.data_ptr(), "bf16", nq, k, stream=main))
            for j, (s, i, done) in enumerate(got):
                done.synchronize()
                assert np.array_equal(i.cpu().numpy(), want[lo + j][1]), f"step {lo + j}"
                assert np.array_equal(s.cpu().numpy(), want[lo + j][0]), f"step {lo + j}"
        searcher.close()


def test_two_searches_in_flight_release_the_callers_stream_at_the_query_copy(ts):
    """pipeline = 2: a search runs on a lane stream of its own and reads a copy of the query batch that the lane took first, so
    the caller's stream is held until that COPY is done - not until the search is (which would serialise the lanes: search
    i + 1 could not start before search i had finished, and an encoder forward for batch i + 1 could not overlap search i).
    Timing events: one on the caller's stream right behind each call, one on the lane right behind its search; the caller's
    must come first, by most of the search's duration."""
    import torch
    from theoremsearch_amd.distributed import ShardedSearcher
    n, nq, k = 2_000_000, 256, 10
    _, c = oracle.inputs(n, 1, 768, 23, "ip")
    q = oracle.f32_to_bf16_bits(np.random.default_rng(7).standard_normal((nq, 768)).astype(np.float32))
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        want = ix.search(q, k)
        searcher = ShardedSearcher(index=ix, pipeline=2)
        qd = torch.from_numpy(q.view(np.int16)).cuda()
        main = torch.cuda.Stream()
        torch.cuda.synchronize()
        for _ in range(4):                                   # warm: lanes, views, buffers exist
            searcher.search_device(qd.data_ptr(), "bf16", nq, k, stream=main)
        torch.cuda.synchronize()
        after_call, after_search, outs = [], [], []
        for step in range(6):
            lane = searcher._lanes[searcher._step % 2][1]   # the lane this call will use
            outs.append(searcher.search_device(qd.data_ptr(), "bf16", nq, k, stream=main))
            e_main, e_lane = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e_main.record(main)
            e_lane.record(lane)
            after_call.append(e_main)
            after_search.append(e_lane)
        torch.cuda.synchronize()
        ahead = [after_call[i].elapsed_time(after_search[i]) for i in range(6)]     # ms the caller's stream is ahead of the search's end
        whole = after_search[0].elapsed_time(after_search[5]) / 5.0                  # ms per search, steady state
        assert min(ahead) > 0.5 * whole, f"the caller's stream waited for the search: ahead {ahead} ms, a search takes {whole:.3f} ms"
        for s_, i_, done in outs:
            done.synchronize()
            assert np.array_equal(i_.cpu().numpy(), want[1]) and np.array_equal(s_.cpu().numpy(), want[0])
        searcher.close()


def test_answers_do_not_move_with_the_tile_shares_of_the_full_pass(ts):
    """The full pass takes each workgroup's tile range from a table that the final select moves after every search
    (towards equal finishing times of the XCDs).  Same queries, ten searches in a row: the table moves, the answers may
    not; an append changes the number of tiles and the table starts again from equal shares."""
    q, c = oracle.inputs(400_000, 200, 768, 909, "ip")
    with ts.TheoremIndex(300_000, 768, dtype="bf16", metric="ip") as ix:
        ix.upload(c[:300_000], 0)
        s0, i0, st = ix.search(q, 10, algo="mfma", return_stats=True)
        assert st["fallback_queries"] == 0
        check(q, c[:300_000], "ip", "bf16", 10, s0, i0)
        for _ in range(9):
            s, i = ix.search(q, 10, algo="mfma")
            assert np.array_equal(i, i0) and np.array_equal(s, s0)
        ix.set_option("TS_MFMA_BALANCE", 0)                 # equal shares: the same answers
        s, i = ix.search(q, 10, algo="mfma")
        assert np.array_equal(i, i0) and np.array_equal(s, s0)
        ix.set_option("TS_MFMA_BALANCE", None)
        assert ix.append(c[300_000:]) == 300_000
        for _ in range(3):
            s1, i1 = ix.search(q, 10, algo="mfma")
            check(q, c, "ip", "bf16", 10, s1, i1)


# ---- round 3: queries read in place, the one-launch exact re-run, order of calls across streams ---------------------
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_device_queries_in_the_storage_form_are_read_in_place(ts, dtype):
    """Device queries that already are what the matrix kernels multiply (storage dtype, inner-product index, a whole
    launch's worth of rows) skip the preparation launch: same answers, bit for bit, as the prepared path - including a
    query whose candidates overflow and that the exact re-run (one launch: the last workgroup reduces the partial lists)
    has to serve from the caller's own matrix."""
    import torch
    rng = np.random.default_rng(31)
    n, d, k = 60_000, 768, 10
    c = rng.standard_normal((n, d), dtype=np.float32) * np.float32(0.05)
    hot = rng.standard_normal(d).astype(np.float32) * np.float32(0.05)
    rows = rng.choice(n, size=20_000, replace=False)
    c[rows] = hot                                            # query 0 = the hot row: 20,000 ties, candidate overflow
    for nq in (64, 128, 256):
        q = rng.standard_normal((nq, d), dtype=np.float32) * np.float32(0.05)
        q[0] = hot
        q_store = oracle.f32_to_bf16_bits(q) if dtype == "bf16" else q
        with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
            qh = oracle.bf16_bits_to_f32(q_store) if dtype == "bf16" else q
            want_s, want_i, st = ix.search(qh, k, algo="mfma", return_stats=True)      # host queries: prepared path
            assert st["fallback_queries"] >= 1
            qd = torch.from_numpy(q_store.view(np.int16) if dtype == "bf16" else q_store).cuda()
            out_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            out_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            for _ in range(2):                               # twice: the re-run's ticket counter must be left at zero
                out_i.fill_(-7)
                torch.cuda.synchronize()
                ix.search_device(qd.data_ptr(), dtype, nq, k, out_s.data_ptr(), out_i.data_ptr(), 0, algo="mfma")
                ix.synchronize()
                assert np.array_equal(out_i.cpu().numpy(), want_i) and np.array_equal(out_s.cpu().numpy(), want_s), (dtype, nq)
            assert want_i[0].tolist() == sorted(rows.tolist())[:k]
            check(qh, c, "ip", dtype, k, want_s, want_i)


def test_exact_rerun_with_large_k_and_many_queries(ts):
    """The one-launch re-run with four keys per lane (k > 64) and more failing queries than one scan pass serves."""
    rng = np.random.default_rng(32)
    n, d, k = 80_000, 768, 100
    c = rng.standard_normal((n, d), dtype=np.float32) * np.float32(0.05)
    hots = rng.standard_normal((6, d)).astype(np.float32) * np.float32(0.05)
    for j in range(6):
        c[j * 13000:(j + 1) * 13000 - 1000] = hots[j]        # six piles of 12,000 equal rows: six queries overflow
    q = np.concatenate([hots, rng.standard_normal((10, d), dtype=np.float32) * np.float32(0.05)])
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        s, i, st = ix.search(q, k, algo="mfma", return_stats=True)
        assert st["fallback_queries"] >= 6
        check(q, c, "ip", "bf16", k, s, i)
        s2, i2 = ix.search(q, k, algo="scan")
        assert np.array_equal(i[:6], i2[:6])


def test_a_callers_stream_may_be_destroyed_after_its_own_sync(ts):
    """the library must not touch a caller's stream after the call that was given it has returned - the next
    call (on another stream), ts_index_synchronize and ts_index_destroy only use the order event."""
    import torch
    q, c = oracle.inputs(50_000, 64, 768, 41, "ip")
    qd = torch.from_numpy(q).cuda()
    with ts.TheoremIndex.from_embeddings(c, dtype="bf16", metric="ip") as ix:
        want_s, want_i = ix.search(q, 10)
        out_s = torch.empty((64, 10), dtype=torch.float32, device="cuda")
        out_i = torch.empty((64, 10), dtype=torch.int64, device="cuda")
        for _ in range(3):
            st = torch.cuda.Stream()
            ix.search_device(qd.data_ptr(), "f32", 64, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
            st.synchronize()
            del st                                           # the handle may be recycled by the next Stream()
            s2, i2 = ix.search(q, 10)                        # own stream: ordered behind the event, not the dead stream
            assert np.array_equal(i2, want_i) and np.array_equal(out_i.cpu().numpy(), want_i)
        st = torch.cuda.Stream()
        ix.search_device(qd.data_ptr(), "f32", 64, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
        st.synchronize()
        del st
        ix.synchronize()
        assert np.array_equal(out_s.cpu().numpy(), want_s)


def test_the_product_library_refuses_timing_only_kernel_variants(ts):
    """VERDICT r2 / ADVICE: kernels that return wrong answers (TS_MFMA_VARIANT != 0) are not in libtsearch.so - only in the
    diagnostic build (make diag -> libtsearch_diag.so, selected with TS_LIB)."""
    import os
    from theoremsearch_amd import _ffi
    if os.path.basename(_ffi.lib_path()) != "libtsearch.so":
        pytest.skip("a diagnostic build is loaded (TS_LIB)")
    with ts.TheoremIndex(1000, 768, dtype="bf16", metric="ip") as ix:
        for v in (1, 2, 3, 5, 7):
            with pytest.raises(_ffi.TSearchError) as e:
                ix.set_option("TS_MFMA_VARIANT", v)
            assert e.value.code == -5
        ix.set_option("TS_MFMA_VARIANT", 0)
        ix.set_option("TS_MFMA_VARIANT", None)
        # round 4: the knobs that only the A/B tools ever turned moved to the diagnostic build with them
        for name in ("TS_MFMA_NO_IDLE", "TS_MFMA_MIN_RANK", "TS_MFMA_GROUPS", "TS_MFMA_STAT_CANDS", "TS_MFMA_TARGET_CANDS",
                     "TS_MFMA_TARGET_SPARSE", "TS_MFMA_MIN_ROWS", "TS_SCAN_GENERIC", "TS_SCAN_MAX_QUERIES", "TS_PROBE_SPREAD"):
            with pytest.raises(_ffi.TSearchError) as e:
                ix.set_option(name, 1)
            assert e.value.code == -5, name


# ---- citation-weighted ranking on the device ------------------------------------------------
@pytest.mark.parametrize("dtype,d", [("f32", 1024), ("bf16", 768), ("f32", 200)])
def test_biased_search_is_the_citation_weighted_ranking_over_all_rows(ts, dtype, d):
    """ts_search_biased against the oracle's restatement of streamlit_app.py:348-364 with the pool widened to the whole
    corpus: weighted = similarity + w * ln(citations) (0 for NULL / <= 0), ORDER BY weighted DESC, similarity DESC.
    Citation counts: None, 0, 1, small, and a few enormous ones that lift far-away rows into the answer (rows the
    reference's max(50, 10 k) pool never sees)."""
    from theoremsearch_amd import pgvector
    n, nq, k, w = 20_000, 6, 10, 0.02
    q, c = oracle.inputs(n, nq, d, 300 + d, "ip")
    rng = np.random.default_rng(9)
    cites = [None if u < 0.1 else (0 if u < 0.2 else int(v)) for u, v in zip(rng.random(n), rng.integers(1, 400, n))]
    for r in rng.choice(n, 12, replace=False):
        cites[int(r)] = int(10 ** rng.integers(6, 9))          # w * ln(1e8) = 0.37: ten standard deviations of the scores
    bias = pgvector.citation_bias(cites)
    qp, cp = oracle.prepared_inputs(q, c, "ip", dtype)
    sim64 = oracle.scores_fp64(qp, cp)                           # fp64 similarities of what the index multiplies
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric="ip") as ix:
        ws, sims, idx = ix.search_biased(q, k, bias, w)
        assert idx.min() >= 0
        for b in range(nq):
            weighted64 = sim64[b] + w * bias.astype(np.float64)
            want_i, want_sim, want_w = oracle.citation_weighted_rerank(np.arange(n), sim64[b], cites, w, k)
            got_w = weighted64[idx[b]]
            # pinned ranks (fp64 gap > 1e-6 on both sides) must match exactly, the rest as sets; scores within 1e-5
            gaps = want_w[:-1] - want_w[1:]
            for r in range(k):
                lo = gaps[r - 1] if r else np.inf
                hi = gaps[r] if r < k - 1 else np.inf
                if lo > 1e-6 and hi > 1e-6 and r < k - 1:
                    assert idx[b, r] == want_i[r], (dtype, b, r)
            assert np.all(got_w >= want_w[-1] - 1e-6)
            assert np.allclose(ws[b], want_w, atol=1e-5) and np.allclose(sims[b], sim64[b][idx[b]], atol=1e-5)
        assert np.intersect1d(idx[0], np.flatnonzero(bias > 10)).size > 0       # the heavily cited rows made it
        # w = 0 is the plain search
        z_s, z_sim, z_i = ix.search_biased(q, k, bias, 0.0)
        p_s, p_i = ix.search(q, k, algo="scan")
        assert np.array_equal(z_i, p_i) and np.array_equal(z_s, p_s) and np.array_equal(z_sim, p_s)
        # with a WHERE-clause mask: the k best allowed rows by weighted score
        mask = rng.random(n) < 0.4
        m_s, m_sim, m_i = ix.search_biased(q, k, bias, w, mask=mask)
        assert mask[m_i].all()
        keep = np.flatnonzero(mask)
        for b in range(nq):
            weighted64 = sim64[b] + w * bias.astype(np.float64)
            best = np.sort(weighted64[keep])[::-1][:k]
            assert np.allclose(m_s[b], best, atol=1e-5)


def test_pgvector_adapter_exact_form_equals_the_pool_form_when_the_pool_holds_the_answer(ts):
    """pgvector.search(..., exact=True) (the kernel over all rows) returns what the reference's pool form returns whenever
    the max(50, 10 k) nearest rows contain the weighted top-k - small citation counts - and differs, correctly, when a
    heavily cited theorem lies outside the pool."""
    from theoremsearch_amd import pgvector
    n, d, k = 8_000, 768, 5
    q, c = oracle.inputs(n, 1, d, 77, "ip")
    rng = np.random.default_rng(3)
    cites = [int(v) for v in rng.integers(1, 4, n)]              # ln <= 1.1: with w = 0.001 a nudge among near neighbours
    with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="ip") as ix:
        pool = pgvector.search(ix, q[0], k, citation_weight=0.001, citations=cites)
        exact = pgvector.search(ix, q[0], k, citation_weight=0.001, citations=cites, exact=True)
        assert [r["row"] for r in exact] == [r["row"] for r in pool]
        assert np.allclose([r["score"] for r in exact], [r["score"] for r in pool], atol=1e-5)
        far = int(np.argsort(oracle.scores_fp64(*oracle.prepared_inputs(q, c, "ip", "f32"))[0])[n // 2])   # a median row
        cites[far] = 10 ** 9
        pool = pgvector.search(ix, q[0], k, citation_weight=0.05, citations=cites)
        exact = pgvector.search(ix, q[0], k, citation_weight=0.05, citations=cites, exact=True)
        assert exact[0]["row"] == far and far not in [r["row"] for r in pool]



def test_biased_search_argument_checks(ts):
    import ctypes as C
    from theoremsearch_amd import _ffi
    lib = _ffi.load()
    q, c = oracle.inputs(3000, 2, 768, 5, "ip")
    with ts.TheoremIndex.from_embeddings(c, metric="ip") as ix:
        bias = np.zeros(3000, np.float32)
        with pytest.raises(ValueError):
            ix.search_biased(q, 5, bias[:10], 0.1)                       # one value per row
        s = np.empty((2, 5), np.float32)
        i = np.empty((2, 5), np.int64)
        args = (ix.handle, _ffi.as_ptr(q), 0, 0, 2, 5)
        assert lib.ts_search_biased(*args, None, 0, 0.1, None, 0, _ffi.as_ptr(s), None, _ffi.as_ptr(i), 0, None) == -1   # bias NULL
        assert lib.ts_search_biased(*args, _ffi.as_ptr(bias), 0, float("nan"), None, 0, _ffi.as_ptr(s), None, _ffi.as_ptr(i), 0, None) == -1
        assert lib.ts_search_biased(*args, _ffi.as_ptr(bias), 0, 0.1, None, 0, _ffi.as_ptr(s), None, _ffi.as_ptr(i), 0, None) == 0   # out_sims optional
        sub = ix.subset(np.arange(0, 3000, 3))
        with pytest.raises(_ffi.TSearchError) as e:
            sub.search_biased(q, 5, bias[:1000], 0.1)
        assert e.value.code == -5
        sub.close()
        # k results fewer than k rows allowed: padded like every search
        mask = np.zeros(3000, bool)
        mask[[5, 17]] = True
        ws, sims, idx = ix.search_biased(q, 5, bias, 0.1, mask=mask)
        assert set(idx[0, :2].tolist()) == {5, 17} and (idx[:, 2:] == -1).all() and np.isneginf(ws[:, 2:]).all() and np.isneginf(sims[:, 2:]).all()
