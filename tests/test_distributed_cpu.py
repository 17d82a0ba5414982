"""The N > 1 path on CPU: world_size-2 gloo ranks exercise partition + exchange + merge of
ShardedSearcher.  The per-shard search and the merge are injected (numpy oracle) because the HIP
kernels need a GPU; what is tested is that the distributed plumbing returns the whole-corpus answer."""
import datetime
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _numpy_merge(scores, idx, k):
    nparts, nq, _ = scores.shape
    out_s = np.full((nq, k), -np.inf, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    for b in range(nq):
        s, i = scores[:, b, :].reshape(-1), idx[:, b, :].reshape(-1)
        keep = i >= 0
        order = np.lexsort((i[keep], -s[keep]))[:k]
        out_s[b, : order.size] = s[keep][order]
        out_i[b, : order.size] = i[keep][order]
    return out_s, out_i


def _worker(rank, world, port, n, nq, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from theoremsearch_amd.distributed import ShardedSearcher, shard_bounds
        q, c = oracle.golden_inputs(n, nq, 64, 123, "ip")
        lo, hi = shard_bounds(n, world, rank)

        def local_search(queries, kk):
            s, i = oracle.search(queries, c[lo:hi], kk, "ip", "f32")
            return s, np.where(i >= 0, i + lo, -1)

        searcher = ShardedSearcher(local_search, merge=_numpy_merge)
        scores, idx = searcher.search(q, k)
        want_s, want_i = oracle.search(q, c, k, "ip", "f32")
        ok = np.array_equal(idx, want_i) and np.allclose(scores, want_s, atol=1e-6)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,nq,k", [(1001, 5, 10), (7, 3, 5)])
def test_sharded_search_two_gloo_ranks(n, nq, k):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, nq, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _rank_worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from theoremsearch_amd.distributed import ShardedSearcher, shard_bounds
        nq = 7
        q, c = oracle.golden_inputs(n, nq, 64, 321, "ip")
        c[5] = c[n - 2]                     # an exact tie across the two shards: the lower global id ranks first
        lo, hi = shard_bounds(n, world, rank)
        truth = oracle.scores_fp64(q, c)
        s32 = truth.astype(np.float32)      # the "kernel" scores of this stand-in

        def local_rank_of(queries, rows):
            ranks = np.full(len(rows), -1, np.int64)
            scores = np.full(len(rows), np.nan, np.float32)
            for i, r in enumerate(rows):
                if lo <= r < hi:
                    ranks[i] = oracle.rank_of(s32[i:i + 1, lo:hi], [r - lo])[0]
                    scores[i] = s32[i, r]
            return ranks, scores

        def local_count_above(queries, sc, ids):
            # queries arrive filtered: recover which ones by matching rows of q
            out = np.empty(len(ids), np.int64)
            for j, (qq, t, gid) in enumerate(zip(queries, sc, ids)):
                i = int(np.flatnonzero((q == qq).all(axis=1))[0])
                loc = s32[i, lo:hi]
                gids = np.arange(lo, hi)
                out[j] = int(np.sum((loc > t) | ((loc == t) & (gids < gid))))
            return out

        searcher = ShardedSearcher(lambda *_: None)
        rows = np.array([0, n - 1, 5, n - 2, n // 2, n // 2 - 1, n + 3])
        got = searcher.rank_of(q, rows, local_rank_of=local_rank_of, local_count_above=local_count_above)
        want = oracle.rank_of(s32, rows)
        ret[rank] = bool(np.array_equal(got, want)) and got[-1] == -1
    finally:
        dist.destroy_process_group()


def test_sharded_rank_of_two_gloo_ranks():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank_worker, args=(world, _free_port(), 501, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_bench_launches_eight_ranks_and_runs_its_host_collectives():
    """`python bench.py --gpus 8 --rehearse-launch`, bare, as the driver starts the 8-GPU run: the parent starts EIGHT rank
    processes (file rendezvous, one relay thread each, OMP_NUM_THREADS = cores / 8), the ranks meet over gloo and run the
    host collectives of a sharded run - barriers, the max-over-ranks reductions, all_gather_object of truth tables of the
    real shape, one all-gather of the packed per-shard top-k block - and rank 0's single JSON line comes back through the
    parent.  No device and no search (a one-GPU box may hold six processes on its card; tests/test_mirrors_gpu.py rehearses four
    ranks through the kernels).  Then a rank that fails: non-zero exit code, no line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS", "TS_BENCH_INIT_FILE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--rehearse-launch"], capture_output=True,
                         text=True, timeout=280, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc["n_ranks"] == 8 and doc["backend"] == "gloo" and doc["launched_by"] == "bench.py" and doc["rendezvous"] == "file"
    assert doc["hsa_enable_ipc_mode_legacy"] == "0"              # set for the ranks even from an environment without it
    assert int(doc["omp_num_threads"]) == max(1, len(os.sched_getaffinity(0)) // 8)
    assert doc["packed_block_bytes"] == 256 * 10 * 12 and doc["truth_table_bytes_per_rank"] > 0
    # the real run without a device: every rank refuses, the parent relays no line and a non-zero exit code
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "1000", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=280, cwd=root, env=env)
    if not __import__("conftest").gpu_available():
        assert bad.returncode != 0 and not bad.stdout.strip(), (bad.returncode, bad.stdout)


def test_the_exchange_placement_rule_does_not_decide_on_noise():
    """bench.decide_placement: the first round of each mode is discarded (RCCL's lazy set-up: round 4's trial saw 1.2 ms then
    5-11 ms in consecutive rounds of the same mode), medians are compared, and `inline` must win by more than 2 % to displace
    `overlap`."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    use, mo, mi = bench.decide_placement([9.0, 0.450, 0.452, 0.449], [0.30, 0.455, 0.451, 0.460])
    assert use and abs(mo - 0.450) < 1e-9 and abs(mi - 0.455) < 1e-9          # a lucky first inline round does not count
    assert bench.decide_placement([0.5, 0.450, 0.451, 0.449], [0.5, 0.445, 0.446, 0.444])[0]       # 1.1 % better: not enough
    assert not bench.decide_placement([0.5, 0.450, 0.451, 0.449], [0.5, 0.430, 0.431, 0.429])[0]   # 4.4 % better: inline
    assert bench.decide_placement([0.5, 0.450, 5.0, 0.449], [0.5, 0.452, 0.451, 0.453])[0]         # one slow overlap round: the median holds
    assert bench.decide_placement([1.23, 5.15], [1.20, 11.12])[0]                                    # round 4's two noisy rounds: overlap stays
