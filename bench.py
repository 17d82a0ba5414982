#!/usr/bin/env python3
"""bench.py - queries/sec of brute-force top-10 over an [N x 768] theorem-embedding matrix.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W          (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Workload (BASELINE.json): configs[2] "10M x 768 bf16 corpus, batch-256 queries" - the shape the
north-star target is quoted on; it fits one GPU (15.36 GB).  One step = one batch of 256 queries
searched against the whole corpus (inputs resident in HBM).  With N > 1 the SAME corpus is
row-sharded over the ranks (strong scaling): every rank searches its shard, the per-shard top-10
are exchanged with one RCCL all-gather, and every rank merges them.
Other workloads: --workload c2 (1M x 768 fp32, batch 1), --workload c4 (50M x 768 bf16, batch 256).

Synthetic data: corpus chunk c (250,000 rows) = default_rng([1234, c]).standard_normal, rows
L2-normalised in fp32, rounded to bf16 for the bf16 workloads (synthetic.synth_chunk); queries from
default_rng([5678, 0]) the same way.  Rows are stored as given (metric = inner product on
normalised rows = cosine).
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (rows, dtype, nq)
    "c2": (1_000_000, "f32", 1),
    "c3": (10_000_000, "bf16", 256),
    "c4": (50_000_000, "bf16", 256),
    "c5": (10_000_000, "bf16", 256),   # encoder-in-loop: sentence-encoder forward feeds the C3 index
}
D = 768   # overridden by --dim (diagnostic: the production table of the reference is vector(1024), rds_schema.sql:50)
K = 10
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=0, help="override the corpus size (debug)")
    ap.add_argument("--nq", type=int, default=0, help="override the query batch (debug)")
    ap.add_argument("--algo", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--dim", type=int, default=768, help="embedding dimension (diagnostic; BASELINE configs use 768)")
    ap.add_argument("--seq-len", type=int, default=32, help="c5: tokens per synthetic query")
    ap.add_argument("--force-dist", action="store_true", help="debug: run the exchange + merge path even with one rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank uses device 0, exchange over gloo "
                         "through host memory (timings meaningless)")
    ap.add_argument("--mask-frac", type=float, default=0.0,
                    help="diagnostic: filtered search with this fraction of rows allowed (device bitmask)")
    ap.add_argument("--zero-queries", action="store_true", help="diagnostic: all-zero queries (power probe)")
    ap.add_argument("--zero-corpus", action="store_true", help="diagnostic: all-zero corpus (power probe)")
    args = ap.parse_args()
    global D
    D = args.dim
    # Exactly ONE line goes to stdout (the JSON): libraries print banners there (RCCL prints its version / host /
    # library path on communicator creation), so fd 1 points at stderr until the result line is written.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import synthetic                # input generator (plain numpy)
    from oracle import oracle       # checker legs only: recall@10 and cpu_baseline
    import theoremsearch_amd as ts
    from theoremsearch_amd import _ffi

    if _ffi.device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device (libtsearch has no CPU path)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rows_total, dtype, nq = WORKLOADS[args.workload]
    if args.rows:
        rows_total = args.rows
    if args.nq:
        nq = args.nq
    bf16 = dtype == "bf16"
    elem = 2 if bf16 else 4

    # ---- this rank's shard: contiguous rows [lo, hi), generated chunk by chunk -------------------
    lo = rows_total * rank // world
    hi = rows_total * (rank + 1) // world
    n_local = hi - lo
    CH = synthetic.CHUNK_ROWS
    chunks = list(range(lo // CH, (hi + CH - 1) // CH))
    ix = ts.TheoremIndex(n_local, D, dtype=dtype, metric="ip", device=local_rank, row_offset=lo)
    ncpu = len(os.sched_getaffinity(0))
    nthreads = args.gen_threads or max(1, min(16, ncpu // max(1, min(world, 8))))
    t_gen = time.time()
    cache = {}                                   # host copies for the recall check / cpu baseline
    cache_ok = n_local * D * elem <= (40 << 30) and not args.no_recall

    def make(c):
        data = synthetic.synth_chunk(c, CH, D, bf16=bf16) if not args.zero_corpus else np.zeros((CH, D), np.uint16 if bf16 else np.float32)
        a, b = max(lo, c * CH), min(hi, (c + 1) * CH)
        ix.upload(data[a - c * CH:b - c * CH], a - lo)
        if cache_ok or (c == 0 and rank == 0):
            cache[c] = data
        return c

    with ThreadPoolExecutor(nthreads) as ex:
        for i, c in enumerate(ex.map(make, chunks)):
            if (i + 1) % 8 == 0:
                log(rank, f"generated+uploaded {i + 1}/{len(chunks)} chunks ({time.time() - t_gen:.0f}s)")
    log(rank, f"corpus ready: {n_local} rows/rank x {D} {dtype} in {time.time() - t_gen:.1f}s ({nthreads} threads)")

    q_host = synthetic.synth_queries(0, nq, D, bf16=bf16)      # uint16 bits or float32
    if args.zero_queries:
        q_host = np.zeros_like(q_host)
    main = torch.cuda.current_stream()
    q_dev = torch.from_numpy(q_host.view(np.int16) if bf16 else q_host).cuda()
    # one result block per step parity: scores [nq x K] f32 at offset 0, ids [nq x K] i64 at `idx_off`
    idx_off = (nq * K * 4 + 7) // 8 * 8
    blk = idx_off + nq * K * 8
    res = [torch.empty(blk, dtype=torch.uint8, device="cuda") for _ in range(2)]
    use_dist = world > 1 or args.force_dist
    lib = _ffi.load()
    import ctypes as C
    if use_dist:
        # the exchange + merge of step i run on a side stream and overlap the search of step i+1
        side = torch.cuda.Stream()
        gat = [torch.empty(world * blk, dtype=torch.uint8, device="cuda") for _ in range(2)]
        fin_s = [torch.empty((nq, K), dtype=torch.float32, device="cuda") for _ in range(2)]
        fin_i = [torch.empty((nq, K), dtype=torch.int64, device="cuda") for _ in range(2)]
        ev_search = [torch.cuda.Event() for _ in range(2)]
        ev_done = [torch.cuda.Event() for _ in range(2)]
    encoder = None
    if args.workload == "c5":
        # BASELINE.json configs[4]: encoder forward (PyTorch-ROCm, random-init BERT-base-shaped stand-in: no weights
        # offline) on synthetic token sequences, pooled + normalised on the device, handed to the search by pointer
        from theoremsearch_amd.encoder import SentenceEncoder
        encoder = SentenceEncoder()
        g = torch.Generator(device="cpu").manual_seed(5678)
        tok_ids = torch.randint(1000, 30000, (nq, args.seq_len), generator=g).cuda()
        tok_ids[:, 0], tok_ids[:, -1] = 101, 102
        tok_mask = torch.ones_like(tok_ids)
    step_no = [0]
    mask_ptr, mask_host = 0, None
    if args.mask_frac > 0:
        mask_host = np.random.default_rng(99 + rank).random(n_local) < args.mask_frac
        bits = np.packbits(mask_host, bitorder="little")
        words = np.zeros((n_local + 31) // 32 * 4, dtype=np.uint8)
        words[: bits.shape[0]] = bits
        mask_dev = torch.from_numpy(words).cuda()
        mask_ptr = mask_dev.data_ptr()

    def encode_queries():
        with torch.inference_mode():
            hidden = encoder.model(input_ids=tok_ids, attention_mask=tok_mask).last_hidden_state
            return encoder.pool(hidden, tok_mask, True)       # fused mean-pool + L2 normalise (ts_pool_normalize)

    def step():
        i = step_no[0]
        step_no[0] += 1
        b = i & 1
        if use_dist and i >= 2:
            main.wait_event(ev_done[b])        # step i-2's exchange has consumed res[b]
        base = res[b].data_ptr()
        if encoder is not None:
            emb = encode_queries()
            ix.search_device(emb.data_ptr(), "f32", nq, K, base, base + idx_off, main.cuda_stream, algo=args.algo)
        else:
            ix.search_device(q_dev.data_ptr(), dtype, nq, K, base, base + idx_off, main.cuda_stream, algo=args.algo,
                             mask_ptr=mask_ptr)
        if use_dist:
            ev_search[b].record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_search[b])
                if args.share_gpu:                                   # rehearsal: gloo moves host memory
                    side.synchronize()
                    host = torch.empty(gat[b].shape, dtype=gat[b].dtype)
                    dist.all_gather_into_tensor(host, res[b].cpu())
                    gat[b].copy_(host)
                else:
                    dist.all_gather_into_tensor(gat[b], res[b])      # ONE collective: 12 * nq * K bytes per rank
                _ffi.check(lib.ts_merge_topk_packed(local_rank, C.c_void_p(gat[b].data_ptr()), blk, idx_off, world, nq,
                                                    K, K, C.c_void_p(fin_s[b].data_ptr()), C.c_void_p(fin_i[b].data_ptr()),
                                                    C.c_void_p(side.cuda_stream)))
                ev_done[b].record(side)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ix.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = ix.profile_read()
    ix.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    qps = nq * args.steps / dt

    # ---- roofline of the dominant kernel (live hipEvent brackets inside the library) -------------
    kern_ms = prof["total_ms"] / max(1, prof["launches"])
    launches_per_step = prof["launches"] / max(1, args.steps)
    alg_bytes = prof["rows_per_launch"] * D * elem           # corpus rows read once per launch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from rocprofv3 --pmc (offline pass)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.workload}_n{world}") if (not args.nq and not args.rows) else None
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "mfma_topk_kernel" if (bf16 and nq > 4 and args.algo != "scan" and not mask_ptr) else "scan_kernel",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel_ms": round(kern_ms, 4), "launches_per_step": launches_per_step,
                "algorithmic_bytes_per_launch": alg_bytes}

    # ---- how the estimated thresholds did (one extra search outside the timed region) ---------------
    search_stats = None
    if bf16 and nq > 4 and args.algo in ("auto", "mfma") and not mask_ptr and encoder is None:
        _, _, st_ = ix.search(q_host, K, algo=args.algo, return_stats=True)
        search_stats = {"levels": st_["levels"], "fallback_queries": st_["fallback_queries"],
                        "candidates_per_query": round(st_["candidates"] / float(nq), 1)}
        log(rank, f"threshold levels {st_['levels']}, candidates per query {search_stats['candidates_per_query']}, "
                  f"queries re-run exactly {st_['fallback_queries']}")

    # ---- recall@10 against the oracle (fp64 scores of the same bf16/fp32 values) -------------------
    last = (step_no[0] - 1) & 1
    if use_dist:
        res_s, res_i = fin_s[last].cpu().numpy(), fin_i[last].cpu().numpy()
    else:
        raw = res[last].cpu().numpy()
        res_s = raw[: nq * K * 4].view(np.float32).reshape(nq, K)
        res_i = raw[idx_off: idx_off + nq * K * 8].view(np.int64).reshape(nq, K)
    recall = None
    if not args.no_recall and encoder is not None:
        q_host = oracle.f32_to_bf16_bits(encode_queries().cpu().numpy())     # what the index multiplies: bf16-rounded
    if not args.no_recall:
        nchk = min(nq, 8)
        qf = oracle.bf16_bits_to_f32(q_host[:nchk]) if bf16 else q_host[:nchk]
        best_s = np.full((nchk, K), -np.inf)
        t_chk = time.time()

        def local_truth(c):
            data = cache[c] if c in cache else synthetic.synth_chunk(c, CH, D, bf16=bf16)
            a, b = max(lo, c * CH), min(hi, (c + 1) * CH)
            blk = data[a - c * CH:b - c * CH]
            blk = oracle.bf16_bits_to_f32(blk) if bf16 else blk
            s = qf.astype(np.float64) @ blk.astype(np.float64).T
            if mask_host is not None:
                s[:, ~mask_host[a - lo:b - lo]] = -np.inf
            top = -np.sort(-s, axis=1)[:, :K] if s.shape[1] > K else s
            # fp64 scores of the rows the GPU returned that live in this chunk
            got = {}
            for b_ in range(nchk):
                for j in res_i[b_]:
                    if a <= j < b:
                        got[(b_, int(j))] = float(s[b_, j - a])
            return top, got

        got_scores = {}
        with ThreadPoolExecutor(max(1, nthreads // 2)) as ex:
            for top, got in ex.map(local_truth, chunks):
                best_s = -np.sort(-np.concatenate([best_s, top], axis=1), axis=1)[:, :K]
                got_scores.update(got)
        if world > 1:
            objs = [None] * world
            dist.all_gather_object(objs, (best_s, got_scores))
            best_s = -np.sort(-np.concatenate([o[0] for o in objs], axis=1), axis=1)[:, :K]
            got_scores = {k_: v for o in objs for k_, v in o[1].items()}
        hits = 0
        for b_ in range(nchk):
            kth = best_s[b_, K - 1]
            for j in res_i[b_]:
                if got_scores.get((b_, int(j)), -np.inf) >= kth - 1e-6:
                    hits += 1
        recall = hits / float(nchk * K)
        log(rank, f"recall@10 over {nchk} queries vs fp64 oracle: {recall:.4f} ({time.time() - t_chk:.0f}s)")

    # ---- CPU baseline: the reference's formulation on the host cores, bounded sample ---------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nch = [c for c in sorted(cache) if c < 4]                     # up to 1M rows: ~10 s of host work at batch 256
        rows_s = np.concatenate([cache[c] for c in nch], axis=0)[: min(len(nch) * CH, rows_total)] if nch else cache[0]
        c_s = oracle.bf16_bits_to_f32(rows_s) if bf16 else rows_s
        q_s = oracle.bf16_bits_to_f32(q_host) if bf16 else q_host
        # the oracle's port of the reference formulation: util.cos_sim + np.argsort(-S)[:, :10], all host cores
        top, t_cpu = oracle.cpu_reference_topk(q_s, c_s, K, threads=ncpu)
        scale = rows_total / float(c_s.shape[0])
        cpu = {"value": round(nq / (t_cpu * scale), 3), "unit": "queries/s", "cores": ncpu, "kind": "port",
               "sample": f"{c_s.shape[0]} of {rows_total} rows x {nq} queries, fp32 torch-CPU cos_sim + np.argsort "
                         f"in {t_cpu:.2f}s; value scaled linearly to the full corpus (x{1 / scale:.4f})",
               "measured_qps_on_sample": round(nq / t_cpu, 2)}
        del top

    if rank == 0:
        line = {
            "metric": "queries/sec, brute-force top-10 over N x 768 theorem embeddings",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{rows_total}x{D} {dtype} corpus, batch-{nq} queries, top-{K} "
                                   f"(BASELINE.json configs[{ {'c2': 1, 'c3': 2, 'c4': 3, 'c5': 4}[args.workload] }])" + (f", encoder forward in the loop ({args.seq_len} tokens/query, random-init BERT-base shape)" if encoder is not None else ""),
                       "rows": rows_total, "dim": D, "batch": nq, "k": K,
                       **({"mask_frac": args.mask_frac} if args.mask_frac > 0 else {}),
                       "parallelism": f"corpus row-sharded x{world}" + ((", gloo rehearsal on one GPU" if args.share_gpu else ", RCCL all-gather of per-shard top-k") if use_dist else "")},
            "recall_at_10": recall,
            "search_stats": search_stats,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
