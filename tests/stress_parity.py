"""Randomised parity sweep on the GPU box: many (n, d, dtype, metric, nq, k, algo, mask) combinations against the
oracle's fp64 truth, for a fixed time budget.  Prints a line per case (so a stall is visible) and arms
faulthandler so that a hung call dumps the Python stack and exits.

    python tests/stress_parity.py --seconds 240 --seed 1 [--big | --pairs]
"""
import argparse
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import theoremsearch_amd as ts  # noqa: E402
from oracle import oracle  # noqa: E402  (checker)


def one_case(rng, big: bool, case: int = 0, watchdog: bool = True, pairs: bool = False) -> str:
    """One random combination, checked; returns its description.  Used by the time-boxed sweep below and, with a
    fixed seed list, by tests/test_fullsize_gpu.py (collected by pytest)."""
    d = int(rng.choice([768, 768, 768, 1024, 1024, 384, 384, 512, 512, 40]))
    dtype = str(rng.choice(["bf16", "bf16", "f32"]))
    metric = str(rng.choice(["cos", "ip"]))
    n = int(rng.choice([1, 33, 1000, 16384, 16385, 40000, 70001, 150000, 300000]))
    if d == 1024:
        n = min(n, 150000)
    nq = int(rng.choice([1, 2, 4, 5, 31, 32, 33, 64, 100, 128, 129, 160, 200, 256, 257, 300]))
    k = int(rng.choice([1, 5, 10, 10, 10, 50, 64, 65, 200, 256]))
    if big:
        d, dtype = 768, "bf16"
        n = int(rng.choice([1_000_000, 1_700_001, 2_500_000]))
        nq = int(rng.choice([5, 16, 33, 130]))
        k = int(rng.choice([1, 10, 10, 50, 256]))
    if pairs:
        # the paired pass of the production table's shape (bf16 x 1024, 193+ queries in a launch): tile ranges of every
        # parity and length over the pairs of the default grid and of smaller ones, ragged last tiles, ties, masks
        d, dtype = 1024, "bf16"
        n = int(rng.choice([16384, 16385, 20_000 + int(rng.integers(0, 4096)), 65_536 + int(rng.integers(0, 64)),
                            int(rng.integers(100_000, 420_000))]))
        nq = int(rng.choice([193, 200, 255, 256, 256, 256, 257, 300, 449, 512]))
        k = int(rng.choice([1, 5, 10, 10, 64, 65, 200]))
    mfma_ok = d in (384, 512, 768, 1024)               # bf16, and fp32 on the exact-fp32 matrix instructions
    algo = str(rng.choice(["auto", "scan", "mfma"])) if mfma_ok else str(rng.choice(["auto", "scan"]))
    if big:
        algo = "auto" if rng.random() < 0.4 else "mfma"     # "auto" may carry a dense host mask (masked MFMA pass)
    if pairs:
        algo = str(rng.choice(["auto", "mfma"]))
    if algo == "scan" and nq > 64:
        nq = int(rng.choice([1, 4, 7, 33]))          # the scan serves 4 queries per pass: keep the sweep moving
    use_mask = algo != "mfma" and rng.random() < (0.7 if big else 0.25) and (nq <= 8 or (mfma_ok and n >= 16384))
    grid = int(rng.choice([0, 0, 16, 64, 208, 240])) if pairs else 0      # 0: the device's CU count
    seed = int(rng.integers(0, 2**31))
    if watchdog:
        faulthandler.dump_traceback_later(300 if big else 120, exit=True)
    t0 = time.time()
    q, c = oracle.inputs(n, nq, d, seed, metric)
    if rng.random() < 0.15 and n > 100:              # duplicates: exact ties
        c[rng.integers(0, n, 20)] = c[0]
    if big and rng.random() < 0.5:              # a cluster near the queries: heavy upper tail
        u = q.mean(axis=0)
        members = rng.choice(n, n // 25, replace=False)
        c[members] += (rng.random(members.size).astype(np.float32) * np.float32(4.0))[:, None] * u
    mask = (rng.random(n) < float(rng.choice([0.05, 0.3, 0.8]))) if use_mask else None
    # round 3: now and then the citation-weighted ranking (scan kernel + one fp32 side array), or device queries in the
    # storage form (read in place by the matrix kernels when they fill a launch; prepared otherwise)
    mode = "plain"
    if not big and algo != "mfma" and nq <= 8 and rng.random() < 0.3:
        mode = "biased"
    elif mask is None and metric == "ip" and rng.random() < 0.3:
        mode = "device"
    bias, w = None, 0.0
    with ts.TheoremIndex.from_embeddings(c, dtype=dtype, metric=metric) as ix:
        if grid:
            ix.set_option("TS_MFMA_GRID", grid)
        if mode == "biased":
            bias = np.log(rng.integers(1, 5000, n).astype(np.float64)).astype(np.float32) * (rng.random(n) < 0.8)
            w = float(rng.choice([0.0, 0.001, 0.02]))
            scores, _, idx = ix.search_biased(q, k, bias, w, mask=mask)
        elif mode == "device":
            import torch
            qs = oracle.f32_to_bf16_bits(q) if dtype == "bf16" else q
            qd = torch.from_numpy(qs.view(np.int16) if dtype == "bf16" else qs).cuda()
            o_s = torch.empty((nq, k), dtype=torch.float32, device="cuda")
            o_i = torch.empty((nq, k), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            ix.search_device(qd.data_ptr(), dtype, nq, k, o_s.data_ptr(), o_i.data_ptr(), 0, algo=algo)
            ix.synchronize()
            scores, idx = o_s.cpu().numpy(), o_i.cpu().numpy()
        elif mask is not None:
            scores, idx = ix.search(q, k, mask=mask)
        else:
            scores, idx = ix.search(q, k, algo=algo)
    rows = np.flatnonzero(mask) if mask is not None else np.arange(n)
    qp, cp = oracle.prepared_inputs(q, c[rows], metric, dtype)
    truth = oracle.scores_fp64(qp, cp)
    if mode == "biased":
        truth = truth + w * bias[rows].astype(np.float64)[None, :]
    m = min(k, rows.size)
    local = np.full_like(idx, -1)
    valid = idx >= 0
    local[valid] = np.searchsorted(rows, idx[valid])
    assert (idx[:, m:] == -1).all() and (idx[:, :m] >= 0).all(), "padding"
    assert (rows[local[:, :m]] == idx[:, :m]).all(), "ids outside the allowed rows"
    stats = oracle.check_topk_against_truth(truth, local, scores, k, gap=1e-6, score_tol=1e-5)
    assert stats["recall"] == 1.0, stats
    if watchdog:
        faulthandler.cancel_dump_traceback_later()
    return (f"case {case}: n={n} d={d} {dtype} {metric} nq={nq} k={k} algo={algo} mask={use_mask} {mode}"
            f"{f' grid={grid}' if grid else ''} ok ({time.time() - t0:.1f}s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="bf16 x 768 corpora of 1M-2.5M rows through the MFMA path (threshold estimates at scale)")
    ap.add_argument("--pairs", action="store_true", help="bf16 x 1024 with 193+ queries: the paired (k-split) pass, over several grids")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    case = 0
    while time.time() < t_end:
        case += 1
        print(one_case(rng, args.big, case, pairs=args.pairs), flush=True)
    print(f"stress: {case} cases passed", flush=True)


if __name__ == "__main__":
    main()
