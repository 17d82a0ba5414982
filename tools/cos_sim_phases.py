#!/usr/bin/env python3
"""Where a `util.cos_sim` call of the evaluation script's shape goes (BASELINE.json configs[0]: 73 queries x 1,000 theorems x
768, compare_embeddings.py:61): the throw-away index (create, upload + normalise), the score kernel with its two copies, the
close - and beside it the host part of `evaluate_retrieval` (the selection and the six metrics).  One JSON line."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from theoremsearch_amd import compare_embeddings as ce  # noqa: E402
from theoremsearch_amd import util  # noqa: E402
from theoremsearch_amd.index import TheoremIndex  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--nq", type=int, default=73)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    n, nq, d = args.rows, args.nq, 768
    rng = np.random.default_rng(7)
    s_emb = synthetic.synth_chunk(0, n, d)
    gold = rng.choice(n, nq, replace=False)
    q_emb = (s_emb[gold] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.9 / np.sqrt(d))).astype(np.float32)
    qrels = {j: {int(i): (1.0 if i == gold[j] else 0.5) for i in range(int(gold[j]) // 4 * 4, min(n, int(gold[j]) // 4 * 4 + 4))}
             for j in range(nq)}
    t = {"create": 0.0, "upload": 0.0, "scores": 0.0, "close": 0.0}
    for rep in range(args.reps + 20):
        if rep == 20:
            t = dict.fromkeys(t, 0.0)
        a = time.perf_counter()
        ix = TheoremIndex(n, d, dtype="f32", metric="cos", device=0)
        b = time.perf_counter()
        ix.upload(s_emb, 0)
        c = time.perf_counter()
        sim = ix.scores(q_emb)
        e = time.perf_counter()
        ix.close()
        f = time.perf_counter()
        t["create"] += b - a
        t["upload"] += c - b
        t["scores"] += e - c
        t["close"] += f - e
    phases = {k_: round(v / args.reps * 1e3, 4) for k_, v in t.items()}
    a = time.perf_counter()
    for _ in range(args.reps):
        util.cos_sim(q_emb, s_emb)
    cos_ms = (time.perf_counter() - a) / args.reps * 1e3
    a = time.perf_counter()
    for _ in range(args.reps):
        shared = ce._SharedRanking(sim)
        ce._top(shared, 3)
        with contextlib.redirect_stdout(io.StringIO()):
            for fn, kk in ((ce.precision_at_k, 1), (ce.hit_at_k, 3), (ce.mrr_at_k, 3), (ce.ndcg_at_k, 3), (ce.err_at_k, 3),
                           (ce.q_measure_at_k, 3)):
                fn(shared, qrels, k=kk)
    host_ms = (time.perf_counter() - a) / args.reps * 1e3
    print(json.dumps({"rows": n, "nq": nq, "reps": args.reps, "phases_ms": phases, "cos_sim_ms": round(cos_ms, 4),
                      "selection_and_six_metrics_ms": round(host_ms, 4)}))


if __name__ == "__main__":
    main()
