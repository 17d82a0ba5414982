#!/usr/bin/env python3
"""fp64 truth of BASELINE.json configs[2] and configs[3] at their full sizes, as small fixtures (test infrastructure,
like everything under oracle/): the k + 64 best rows (canonical order: score descending, row ascending) of each of the
256 benchmark queries over the 10M x 768 (configs[2]) and 50M x 768 (configs[3]) bf16 synthetic corpus, plus a SHA-256
digest of the queries and of EVERY corpus chunk they belong to.

    python oracle/gen_fullsize_truth.py            # -> tests/golden/fullsize_c3_truth.npz, fullsize_c4_truth.npz
    python oracle/gen_fullsize_truth.py c3         # only the 10M-row file

Inputs: synthetic.synth_chunk / synth_queries (random-normal rows, L2-normalised in fp32, rounded to
bf16) - the rows bench.py and tests/test_fullsize_gpu.py upload; the 10M-row corpus is the first 40 chunks of the 50M-row
one, so one pass writes both files.  The scores are fp64 products of the bf16 values, reduced chunk by chunk through
oracle.ChunkedTruth (the protocol of compare_embeddings.py:61,105 at a size whose [256 x N] matrix does not fit).
tests/test_fullsize_gpu.py hashes every chunk it generates on the box and trusts the file only when all of them match; it
recomputes the truth there when they differ (another numpy's generator stream, say).
"""
import hashlib
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

D, NQ, K = 768, 256, 10
ROWS = {"c3": 10_000_000, "c4": 50_000_000}


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(path: str, truth, rows: int, q: np.ndarray, digests: dict) -> None:
    nchunks = (rows + synthetic.CHUNK_ROWS - 1) // synthetic.CHUNK_ROWS
    np.savez_compressed(path, best_s=truth.best_s, best_i=truth.best_i, n=np.int64(truth.n), k=np.int64(K),
                        rows=np.int64(rows), d=np.int64(D), nq=np.int64(NQ), query_digest=digest(q),
                        chunk_ids=np.arange(nchunks), chunk_digests=np.array([digests[c] for c in range(nchunks)]))
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def main(which, out_dir: str) -> None:
    ch = synthetic.CHUNK_ROWS
    rows_max = max(ROWS[w] for w in which)
    q = synthetic.synth_queries(0, NQ, D, bf16=True)
    truth = oracle.ChunkedTruth(oracle.bf16_bits_to_f32(q), np.zeros((NQ, K), dtype=np.int64), K)
    chunks = list(range((rows_max + ch - 1) // ch))
    digests = {}
    t0 = time.time()

    def one(c):
        data = synthetic.synth_chunk(c, ch, D, bf16=True)[: min(rows_max, (c + 1) * ch) - c * ch]
        digests[c] = digest(data)
        return c, truth.q64 @ oracle.bf16_bits_to_f32(data).astype(np.float64).T

    with ThreadPoolExecutor(4) as ex:
        for c, s in ex.map(one, chunks):
            truth.add_scores(s, c * ch)
            print(f"chunk {c + 1}/{len(chunks)}  {time.time() - t0:.0f}s", flush=True)
            for w in which:
                if (c + 1) * ch == ROWS[w]:
                    save(os.path.join(out_dir, f"fullsize_{w}_truth.npz"), truth, ROWS[w], q, digests)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in ROWS] or ["c3", "c4"]
    main(which, os.path.join(ROOT, "tests", "golden"))
