#!/usr/bin/env python3
"""fp64 truth of BASELINE.json configs[2] at its full size, as a small fixture (test infrastructure, like everything
under oracle/): the k + 64 best rows (canonical order: score descending, row ascending) of each of the 256 benchmark
queries over the 10M x 768 bf16 synthetic corpus, plus digests of the inputs they belong to.

    python oracle/gen_fullsize_truth.py            # -> tests/golden/fullsize_c3_truth.npz (about 300 KB)

Inputs: synthetic.synth_chunk / synth_queries (random-normal rows, L2-normalised in fp32, rounded to
bf16) - the rows bench.py and tests/test_fullsize_gpu.py upload.  The scores are fp64 products of the bf16 values, reduced
chunk by chunk through oracle.ChunkedTruth (the protocol of compare_embeddings.py:61,105 at a size whose [256 x 10M] matrix
does not fit).  tests/test_fullsize_gpu.py checks the digests on the box before trusting the file and recomputes the truth
there when they differ (another numpy's generator stream, say).
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

ROWS, D, NQ, K = 10_000_000, 768, 256, 10
DIGEST_CHUNKS = (0, 17, 39)


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main(out_path: str) -> None:
    ch = synthetic.CHUNK_ROWS
    q = synthetic.synth_queries(0, NQ, D, bf16=True)
    truth = oracle.ChunkedTruth(oracle.bf16_bits_to_f32(q), np.zeros((NQ, K), dtype=np.int64), K)
    chunks = list(range((ROWS + ch - 1) // ch))
    digests = {}
    t0 = time.time()

    def one(c):
        data = synthetic.synth_chunk(c, ch, D, bf16=True)[: min(ROWS, (c + 1) * ch) - c * ch]
        if c in DIGEST_CHUNKS:
            digests[c] = digest(data)
        return c, truth.q64 @ oracle.bf16_bits_to_f32(data).astype(np.float64).T

    with ThreadPoolExecutor(4) as ex:
        for c, s in ex.map(one, chunks):
            truth.add_scores(s, c * ch)
            print(f"chunk {c + 1}/{len(chunks)}  {time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(out_path, best_s=truth.best_s, best_i=truth.best_i, n=np.int64(truth.n), k=np.int64(K),
                        rows=np.int64(ROWS), d=np.int64(D), nq=np.int64(NQ), query_digest=digest(q),
                        chunk_ids=np.array(DIGEST_CHUNKS), chunk_digests=np.array([digests[c] for c in DIGEST_CHUNKS]))
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "fullsize_c3_truth.npz"))
