#!/usr/bin/env python3
"""ts_shards_search (one process, one host thread per shard) on a one-GPU box: 8 shards of a 10M x 768 bf16 corpus, all on
device 0, 256 host queries in, host results out.  Reports ms per call over `--calls` calls behind a warm-up; run it twice
(TS_SHARDS_THREADS=0 = every device enqueued from the caller's thread, in order) for the A/B of DESIGN.md section 7.
The eight searches share ONE device here, so a call cannot take less than eight shard passes; what the threads change is the
host time in front of each device's first launch."""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
from theoremsearch_amd.distributed import Shards  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--calls", type=int, default=300)
    ap.add_argument("--nq", type=int, default=256)
    args = ap.parse_args()
    ch = synthetic.CHUNK_ROWS
    sh = Shards(args.rows, 768, args.shards, dtype="bf16", metric="ip", devices=[0] * args.shards)
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(lambda c: sh.upload(synthetic.synth_chunk(c, ch, 768, bf16=True)[: min(args.rows, (c + 1) * ch) - c * ch], c * ch),
                    range((args.rows + ch - 1) // ch)))
    q = synthetic.synth_queries(0, args.nq, 768, bf16=True)
    for _ in range(50):
        sh.search(q, 10)
    t0 = time.perf_counter()
    for _ in range(args.calls):
        s, i = sh.search(q, 10)
    dt = (time.perf_counter() - t0) / args.calls * 1e3
    print(json.dumps({"threads": os.environ.get("TS_SHARDS_THREADS", "1"), "rows": args.rows, "shards": args.shards, "nq": args.nq,
                      "calls": args.calls, "ms_per_call": round(dt, 4), "ms_per_shard": round(dt / args.shards, 4),
                      "first_ids": i[0, :3].tolist()}))
    sh.close()


if __name__ == "__main__":
    main()
