#!/usr/bin/env python3
"""How large must the threshold sample be?  One corpus, TS_MFMA_FIRST_ROWS = 8192 / 4096 / 2048: over many DIFFERENT query
batches the queries the estimate failed for (exact re-runs), the candidates per query of the full pass, and the time of
a search (device buffers, back to back).

    python tools/sample_size_ab.py --rows 10000000 --batches 40
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--sizes", default="8192,4096,2048")
    args = ap.parse_args()
    import torch
    import synthetic
    import theoremsearch_amd as ts
    ch = synthetic.CHUNK_ROWS
    ix = ts.TheoremIndex(args.rows, 768, dtype="bf16", metric="ip")

    def make(c):
        data = synthetic.synth_chunk(c, ch, 768, bf16=True)
        ix.upload(data[: min(args.rows, (c + 1) * ch) - c * ch], c * ch)

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(make, range((args.rows + ch - 1) // ch)))
    out = {"rows": args.rows, "nq": args.nq, "batches": args.batches, "sizes": {}}
    qd = torch.from_numpy(synthetic.synth_queries(0, args.nq, 768, bf16=True).view(np.int16)).cuda()
    o_s = torch.empty((args.nq, 10), dtype=torch.float32, device="cuda")
    o_i = torch.empty((args.nq, 10), dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    for size in [int(x) for x in args.sizes.split(",")]:
        ix.set_option("TS_MFMA_FIRST_ROWS", size)
        reruns, cands = [], []
        for b in range(args.batches):
            q = synthetic.bf16_bits_to_f32(synthetic.synth_queries(100 + b, args.nq, 768, bf16=True))
            _, _, s_ = ix.search(q, 10, algo="mfma", return_stats=True)
            reruns.append(int(s_["fallback_queries"]))
            cands.append(s_["candidates"] / args.nq)
        for _ in range(20):
            ix.search_device(qd.data_ptr(), "bf16", args.nq, 10, o_s.data_ptr(), o_i.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ix.search_device(qd.data_ptr(), "bf16", args.nq, 10, o_s.data_ptr(), o_i.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        out["sizes"][str(size)] = {"queries": args.batches * args.nq, "reruns_total": int(sum(reruns)), "reruns_max_per_batch": int(max(reruns)),
                                   "candidates_per_query_mean": round(float(np.mean(cands)), 1),
                                   "candidates_per_query_min_max": [round(float(min(cands)), 1), round(float(max(cands)), 1)],
                                   "ms_per_search": round((time.perf_counter() - t0) / 200 * 1e3, 4)}
        print(size, out["sizes"][str(size)], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
