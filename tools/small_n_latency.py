#!/usr/bin/env python3
"""Single-query latency on the corpus sizes the reference's apps hold (app_showcase_model.py: a few thousand theorems,
cos_sim + topk(200); streamlit_app.py: one query against the table): device-resident query, device results, per-search
time over 200 back-to-back searches, for N = 1k ... 1M rows x 768 fp32, cosine, k = 10 and 200."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import theoremsearch_amd as ts
    from oracle import oracle
    out = {}
    for n in (1_000, 10_000, 100_000, 1_000_000):
        q, c = oracle.inputs(n, 1, 768, 7 + n, "cos")
        with ts.TheoremIndex.from_embeddings(c, dtype="f32", metric="cos") as ix:
            qd = torch.from_numpy(q).cuda()
            st = torch.cuda.Stream()
            for k in (10, 200):
                kk = min(k, n)
                os_ = torch.empty((1, kk), dtype=torch.float32, device="cuda")
                oi = torch.empty((1, kk), dtype=torch.int64, device="cuda")
                for _ in range(20):
                    ix.search_device(qd.data_ptr(), "f32", 1, kk, os_.data_ptr(), oi.data_ptr(), st.cuda_stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    ix.search_device(qd.data_ptr(), "f32", 1, kk, os_.data_ptr(), oi.data_ptr(), st.cuda_stream)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 200 * 1e6
                # one synchronous call: what a Streamlit session thread sees (host query in, host results out)
                t1 = time.perf_counter()
                for _ in range(50):
                    ix.search(q, kk)
                sync_us = (time.perf_counter() - t1) / 50 * 1e6
                out[f"n{n}_k{k}"] = {"pipelined_us": round(us, 1), "synchronous_host_us": round(sync_us, 1)}
                print(f"N = {n:>9}, k = {k:>3}: {us:8.1f} us per search back to back, {sync_us:8.1f} us per synchronous host call", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
