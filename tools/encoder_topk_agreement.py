#!/usr/bin/env python3
"""What the encoder's arithmetic does to the ANSWERS: 256 synthetic queries encoded by the same random-init stand-in in fp32 (the
reference's arithmetic), fp32x3 (fp32 storage, GEMMs from bf16 pieces) and bf16, each searched over the same 10M x 768 bf16 index
(d = 1024 for the Qwen3 shape), top-10: how many ids agree with the fp32 encoder's answer, position by position and as sets.
The index rounds every query to bf16 before it multiplies, so two encoders that agree to well below 2^-9 per component give
the same rounded query almost everywhere; differences show up where the 10th and 11th scores of a query are near-ties."""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic  # noqa: E402
import theoremsearch_amd as ts  # noqa: E402
from theoremsearch_amd.encoder import SentenceEncoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--seq-len", type=int, default=32)
    args = ap.parse_args()
    names = {"bert": "math-similarity/Bert-MLM_arXiv-MP-class_zbMath", "qwen": "Qwen/Qwen3-Embedding-0.6B", "gemma": "google/embeddinggemma-300m"}
    nq, K = 256, 10
    g = torch.Generator(device="cpu").manual_seed(5678)
    tok = torch.randint(1000, 30000, (nq, args.seq_len), generator=g).cuda()
    tok[:, 0], tok[:, -1] = 101, 102
    mask = torch.ones_like(tok)
    indexes = {}
    for enc_key, name in names.items():
        d = 1024 if enc_key == "qwen" else 768
        if d not in indexes:
            ch = synthetic.CHUNK_ROWS
            ix = ts.TheoremIndex(args.rows, d, dtype="bf16", metric="ip")
            with ThreadPoolExecutor(16) as ex:
                list(ex.map(lambda c: ix.upload(synthetic.synth_chunk(c, ch, d, bf16=True)[: min(args.rows, (c + 1) * ch) - c * ch], c * ch),
                            range((args.rows + ch - 1) // ch)))
            indexes[d] = ix
        ix = indexes[d]
        answers, embs = {}, {}
        for mode in ("fp32", "fp32x3", "bf16"):
            enc = SentenceEncoder(name, allow_random_init=True, dtype=torch.bfloat16 if mode == "bf16" else torch.float32,
                                  fp32_gemm="bf16x3" if mode == "fp32x3" else "blas")
            with torch.inference_mode():
                emb = enc.pool(enc.forward_hidden(tok, mask, no_padding=True), mask, True).float().cpu().numpy()
            embs[mode] = emb
            answers[mode] = ix.search(emb, K)
            del enc
            torch.cuda.empty_cache()
        ref_s, ref_i = answers["fp32"]
        row = {"encoder": enc_key, "tokens": args.seq_len, "rows": args.rows, "queries": nq, "k": K}
        for mode in ("fp32x3", "bf16"):
            s_, i_ = answers[mode]
            row[mode] = {"one_minus_min_cosine_vs_fp32": float(1.0 - np.min(np.sum(embs["fp32"].astype(np.float64) * embs[mode].astype(np.float64), axis=1))),
                         "ids_equal_by_position": int(np.sum(i_ == ref_i)), "positions": int(ref_i.size),
                         "queries_with_the_same_top10_set": int(sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(i_, ref_i))),
                         "queries_with_the_same_best_row": int(np.sum(i_[:, 0] == ref_i[:, 0])),
                         "max_abs_score_difference": float(np.abs(s_ - ref_s).max())}
        print(json.dumps(row), flush=True)
    for ix in indexes.values():
        ix.close()


if __name__ == "__main__":
    main()
