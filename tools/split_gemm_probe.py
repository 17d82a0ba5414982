#!/usr/bin/env python3
"""fp32-class GEMMs of the encoder forward on the bf16 matrix pipe: x [M x K] fp32 and w [N x K] fp32 as bf16 pieces
(ts_split_pieces: hi = bf16(v), lo = bf16(v - hi)), ONE bf16 GEMM with fp32 accumulation over the three-fold depth
(torch.mm(.., out_dtype=torch.float32) = hipBLASLt) against the library's fp32 GEMM and the plain bf16 GEMM: time per call and
error against fp64, for the four GEMM shapes of a BERT-base layer at 256 x 32 and 256 x 128 tokens."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from theoremsearch_amd import _ffi  # noqa: E402

lib = _ffi.load()


def split(x, pattern):
    rows, k = x.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=x.device)
    _ffi.check(lib.ts_split_pieces(0, C.c_void_p(x.data_ptr()), rows, k, pattern, C.c_void_p(out.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    torch.manual_seed(0)
    res = []
    for M in (8192, 32768):
        for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") * 0.02
            ref = (x[:512].double() @ w.double().t())
            scale = (x[:512].double().norm(dim=1, keepdim=True) * w.double().norm(dim=1)[None, :])      # |x||w| per entry
            w3 = split(w, 1)
            xb, wb = x.bfloat16(), w.bfloat16()
            row = {"M": M, "K": K, "N": N}
            y32 = torch.nn.functional.linear(x, w)
            row["fp32_us"] = round(timeit(lambda: torch.nn.functional.linear(x, w)), 1)
            row["fp32_err"] = float(((y32[:512].double() - ref).abs() / scale).max())
            try:
                y3 = torch.mm(split(x, 0), w3.t(), out_dtype=torch.float32)
                row["split_us"] = round(timeit(lambda: torch.mm(split(x, 0), w3.t(), out_dtype=torch.float32)), 1)
                x3 = split(x, 0)
                row["split_gemm_only_us"] = round(timeit(lambda: torch.mm(x3, w3.t(), out_dtype=torch.float32)), 1)
                row["split_err"] = float(((y3[:512].double() - ref).abs() / scale).max())
            except Exception as e:          # noqa: BLE001
                row["split_error"] = f"{type(e).__name__}: {e}"[:300]
            try:
                yb = torch.mm(xb, wb.t(), out_dtype=torch.float32)
                row["bf16_f32out_us"] = round(timeit(lambda: torch.mm(xb, wb.t(), out_dtype=torch.float32)), 1)
                row["bf16_err"] = float(((yb[:512].double() - ref).abs() / scale).max())
            except Exception as e:          # noqa: BLE001
                row["bf16_f32out_error"] = f"{type(e).__name__}: {e}"[:300]
            row["bf16_us"] = round(timeit(lambda: torch.nn.functional.linear(xb, wb)), 1)
            flops = 2.0 * M * K * N
            row["fp32_tflops"] = round(flops / row["fp32_us"] / 1e6, 1)
            if "split_gemm_only_us" in row:
                row["split_effective_tflops"] = round(flops / row["split_us"] / 1e6, 1)
            print(json.dumps(row), flush=True)
            res.append(row)


if __name__ == "__main__":
    main()
