"""The hand-written GEMM experiment (tools/microbench/linear_gemm, not part of libtsearch; profiles/r03_linear_gemm_ab.txt):
ts_linear_bf16 against torch's F.linear (the BLAS library) on the GEMM shapes of a BERT-base forward at 256 x 32 tokens:
correctness against an fp64 product of the same bf16 operands, then interleaved timing rounds in one process.

    python tools/linear_ab.py [--m 8192] [--rounds 5] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_LIB = None


def _lib():
    """libts_linear.so (`make -C tools/microbench gemm`), or the timing-only build TS_LINEAR_LIB names."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("TS_LINEAR_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench", "build",
                                                               "libts_linear.so")
        _LIB = C.CDLL(path)
        _LIB.ts_linear_bf16.restype = C.c_int
        _LIB.ts_linear_bf16.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_int, C.c_int, C.c_void_p]
        _LIB.ts_linear_last_error.restype = C.c_char_p
    return _LIB


def ts_linear(x, w, b, act=0, tile=0, out=None):
    lib = _lib()
    m, k = x.shape
    n = w.shape[0]
    y = out if out is not None else torch.empty((m, n), dtype=torch.bfloat16, device=x.device)
    rc = lib.ts_linear_bf16(x.device.index or 0, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()),
                            C.c_void_p(b.data_ptr()) if b is not None else None, C.c_void_p(y.data_ptr()), m, n, k, act, tile,
                            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"ts_linear_bf16: {rc} {lib.ts_linear_last_error().decode()}")
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    F = torch.nn.functional
    shapes = [("qkv", 2304, 768, 0, [0, 288, 192, 96]), ("attn_out", 768, 768, 0, [0, 96]), ("ffn_up+gelu", 3072, 768, 1, [0, 192, 96]),
              ("ffn_down", 768, 3072, 0, [0, 96])]
    report = {"m": args.m, "shapes": []}
    for name, n, k, act, tiles in shapes:
        x = (torch.randn(args.m, k, device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * (k ** -0.5)).to(torch.bfloat16)
        b = (torch.randn(n, device=dev) * 0.1).to(torch.bfloat16)
        # truth: fp64 product of the same operands, then the same two roundings
        ref = (x[:512].double() @ w.double().T + b.double()).float().to(torch.bfloat16)
        if act:
            ref = F.gelu(ref.float()).to(torch.bfloat16)
        row = {"name": name, "n": n, "k": k, "act": act, "flops": 2.0 * args.m * n * k}
        for tile in tiles:
            y = ts_linear(x, w, b, act, tile)
            torch.cuda.synchronize()
            d = (y[:512].float() - ref.float()).abs()
            tol = 2.0 ** -7 * ref.float().abs().clamp(min=1e-2)            # one bf16 ulp of the result
            bad = int((d > tol).sum())
            row[f"tile{tile}_max_abs_err"] = float(d.max())
            row[f"tile{tile}_beyond_one_ulp"] = bad
            yt = F.linear(x, w, b)
            if act:
                yt = F.gelu(yt)
            row[f"tile{tile}_differs_from_torch"] = int((yt != y).sum())
            # rows beyond 512 checked against torch's own result (loose: both are within an ulp of the truth)
            row[f"tile{tile}_max_diff_vs_torch"] = float((yt.float() - y.float()).abs().max())
        # ragged m: the last row block is clamped / predicated
        mr = args.m - 37
        yr = ts_linear(x[:mr].contiguous(), w, b, act, 0)
        yt = F.linear(x[:mr], w, b)
        if act:
            yt = F.gelu(yt)
        row["ragged_max_diff_vs_torch"] = float((yt.float() - yr.float()).abs().max())
        # timing: interleaved rounds
        def run_torch():
            o = F.linear(x, w, b)
            return F.gelu(o) if act else o
        variants = [("torch", run_torch)] + [(f"tile{t}", (lambda t=t: ts_linear(x, w, b, act, t))) for t in tiles]
        times = {nm: [] for nm, _ in variants}
        for _ in range(3):
            for _, fn in variants:
                fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for nm, fn in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                e1.synchronize()
                times[nm].append(e0.elapsed_time(e1) / args.iters * 1e3)
        for nm in times:
            t = sorted(times[nm])
            row[nm + "_us_median"] = round(t[len(t) // 2], 2)
            row[nm + "_us_min"] = round(t[0], 2)
            row[nm + "_tflops"] = round(row["flops"] / (t[len(t) // 2] * 1e-6) / 1e12, 1)
        report["shapes"].append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
