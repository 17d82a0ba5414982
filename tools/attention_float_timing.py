#!/usr/bin/env python3
"""ts_attention_float against torch's attention as the fused forwards ran it (scaled_dot_product_attention on views of the stacked
projection, the context transposed back: the copies are part of what the kernel replaces), fp32, 256 sequences, the three
families' head shapes at 32 and 128 tokens: us per call (hipEvent around 30 calls, best of 3)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theoremsearch_amd.fused_forward import attention_float  # noqa: E402

F = torch.nn.functional


def timed(fn, n=30, rounds=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return round(best, 1)


B = 256
for name, hq, hkv, hd, causal, scale in (("bert", 12, 12, 64, False, 0.125), ("qwen3", 16, 8, 128, True, 128 ** -0.5), ("gemma3", 3, 1, 256, False, 0.0625)):
    for S in (32, 128):
        qkv = torch.randn(B, S, (hq + 2 * hkv) * hd, device="cuda")
        nq, nkv = hq * hd, hkv * hd

        def torch_way():
            q = qkv[..., :nq].view(B, S, hq, hd).transpose(1, 2)
            k = qkv[..., nq:nq + nkv].view(B, S, hkv, hd).transpose(1, 2)
            v = qkv[..., nq + nkv:].view(B, S, hkv, hd).transpose(1, 2)
            if hq != hkv:
                k, v = k.repeat_interleave(hq // hkv, dim=1), v.repeat_interleave(hq // hkv, dim=1)
            return F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale).transpose(1, 2).reshape(B, S, nq)

        want = torch_way()
        got = attention_float(qkv, None, B, S, hq, hkv, hd, causal, scale)[0]
        row = {"family": name, "tokens": S, "heads": f"{hq}/{hkv} x {hd}", "causal": causal,
               "max_abs_diff": float((got - want).abs().max()),
               "torch_us": timed(torch_way), "ts_attention_float_us": timed(lambda: attention_float(qkv, None, B, S, hq, hkv, hd, causal, scale)),
               "pieces_only_us": timed(lambda: attention_float(qkv, None, B, S, hq, hkv, hd, causal, scale, want_pieces=True, want_context=False)),
               "gflop": round(4.0 * B * hq * S * S * hd / 1e9 * (0.5 if causal else 1.0), 2)}
        print(json.dumps(row), flush=True)
