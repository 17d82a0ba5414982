"""Device time of the encoder-side kernels of libtsearch at the shapes of the encoder-in-loop step (256 sequences x 32 tokens x
768, bf16): ts_add_layernorm, ts_pool_normalize, ts_embed_layernorm and ts_attention_short (against torch's attention), us per call (hipEvent around 50 calls, best of 5)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from theoremsearch_amd import _ffi  # noqa: E402

lib = _ffi.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, S, D = 256, 32, 768
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=50, rounds=5):
    for _ in range(5):
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
    return round(best, 2)


res = {}
for dt, code in ((torch.bfloat16, 1), (torch.float32, 0)):
    a = torch.randn(B * S, D, device=dev).to(dt)
    b = torch.randn(B * S, D, device=dev).to(dt)
    g = torch.ones(D, device=dev).to(dt)
    be = torch.zeros(D, device=dev).to(dt)
    out = torch.empty_like(a)
    res[f"add_layernorm_{'bf16' if code else 'f32'}"] = timed(lambda: lib.ts_add_layernorm(
        0, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(be.data_ptr()), 1e-12, B * S, D, code,
        C.c_void_p(out.data_ptr()), st))
    hidden = a.view(B, S, D)
    mask = torch.ones((B, S), dtype=torch.int64, device=dev)
    pooled = torch.empty((B, D), dtype=dt, device=dev)
    for pooling, name in ((0, "mean"), (2, "cls")):
        res[f"pool_{name}_{'bf16' if code else 'f32'}"] = timed(lambda: lib.ts_pool_normalize(
            0, C.c_void_p(hidden.data_ptr()), code, C.c_void_p(mask.data_ptr()), B, S, D, pooling, 1, C.c_void_p(pooled.data_ptr()), code, D, st))
for dt, code in ((torch.bfloat16, 1),):
    V, P, T = 30522, 512, 2
    word, pos, typ = (torch.randn(n, D, device=dev).to(dt) for n in (V, P, T))
    g, be = torch.ones(D, device=dev).to(dt), torch.zeros(D, device=dev).to(dt)
    ids = torch.randint(0, V, (B, S), device=dev)
    out = torch.empty((B, S, D), dtype=dt, device=dev)
    res["embed_layernorm_bf16"] = timed(lambda: lib.ts_embed_layernorm(
        0, C.c_void_p(ids.data_ptr()), None, C.c_void_p(word.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(typ.data_ptr()), V, P, T,
        C.c_void_p(g.data_ptr()), C.c_void_p(be.data_ptr()), 1e-12, B * S, S, D, code, C.c_void_p(out.data_ptr()), st))
H = 12
qkv = torch.randn((B, S, 3, H, 64), device=dev).to(torch.bfloat16)
ctx = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device=dev)
res["attention_bf16"] = timed(lambda: lib.ts_attention_short(0, C.c_void_p(qkv.data_ptr()), None, B, S, H, 64, C.c_void_p(ctx.data_ptr()), st))
qv = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
res["torch_sdpa_bf16"] = timed(lambda: torch.nn.functional.scaled_dot_product_attention(qv[0], qv[1], qv[2]))
print(json.dumps(res))
