"""Timing experiments on ts_linear_bf16: run once per library (TS_LINEAR_LIB = a -DTS_LIN_DBG=n build of `make -C tools/microbench gemm-dbg`: 1 no stores, 2 no DMA,
4 no MFMA, 8 fragment reads of the first unit only).  Prints one JSON line per library."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.linear_ab import ts_linear

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 8192
res = {"lib": os.path.basename(os.environ.get("TS_LINEAR_LIB", "product"))}
for name, n, k, tile in [("qkv288", 2304, 768, 288), ("up192", 3072, 768, 192), ("down96", 768, 3072, 96), ("out96", 768, 768, 96)]:
    x = (torch.randn(M, k, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * (k ** -0.5)).to(torch.bfloat16)
    b = (torch.randn(n, device=dev) * 0.1).to(torch.bfloat16)
    y = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ts_linear(x, w, b, 0, tile, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ts_linear(x, w, b, 0, tile, out=y)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    res[name] = round(best, 1)
print(json.dumps(res), flush=True)
