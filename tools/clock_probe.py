#!/usr/bin/env python3
"""In-kernel clock of the full pass of configs[2] (MI355X_MICROARCH.md "DVFS give-back" item 6): >= 2 s of back-to-back
product launches on random data, then launches of the diagnostic build (option TS_MFMA_VARIANT = 3: s_memtime and
s_memrealtime stamped once around the tile loop, values written to a buffer of their own), median over workgroups.
Prints one JSON object with the clock, the cycles per unit, and the derived matrix-pipe utilisation."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the timing-only kernel variants live in the diagnostic build of the library (make -C theoremsearch_amd/csrc diag)
_DIAG = os.path.join(ROOT, "theoremsearch_amd", "libtsearch_diag.so")
if os.path.exists(_DIAG):
    os.environ.setdefault("TS_LIB", _DIAG)


def main():
    import torch
    import synthetic
    import theoremsearch_amd as ts
    rows, d, nq = 10_000_000, 768, 256
    ch = synthetic.CHUNK_ROWS
    ix = ts.TheoremIndex(rows, d, dtype="bf16", metric="ip")

    def make(c):
        ix.upload(synthetic.synth_chunk(c, ch, d, bf16=True), c * ch)

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(make, range(rows // ch)))
    q = synthetic.synth_queries(0, nq, d, bf16=True)
    qd = torch.from_numpy(q.view(np.int16)).cuda()
    out_s = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
    out_i = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()

    def run(n):
        for _ in range(n):
            ix.search_device(qd.data_ptr(), "bf16", nq, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()

    t0 = time.time()
    while time.time() - t0 < 2.5:
        run(50)
    ix.profile_enable(True)
    run(50)
    prod = ix.profile_read()
    ix.set_option("TS_MFMA_VARIANT", 3)
    probes = []
    for _ in range(5):
        run(20)
        probes.append(ix.probe_read())
    diag = ix.profile_read()
    ix.profile_enable(False)
    p = probes[-1]
    units = p["units_per_workgroup"]
    k_steps, mfma_per_step, cyc_per_mfma = 12, 8, 16          # unit = 12 k-steps x (2 row blocks x 4 query blocks) x 16 cycles
    busy = k_steps * mfma_per_step * cyc_per_mfma / p["cycles_per_unit"] if p["cycles_per_unit"] else None
    print(json.dumps({
        "workload": "10M x 768 bf16, batch 256, mfma16_topk_kernel<768, 4>",
        "product_kernel_ms": prod["total_ms"] / max(1, prod["launches"]),
        "diagnostic_kernel_ms": diag["total_ms"] / max(1, diag["launches"]),
        "in_kernel_clock_ghz": p["ghz"], "cycles_per_unit": p["cycles_per_unit"], "units_per_workgroup": units,
        "mfma_cycles_per_unit_minimum": k_steps * mfma_per_step * cyc_per_mfma,
        "matrix_pipe_busy_fraction": busy,
        "formula": "busy = (12 k-steps x 8 MFMA 16x16x32 x 16 cycles per SIMD) / measured shader cycles per unit; "
                   "clock = delta s_memtime / delta s_memrealtime x 100 MHz, median over 256 workgroups",
        "all_probes": probes}, indent=1))


if __name__ == "__main__":
    main()
