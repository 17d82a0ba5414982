"""Times the hot-path extras on the GPU box: filtered search (row bitmask) and rank-of-gold (counting pass),
on the C2 shape (1M x 768 fp32) and on a 4M x 768 bf16 index.  Prints one JSON object; kernel time is the
library's own hipEvent bracket (TheoremIndex.profile_read), wall time includes the host call + copies.

    python tools/measure_extras.py > gpurun_out/extras.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import theoremsearch_amd as ts  # noqa: E402
import synthetic  # noqa: E402


def build(n, dtype):
    ix = ts.TheoremIndex(n, 768, dtype=dtype, metric="ip")
    CH = 250000
    for c in range((n + CH - 1) // CH):
        rows = synthetic.synth_chunk(c, CH, 768, bf16=(dtype == "bf16"))[: min(CH, n - c * CH)]
        ix.upload(rows, c * CH)
    return ix


def timed(ix, fn, reps=20):
    fn()
    ix.profile_enable(True)
    ix.profile_read()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    wall = (time.perf_counter() - t0) / reps
    p = ix.profile_read()
    ix.profile_enable(False)
    kern = p["total_ms"] / max(1, p["launches"])
    return {"wall_ms": round(wall * 1e3, 4), "kernel_ms": round(kern, 4), "launches_per_call": p["launches"] / reps}


def main():
    out = {}
    for name, n, dtype in (("1Mx768_f32", 1_000_000, "f32"), ("4Mx768_bf16", 4_000_000, "bf16")):
        ix = build(n, dtype)
        elem = 4 if dtype == "f32" else 2
        gb = n * 768 * elem / 1e9
        rng = np.random.default_rng(1)
        q1 = synthetic.synth_queries(0, 1)
        q4 = synthetic.synth_queries(1, 4)
        mask = rng.random(n) < 0.1
        res = {
            "search_b1": timed(ix, lambda: ix.search(q1, 10, algo="scan")),
            "search_b1_k200": timed(ix, lambda: ix.search(q1, 200, algo="scan")),
            "search_b4_k200_scan": timed(ix, lambda: ix.search(q4, 200, algo="scan")),
            "search_b4_k200_auto": timed(ix, lambda: ix.search(q4, 200)),
            "search_b1_mask10pct": timed(ix, lambda: ix.search(q1, 10, mask=mask)),
            "search_b4_mask10pct": timed(ix, lambda: ix.search(q4, 10, mask=mask)),
            "rank_of_b1": timed(ix, lambda: ix.rank_of(q1, [n // 3])),
            "rank_of_b4": timed(ix, lambda: ix.rank_of(q4, [5, n // 3, n // 2, n - 1])),
        }
        for v in res.values():
            if "kernel_ms" in v:
                v["GBps"] = round(gb / (v["kernel_ms"] * 1e-3), 1) if v["kernel_ms"] else None
                v["frac_of_8TBps"] = round(v["GBps"] / 8000.0, 4) if v["GBps"] else None
        if dtype == "bf16":
            # filtered batch search: copy the allowed rows once, then unfiltered batch-256 searches of the copy
            q256 = synthetic.synth_queries(2, 256)
            mask50 = rng.random(n) < 0.5
            res["search_b256_mask50pct_mfma"] = timed(ix, lambda: ix.search(q256, 10, mask=mask50))
            res["search_b256_unfiltered"] = timed(ix, lambda: ix.search(q256, 10))
            t0 = time.perf_counter()
            sub = ix.subset(mask)
            t_build = time.perf_counter() - t0
            t0 = time.perf_counter()
            sub2 = ix.subset(mask)
            t_build2 = time.perf_counter() - t0
            sub2.close()
            r_sub = timed(sub, lambda: sub.search(q256, 10))
            t0 = time.perf_counter()
            s_ref, i_ref = ix.search(q256[:16], 10, mask=mask)
            t_scan16 = time.perf_counter() - t0
            s_sub, i_sub = sub.search(q256[:16], 10)
            assert np.array_equal(i_ref, i_sub), "subset index and masked scan disagree"
            res["subset_10pct"] = {"rows": sub.n, "build_ms_first": round(t_build * 1e3, 3), "build_ms": round(t_build2 * 1e3, 3),
                                   "search_b256_wall_ms": r_sub["wall_ms"], "masked_scan_b16_wall_ms": round(t_scan16 * 1e3, 3)}
            sub.close()
        # sanity: the counting pass agrees with the search
        s, i = ix.search(q4, 10, algo="scan")
        r, sc = ix.rank_of(q4, i[:, 3])
        assert (r == 3).all() and np.allclose(sc, s[:, 3], atol=0, rtol=0), (r, sc, s[:, 3])
        out[name] = res
        ix.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
