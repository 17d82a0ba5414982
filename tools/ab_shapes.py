#!/usr/bin/env python3
"""A/B of kernel options on ONE corpus in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): the full-pass
kernel time (hipEvent brackets inside the library) and the whole-search time per batch for each named variant.

    python tools/ab_shapes.py --rows 10000000 --nq 256 --rounds 5 --steps 20 \
        --variant shape16:TS_MFMA_SHAPE=16 --variant shape32:TS_MFMA_SHAPE=32
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
# the timing-only kernel variants live in the diagnostic build of the library (make -C theoremsearch_amd/csrc diag)
_DIAG = os.path.join(ROOT, "theoremsearch_amd", "libtsearch_diag.so")
if os.path.exists(_DIAG):
    os.environ.setdefault("TS_LIB", _DIAG)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variant", action="append", default=[], help="name:KNOB=value[,KNOB=value]")
    ap.add_argument("--out", default="")
    ap.add_argument("--data", default="normal", choices=["normal", "zeros", "same"],
                    help="corpus content: the synthetic normal rows; all-zero rows; one normal row repeated (timing of the "
                         "matrix pipe's data-dependent power: use with a timing-only variant, TS_MFMA_VARIANT=1)")
    args = ap.parse_args()
    import torch
    import synthetic
    import theoremsearch_amd as ts
    bf16 = args.dtype == "bf16"
    ch = synthetic.CHUNK_ROWS
    ix = ts.TheoremIndex(args.rows, args.dim, dtype=args.dtype, metric="ip")

    def make(c):
        data = synthetic.synth_chunk(c, ch, args.dim, bf16=bf16)
        if args.data == "zeros":
            data = np.zeros_like(data)
        elif args.data == "same":
            data = np.ascontiguousarray(np.broadcast_to(data[:1], data.shape))
        ix.upload(data[: min(args.rows, (c + 1) * ch) - c * ch], c * ch)

    with ThreadPoolExecutor(16) as ex:
        list(ex.map(make, range((args.rows + ch - 1) // ch)))
    q = synthetic.synth_queries(0, args.nq, args.dim, bf16=bf16)
    qd = torch.from_numpy(q.view(np.int16) if bf16 else q).cuda()
    out_s = torch.empty((args.nq, 10), dtype=torch.float32, device="cuda")
    out_i = torch.empty((args.nq, 10), dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    variants = []
    for v in args.variant or ["default:"]:
        name, _, spec = v.partition(":")
        knobs = dict(kv.split("=") for kv in spec.split(",") if kv)
        variants.append((name, {k: int(x) for k, x in knobs.items()}))
    all_knobs = sorted({k for _, kn in variants for k in kn})
    res = {name: {"kernel_ms": [], "step_ms": []} for name, _ in variants}
    print("variants:", variants, flush=True)
    ref = None
    for rnd in range(args.rounds + 1):                       # round 0 = warm-up
        for name, kn in variants:
            for k in all_knobs:
                ix.set_option(k, kn.get(k))
            for _ in range(3):
                ix.search_device(qd.data_ptr(), args.dtype, args.nq, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            ix.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ix.search_device(qd.data_ptr(), args.dtype, args.nq, 10, out_s.data_ptr(), out_i.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            p = ix.profile_read()
            ix.profile_enable(False)
            got = out_i.cpu().numpy()
            if rnd == 0:
                _, _, st_ = ix.search(q, 10, return_stats=True)
                print(f"   {name}: levels {st_['levels']}, candidates per query {st_['candidates'] / max(1, args.nq):.1f}, "
                      f"queries re-run exactly {st_['fallback_queries']}", flush=True)
            if ref is None:
                ref = got
            same = bool(np.array_equal(ref, got))
            pr = ix.probe_read() if kn.get("TS_MFMA_VARIANT") == 3 else None
            if pr and pr["ghz"] > 0:
                res[name]["probe"] = pr
                print(f"   clock probe: {pr}", flush=True)
            if rnd:
                res[name]["kernel_ms"].append(p["total_ms"] / max(1, p["launches"]))
                res[name]["step_ms"].append(dt / args.steps * 1e3)
            print(f"round {rnd} {name}: kernel {p['total_ms'] / max(1, p['launches']):.4f} ms x {p['launches'] / args.steps:.0f}/step, "
                  f"step {dt / args.steps * 1e3:.4f} ms, same ids as first variant: {same}", flush=True)
    summary = {}
    for name, r in res.items():
        km, sm = np.array(r["kernel_ms"]), np.array(r["step_ms"])
        summary[name] = {"kernel_ms_median": float(np.median(km)), "kernel_ms_min": float(km.min()),
                         "step_ms_median": float(np.median(sm)), "step_ms_min": float(sm.min())}
        if "probe" in r:
            summary[name]["probe"] = r["probe"]
    print(json.dumps({"rows": args.rows, "nq": args.nq, "dim": args.dim, "dtype": args.dtype, "rounds": args.rounds,
                      "steps": args.steps, "variants": summary}, indent=1))
    if args.out:
        json.dump({"rows": args.rows, "nq": args.nq, "dim": args.dim, "dtype": args.dtype, "rounds": args.rounds,
                   "steps": args.steps, "variants": summary, "raw": res}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
