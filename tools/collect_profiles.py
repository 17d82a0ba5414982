#!/usr/bin/env python3
"""Copy the judged summaries of one round out of gpurun_out/ into profiles/ and derive
profiles/traffic.json (HBM bytes per launch of the dominant kernel from rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly
half of the bytes of a wide coalesced stream (16 B per lane; global_load and LDS-DMA alike), so the
read side is doubled; WRITE_SIZE is taken as is (KiB)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(paths):
    """gpurun merges every run's files into the same directory: the latest run's file is the one that counts."""
    return sorted(paths, key=os.path.getmtime)[-1:] if paths else []


def pmc_per_launch(d, counter, kernel_substr):
    vals = []
    for f in newest(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals


def main(src, tag):
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    traffic = {}
    tpath = os.path.join(out, "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    notes = []
    kernels = (("c3", "mfma16_topk_kernel<768, 4, 0, false"), ("c2", "scan_kernel"), ("c2b", "mfma16_topk_kernel<768, 2, 0, false, true"),
               ("c3q", "mfma16_topk_kernel<1024, 4, 0, false, false, true, true"))   # the k-split form (round 4 / first half of round 5: <1024, 2, 0, false, false, true>)
    for w, kern in kernels:
        # tools/run_profiles_r04.sh leaves <w>_kernel_stats.csv beside the trace directories; round 3's script only the directories
        stats = newest(glob.glob(os.path.join(src, f"{w}_kernel_stats.csv")) or
                       glob.glob(os.path.join(src, f"trace_{w}", "**", "*kernel_stats.csv"), recursive=True))
        if stats:
            shutil.copy(stats[0], os.path.join(out, f"{tag}_{w}_kernel_stats.csv"))
        fetch = pmc_per_launch(os.path.join(src, f"pmc_{w}_FETCH_SIZE"), "FETCH_SIZE", kern)
        write = pmc_per_launch(os.path.join(src, f"pmc_{w}_WRITE_SIZE"), "WRITE_SIZE", kern)
        if fetch:
            # the full-corpus passes are the launches with the largest fetch: the median of those within a factor 2 of it
            def typical(v):
                big = sorted(x for x in v if x >= 0.5 * max(v))
                return big[len(big) // 2]
            f_kib, w_kib = typical(fetch), (typical(write) if write else 0.0)
            hbm = 2.0 * f_kib * 1024.0 + w_kib * 1024.0
            traffic[f"{w}_n1"] = int(hbm)
            notes.append(f"{w}: kernel {kern}: FETCH_SIZE {f_kib:.0f} KiB (x2 gfx950 correction) + WRITE_SIZE {w_kib:.0f} KiB "
                         f"= {hbm / 1e9:.3f} GB per launch")
    for extra in ("clock_probe.json", "shard_steps.json"):
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), os.path.join(out, f"{tag}_{extra}"))
    for w in ("c5", "c5_128", "c5_qwen", "c5_qwen_128", "c5_gemma"):
        stats = newest(glob.glob(os.path.join(src, f"{w}_kernel_stats.csv")) or
                       glob.glob(os.path.join(src, f"trace_{w}", "**", "*kernel_stats.csv"), recursive=True))
        if stats:
            shutil.copy(stats[0], os.path.join(out, f"{tag}_{w}_kernel_stats.csv"))
    for w in ("c3", "c2", "c2b", "c3q", "c5", "c5_128", "c5_qwen", "c5_qwen_128", "c5_gemma", "c1"):
        b = os.path.join(src, f"bench_{w}.json")
        if os.path.exists(b) and os.path.getsize(b) > 0:
            shutil.copy(b, os.path.join(out, f"{tag}_bench_{w}.json"))
    # round 5: every bench line and kernel-stats summary the run left (c5 lines carry encoder, tokens and dtype in their names)
    for b in glob.glob(os.path.join(src, "bench_*.json")):
        if os.path.getsize(b) > 0:
            shutil.copy(b, os.path.join(out, f"{tag}_{os.path.basename(b)}"))
    for st_ in glob.glob(os.path.join(src, "*_kernel_stats.csv")):
        shutil.copy(st_, os.path.join(out, f"{tag}_{os.path.basename(st_)}"))
    stats = newest(glob.glob(os.path.join(src, "shard_1p25M_kernel_stats.csv")) or
                   glob.glob(os.path.join(src, "trace_shard", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_shard_1p25M_kernel_stats.csv"))
    sq = newest(glob.glob(os.path.join(src, "pmc_c3_SQ_WAVE_CYCLES", "**", "*counter_collection.csv"), recursive=True))
    if sq:
        # sums over the dispatches of the full pass, one line per counter
        tot = {}
        for r in csv.DictReader(open(sq[0])):
            if kernels[0][1] in r["Kernel_Name"]:
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        with open(os.path.join(out, f"{tag}_c3_sq_counters_raw.txt"), "w") as f:
            f.write("rocprofv3 --kernel-trace --pmc SQ_* (its own pass), bench.py --workload c3 --steps 3 --warmup 1: per launch of the full pass\n")
            for k_, v in sorted(tot.items()):
                f.write(f"{k_}: launches {len(v)}, mean {sum(v) / len(v):.6g}, min {min(v):.6g}, max {max(v):.6g}\n")
    json.dump(traffic, open(tpath, "w"), indent=1)
    with open(os.path.join(out, f"{tag}_traffic_notes.txt"), "w") as f:
        f.write("HBM traffic per launch of the dominant kernel, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes),\n"
                "command: python3 bench.py --workload <w> --steps 3 --warmup 1 --no-cpu-baseline --no-recall\n")
        f.write("\n".join(notes) + "\n")
    print("\n".join(notes))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
