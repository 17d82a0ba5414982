#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel name, the dispatch with the largest value of
the first counter (the full-corpus pass) and all its counters."""
import csv
import glob
import sys
from collections import defaultdict

def main(d):
    import os
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]   # latest run only
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    by = defaultdict(lambda: defaultdict(dict))  # kernel -> dispatch -> counter -> value
    for r in rows:
        by[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in by.items():
        if "mfma" not in k and "scan_kernel" not in k:
            continue
        # pick dispatch with max SQ_WAVE_CYCLES / first counter
        best = max(disp.items(), key=lambda kv: max(kv[1].values()))
        print(k[:60], "dispatches", len(disp))
        for c, v in sorted(best[1].items()):
            print(f"   {c:32s} {v:16.0f}")

if __name__ == "__main__":
    main(sys.argv[1])
