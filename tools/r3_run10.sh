#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3i}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/c5_fused.json" 2> "$OUT/c5_fused.log" || echo "c5 failed" >&2
timeout -k 10 500 python3 "$R/bench.py" --workload c5 --no-cpu-baseline --tune-gemms > "$OUT/c5_tuned.json" 2> "$OUT/c5_tuned.log" || echo "c5 tuned failed" >&2
timeout -k 10 400 python3 "$R/tools/sample_size_ab.py" --rows 10000000 --batches 40 > "$OUT/sample_size_10M.log" 2>&1 || echo "sample size failed" >&2
timeout -k 10 300 python3 "$R/tools/sample_size_ab.py" --rows 1250000 --batches 40 --sizes 4096,2048 > "$OUT/sample_size_1p25M.log" 2>&1 || echo "sample size failed" >&2
timeout -k 10 300 python3 "$R/bench.py" --workload c2 --no-cpu-baseline > "$OUT/c2.json" 2> "$OUT/c2.log"
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "recall", d.get("recall_at_10"))
PY
tail -4 "$OUT"/sample_size_10M.log "$OUT"/sample_size_1p25M.log | cut -c1-400
