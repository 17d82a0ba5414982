#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3m}"
mkdir -p "$OUT"
cd "$R" && (timeout -k 10 300 python -m pytest tests/test_mirrors_gpu.py -x -q -k "layernorm or fused or encode" > "$OUT/pytest.log" 2>&1; tail -2 "$OUT/pytest.log")
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/c5.json" 2> "$OUT/c5.log" || echo "c5 failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -- python3 "$R/bench.py" --workload c5 --no-cpu-baseline --no-recall --sustained-steps 20 > "$OUT/trace_c5.json" 2> "$OUT/trace_c5.log" || echo "c5 trace failed" >&2
python3 "$R/tools/small_n_latency.py" > "$OUT/small_n.log" 2>&1 || echo "small_n failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob, csv
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "recall", d.get("recall_at_10"))
f = glob.glob(f"{out}/trace_c5/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} per-step {float(r['TotalDurationNs'])/1e6/45:6.3f}")
PY
tail -9 "$OUT/small_n.log" | head -8
