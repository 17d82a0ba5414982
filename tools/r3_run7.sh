#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3f}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200 --no-ceiling"
for i in 1 2; do
  timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_$i.json" 2>> "$OUT/shard.log"
done
timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-ceiling > "$OUT/c3.json" 2> "$OUT/c3.log"
timeout -k 10 300 python3 "$R/bench.py" --workload c2 --no-cpu-baseline > "$OUT/c2.json" 2> "$OUT/c2.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_c3.json" 2> "$OUT/trace_c3.log" || echo "c3 trace failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "recall", d.get("recall_at_10"))
PY
for t in trace_shard trace_c3; do find "$OUT/$t" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-150 | head -9; done
