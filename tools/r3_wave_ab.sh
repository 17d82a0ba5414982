#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3n}"
mkdir -p "$OUT"
cd "$R" && (timeout -k 10 600 python -m pytest tests/test_search_gpu.py tests/test_api_gpu.py -x -q > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log")
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200 --no-ceiling"
for s in 1 0 1 0; do
  TS_MFMA_WAVE_SELECT=$s timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_w${s}_$RANDOM.json" 2>> "$OUT/shard.log"
done
for s in 1 0; do
  TS_MFMA_WAVE_SELECT=$s timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-ceiling --no-recall --steps 100 > "$OUT/c3_w$s.json" 2> "$OUT/c3_w$s.log"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_c3.json" 2> "$OUT/trace_c3.log" || echo "c3 trace failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"))
PY
for t in trace_shard trace_c3; do find "$OUT/$t" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-150 | grep -E "sample|select|scan" ; done
