#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3h}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200 --no-ceiling"
for i in 1 2; do
  timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_$i.json" 2>> "$OUT/shard.log"
done
timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-ceiling > "$OUT/c3.json" 2> "$OUT/c3.log"
timeout -k 10 300 python3 "$R/tools/ab_shapes.py" --rows 1000000 --dim 768 --dtype f32 --nq 1 --rounds 5 --steps 100 --variant equal:TS_SCAN_BALANCE=0 --variant weights:TS_SCAN_BALANCE=1 --out "$OUT/ab_scan_f32.json" > "$OUT/ab_scan_f32.log" 2>&1 || echo "ab scan failed" >&2
timeout -k 10 300 python3 "$R/tools/ab_shapes.py" --rows 10000000 --dim 768 --dtype bf16 --nq 4 --rounds 4 --steps 20 --variant equal:TS_SCAN_BALANCE=0 --variant weights:TS_SCAN_BALANCE=1 --out "$OUT/ab_scan_bf16.json" > "$OUT/ab_scan_bf16.log" 2>&1 || echo "ab scan bf16 failed" >&2
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/c5_fused.json" 2> "$OUT/c5_fused.log" || echo "c5 failed" >&2
TS_ENCODER_FUSED=0 timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/c5_plain.json" 2> "$OUT/c5_plain.log" || echo "c5 plain failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -- python3 "$R/bench.py" --workload c5 --no-cpu-baseline --no-recall --sustained-steps 20 > "$OUT/trace_c5.json" 2> "$OUT/trace_c5.log" || echo "c5 trace failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "recall", d.get("recall_at_10"))
    except Exception as e:
        d = json.load(open(f))
        print(f.split("/")[-1], json.dumps(d.get("variants"))[:600])
PY
for t in trace_shard; do find "$OUT/$t" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-150 | grep -E "sample|select|scan|mfma16" ; done
find "$OUT/trace_c5" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-130 | head -14
