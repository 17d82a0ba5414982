#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3d}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200 --no-ceiling"
for s in 1 0 1 0; do
  TS_MFMA_SAMPLE=$s timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_sample${s}_$RANDOM.json" 2>> "$OUT/shard.log"
done
TS_MFMA_SAMPLE=1 timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-ceiling > "$OUT/c3_sample1.json" 2> "$OUT/c3_sample1.log"
TS_MFMA_SAMPLE=0 timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-ceiling > "$OUT/c3_sample0.json" 2> "$OUT/c3_sample0.log"
timeout -k 10 300 python3 "$R/bench.py" --workload c1 > "$OUT/c1.json" 2> "$OUT/c1.log" || echo "c1 failed" >&2
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/c5_graph.json" 2> "$OUT/c5_graph.log" || echo "c5 graph failed" >&2
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline --pipeline 2 > "$OUT/c5_graph_p2.json" 2> "$OUT/c5_graph_p2.log" || echo "c5 p2 failed" >&2
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline --encoder-eager > "$OUT/c5_eager.json" 2> "$OUT/c5_eager.log" || echo "c5 eager failed" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f.split("/")[-1], "unreadable", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"), "recall", d.get("recall_at_10"), (d.get("parity") or {}).get("violations"))
PY
find "$OUT/trace_shard" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-160 | head -12
