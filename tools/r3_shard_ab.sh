#!/bin/bash
# One box: an eighth of the corpus with the exchange path on - sample size A/B (TS_MFMA_FIRST_ROWS) and a kernel trace.
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3c}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200"
for rep in 1 2; do
for fr in 0 2048 1024; do
  if [ "$fr" = "0" ]; then unset TS_MFMA_FIRST_ROWS; else export TS_MFMA_FIRST_ROWS=$fr; fi
  timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_fr${fr}_$rep.json" 2> "$OUT/shard_fr${fr}_$rep.log"
done
done
unset TS_MFMA_FIRST_ROWS
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/shard_fr*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "sustained", d["sustained"]["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], d["search_stats"])
PY
