#!/bin/bash
# One box: the step of bench.py with one handle on one stream (--pipeline 1) against two handles in flight (--pipeline 2),
# on the whole 10M x 768 bf16 corpus and on an eighth of it with the exchange + merge path on.
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3_pipeline}"
mkdir -p "$OUT"
for p in 1 2; do
  timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --pipeline $p > "$OUT/c3_p$p.json" 2> "$OUT/c3_p$p.log"
  timeout -k 10 200 python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --pipeline $p > "$OUT/shard_p$p.json" 2> "$OUT/shard_p$p.log"
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for name in ("c3_p1", "c3_p2", "shard_p1", "shard_p2"):
    d = json.loads(open(f"{out}/{name}.json").read().strip().splitlines()[-1])
    print(name, "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "q/s", d["value"], "recall", d.get("recall_at_10"))
PY
