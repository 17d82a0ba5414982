#!/bin/bash
# One box, un-profiled: the whole 10M x 768 bf16 corpus (no exchange) and its 2 / 4 / 8-way shares with the exchange + merge
# path on (bench.py --rows N --force-dist, world 1).  Writes gpurun_out/<dir>/shard_steps.json.
# Round 4: the shard runs time 1,000 steps behind 100 warm-ups (SHARD_STEPS / SHARD_WARMUP).  Round 3 timed 100 behind 10: a
# 45 ms window that sits inside the power controller's transient - the kernel trace of such a run shows the full pass of a
# 1.25M-row shard at 363 us in the first steps, 557 us eight steps later and 420 us from step 20 on - so its "step - kernel"
# (the kernel time comes from the sustained leg) mixed two clock states.
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-shards}"
mkdir -p "$OUT"
timeout -k 10 300 python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-recall --no-ceiling > "$OUT/steps_10000000.json" 2> "$OUT/steps_10000000.log"
for n in 5000000 2500000 1250000; do
  timeout -k 10 200 python3 "$R/bench.py" --workload c3 --rows $n --force-dist --no-cpu-baseline --no-recall --no-ceiling --steps ${SHARD_STEPS:-1000} --warmup ${SHARD_WARMUP:-100} > "$OUT/steps_$n.json" 2> "$OUT/steps_$n.log"
done
python3 - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
rows = {}
for n in (10000000, 5000000, 2500000, 1250000):
    d = json.loads(open(f"{out}/steps_{n}.json").read().strip().splitlines()[-1])
    rows[str(n)] = {"ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"], "queries_per_s": d["value"],
                    "step_minus_kernel_ms": round(d["ms_per_step"] - d["roofline"]["kernel_ms"], 4),
                    "sustained_ms_per_step": (d.get("sustained") or {}).get("ms_per_step")}
whole = rows["10000000"]["ms_per_step"]
res = {"note": "bench.py on ONE box, un-profiled, same build: the whole 10M x 768 bf16 corpus (no exchange; 20 timed steps with the "
               "kernel timer's brackets, as the default run) and its 2 / 4 / 8-way shares with the exchange + merge path on "
               "(--rows N --force-dist, world 1; 1,000 timed steps behind 100 warm-ups, without brackets; kernel ms from the "
               "bracketed sustained leg behind them): ms per step of 256 queries, full-pass kernel ms, queries/s",
       "rows": rows,
       "step_ratio_vs_whole_corpus": {k: round(whole / v["ms_per_step"], 2) for k, v in rows.items()}}
json.dump(res, open(f"{out}/shard_steps.json", "w"), indent=1)
print(json.dumps(res["step_ratio_vs_whole_corpus"]), json.dumps(rows))
PY
