#!/bin/bash
# Same box, same build: bench.py on one 1.25M-row shard of configs[2] with the exchange + merge path on (or with AB_ARGS = other
# bench.py arguments), once per environment setting given ("VAR=value[,VAR=value]" each; "-" = nothing set), `reps` rounds
# interleaved.  Prints ms per step of each run.
# Usage: bash tools/ab_env.sh <tag> <reps> <setting> [<setting> ...]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="$1"; REPS="$2"; shift 2
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for r in $(seq 1 "$REPS"); do
  for s in "$@"; do
    envs=""
    [ "$s" != "-" ] && envs="$(echo "$s" | tr ',' ' ')"
    args="${AB_ARGS:---workload c3 --rows ${AB_ROWS:-1250000} --force-dist --steps ${AB_STEPS:-2000} --warmup 200 --sustained-steps 300}"
    line=$(env $envs timeout -k 10 300 python3 "$R/bench.py" $args --no-cpu-baseline --no-recall --no-ceiling 2> "$OUT/ab.log" | tail -1)
    echo "$line" >> "$OUT/ab_${s//[^A-Za-z0-9_=,-]/_}.jsonl"
    python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('round $r', sys.argv[2], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'step-kernel us', round(1e3*(d['ms_per_step']-d['roofline']['kernel_ms']),1))" "$line" "$s"
  done
done
