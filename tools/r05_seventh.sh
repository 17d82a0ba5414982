#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r05g"; mkdir -p "$OUT"
cd "$R"
python -m pytest tests/test_mirrors_gpu.py tests/test_fulldepth_gpu.py -x -q > "$OUT/tests.log" 2>&1; echo "tests rc=$?"; tail -5 "$OUT/tests.log"
for sl in 32 128; do
  timeout -k 10 400 python3 bench.py --workload c5 --encoder bert --encoder-dtype fp32x3 --seq-len $sl --no-cpu-baseline --no-ceiling --steps 10 --warmup 3 --sustained-steps 40 > "$OUT/c5_bert_${sl}_fp32x3.json" 2> "$OUT/c5_bert_${sl}_fp32x3.err" || { echo FAILED; tail -3 "$OUT/c5_bert_${sl}_fp32x3.err"; continue; }
  python3 -c "
import json;d=json.loads(open('$OUT/c5_bert_${sl}_fp32x3.json').read().strip().splitlines()[-1]);print('c5 bert $sl fp32x3: q/s',d['value'],'ms/step',d['ms_per_step'],'pass',d['roofline']['kernel_ms'],'recall',d['recall_at_10'])"
done
