#!/bin/bash
# round 5, fourth GPU session: the whole GPU suite, the full-depth table with fp32x3, c5 with fp32x3, kernel stats of the fp32 / fp32x3 forwards
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r05d"; mkdir -p "$OUT"
cd "$R"
( time python -m pytest tests -m gpu -x -q --durations=15 ) > "$OUT/gpu_suite.log" 2>&1; echo "gpu suite rc=$?"; tail -25 "$OUT/gpu_suite.log"
python3 tests/test_fulldepth_gpu.py > "$OUT/fulldepth.txt" 2> "$OUT/fulldepth.err"; echo "fulldepth rc=$?"; cat "$OUT/fulldepth.txt"
for enc in bert qwen gemma; do for sl in 32 128; do
  timeout -k 10 400 python3 bench.py --workload c5 --encoder $enc --encoder-dtype fp32x3 --seq-len $sl --no-cpu-baseline --no-ceiling --steps 10 --warmup 3 --sustained-steps 40 > "$OUT/c5_${enc}_${sl}_fp32x3.json" 2> "$OUT/c5_${enc}_${sl}_fp32x3.err" || { echo "c5 $enc $sl fp32x3 FAILED"; tail -3 "$OUT/c5_${enc}_${sl}_fp32x3.err"; continue; }
  python3 -c "
import json;d=json.loads(open('$OUT/c5_${enc}_${sl}_fp32x3.json').read().strip().splitlines()[-1]);print('c5 $enc $sl fp32x3: q/s',d['value'],'ms/step',d['ms_per_step'],'pass',d['roofline']['kernel_ms'],'recall',d['recall_at_10'])"
done; done
cd /tmp && export TMPDIR=/tmp
for cfg in "bert 128 fp32" "bert 128 fp32x3" "bert 32 fp32x3" "qwen 32 fp32" "qwen 32 fp32x3"; do
  set -- $cfg
  w="c5_$1_$2_$3"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 "$R/bench.py" --workload c5 --encoder $1 --seq-len $2 --encoder-dtype $3 --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || { echo "trace $w FAILED"; continue; }
  cp "$(ls -t "$OUT"/trace_$w/*/*kernel_stats.csv | head -1)" "$OUT/${w}_kernel_stats.csv"
  echo "== $w"; head -12 "$OUT/${w}_kernel_stats.csv" | cut -c1-150
done
