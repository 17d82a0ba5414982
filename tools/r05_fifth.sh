#!/bin/bash
# round 5, fifth GPU session: the new encoder kernels (attention_float, pieces producers), full-depth table, c5 matrix, then the whole GPU suite
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r05e"; mkdir -p "$OUT"
cd "$R"
python -m pytest tests/test_mirrors_gpu.py -x -q -k "float_attention or split_pieces" > "$OUT/new_kernel_tests.log" 2>&1; echo "new kernel tests rc=$?"; tail -15 "$OUT/new_kernel_tests.log"
python3 tests/test_fulldepth_gpu.py > "$OUT/fulldepth.txt" 2> "$OUT/fulldepth.err"; echo "fulldepth rc=$?"; cat "$OUT/fulldepth.txt"; tail -5 "$OUT/fulldepth.err"
for enc in bert qwen gemma; do for sl in 32 128; do for dt in fp32 fp32x3; do
  timeout -k 10 400 python3 bench.py --workload c5 --encoder $enc --encoder-dtype $dt --seq-len $sl --no-cpu-baseline --no-ceiling --steps 10 --warmup 3 --sustained-steps 40 > "$OUT/c5_${enc}_${sl}_${dt}.json" 2> "$OUT/c5_${enc}_${sl}_${dt}.err" || { echo "c5 $enc $sl $dt FAILED"; tail -3 "$OUT/c5_${enc}_${sl}_${dt}.err"; continue; }
  python3 -c "
import json;d=json.loads(open('$OUT/c5_${enc}_${sl}_${dt}.json').read().strip().splitlines()[-1]);print('c5 $enc $sl $dt: q/s',d['value'],'ms/step',d['ms_per_step'],'pass',d['roofline']['kernel_ms'],'recall',d['recall_at_10'])"
done; done; done
( time python -m pytest tests -m gpu -x -q --durations=12 ) > "$OUT/gpu_suite.log" 2>&1; echo "gpu suite rc=$?"; tail -22 "$OUT/gpu_suite.log"
