#!/bin/bash
# round 5, third GPU session: pair pacing with L2-scope ops (A/B), split-GEMM probe, c5 matrix fp32 / bf16 x three encoders x 32 / 128 tokens
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r05c"; mkdir -p "$OUT"
cd "$R"
python3 tools/split_gemm_probe.py > "$OUT/split_gemm_probe.jsonl" 2> "$OUT/split_gemm_probe.err"; echo "probe rc=$?"; cat "$OUT/split_gemm_probe.jsonl"; tail -3 "$OUT/split_gemm_probe.err"
python -m pytest tests/test_search_gpu.py -x -q -k "pair or 1024" > "$OUT/pair_tests.log" 2>&1; echo "pair tests rc=$?"; tail -3 "$OUT/pair_tests.log"
for lag in 0 1 0 1; do
  TS_MFMA_PAIR_LAG=$lag timeout -k 10 300 python3 bench.py --workload c3q --no-cpu-baseline --no-ceiling --steps 40 --warmup 10 > "$OUT/c3q_lag$lag.json" 2> "$OUT/c3q_lag$lag.log" || { tail -5 "$OUT/c3q_lag$lag.log"; exit 1; }
  python3 -c "
import json;d=json.loads(open('$OUT/c3q_lag$lag.json').read().strip().splitlines()[-1]);r=d['roofline'];print('lag $lag: c3q q/s',d['value'],'ms/step',d['ms_per_step'],'kernel',r['kernel_ms'],'hbm',r['hbm_frac'],'sustained',d['sustained']['kernel_ms'],'violations',d['parity']['violations'], r['power'])"
done
cd /tmp && export TMPDIR=/tmp
for lag in 1; do
  TS_MFMA_PAIR_LAG=$lag timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH_lag$lag" -- python3 "$R/bench.py" --workload c3q --steps 6 --warmup 2 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> "$OUT/pmc_lag$lag.log"
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/pmc_FETCH_lag$lag/**/*counter_collection.csv",recursive=True))[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "mfma16_topk_kernel<1024" in r["Kernel_Name"]]
print("lag $lag FETCH_SIZE x2, GB per launch:", [round(2*x*1024/1e9,2) for x in v])
PY
done
cd "$R"
for enc in bert qwen gemma; do for sl in 32 128; do for dt in fp32 bf16; do
  timeout -k 10 400 python3 bench.py --workload c5 --encoder $enc --encoder-dtype $dt --seq-len $sl --no-cpu-baseline --no-ceiling --steps 10 --warmup 3 --sustained-steps 40 > "$OUT/c5_${enc}_${sl}_${dt}.json" 2> "$OUT/c5_${enc}_${sl}_${dt}.err" || { echo "c5 $enc $sl $dt FAILED"; tail -3 "$OUT/c5_${enc}_${sl}_${dt}.err"; continue; }
  python3 -c "
import json;d=json.loads(open('$OUT/c5_${enc}_${sl}_${dt}.json').read().strip().splitlines()[-1]);print('c5 $enc $sl $dt: q/s',d['value'],'ms/step',d['ms_per_step'],'pass',d['roofline']['kernel_ms'],'recall',d['recall_at_10'])"
done; done; done
