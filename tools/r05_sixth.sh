#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/r05f"; mkdir -p "$OUT"
cd "$R"
python -m pytest tests/test_mirrors_gpu.py -x -q -k "float_attention or split_pieces" > "$OUT/new_kernel_tests.log" 2>&1; echo "new kernel tests rc=$?"; tail -5 "$OUT/new_kernel_tests.log"
python3 tools/attention_float_timing.py > "$OUT/attention_float_timing.jsonl" 2> "$OUT/attention_float_timing.err"; echo "timing rc=$?"; cat "$OUT/attention_float_timing.jsonl"; tail -3 "$OUT/attention_float_timing.err"
python -m pytest tests/test_fulldepth_gpu.py -x -q > "$OUT/fulldepth_tests.log" 2>&1; echo "fulldepth tests rc=$?"; tail -5 "$OUT/fulldepth_tests.log"
cd /tmp && export TMPDIR=/tmp
for cfg in "bert 128 fp32x3" "bert 32 fp32x3" "qwen 32 fp32x3" "gemma 32 fp32x3" "bert 128 fp32"; do
  set -- $cfg
  w="c5_$1_$2_$3"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 "$R/bench.py" --workload c5 --encoder $1 --seq-len $2 --encoder-dtype $3 --steps 10 --warmup 3 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || { echo "trace $w FAILED"; continue; }
  cp "$(ls -t "$OUT"/trace_$w/*/*kernel_stats.csv | head -1)" "$OUT/${w}_kernel_stats.csv"
  python3 - <<PY
import csv,json
rows=list(csv.DictReader(open("$OUT/${w}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
d=json.loads(open("$OUT/trace_$w.json").read().strip().splitlines()[-1])
print("== $w  ms/step", d['ms_per_step'], " kernel time per step ~", round(tot/33/1e6,2))
for r in rows[:10]:
    print(f"  {float(r['Percentage']):6.2f}%  {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:100]}")
PY
done
