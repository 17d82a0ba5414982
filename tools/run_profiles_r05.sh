#!/bin/bash
# Round 5, on the GPU box (gpurun): the bench lines and rocprofv3 passes whose summaries tools/collect_profiles.py copies into
# profiles/.  Usage: bash tools/run_profiles_r05.sh <out-dir under gpurun_out> <part>
#   part a: c3 (bench line; kernel trace of the SAME command; FETCH / WRITE / SQ counter passes), c2 (+ passes), c2b
#   part b: c3q (bench, trace, FETCH / WRITE), c4 on one GPU, c1
#   part c_bert | c_qwen | c_gemma: c5 - that encoder x {32, 128} tokens x {fp32, fp32x3, bf16}: bench lines
#   part c_trace: kernel traces of the BERT lines and of the Qwen3 / Gemma3 fp32x3 lines at 32 tokens
# Trace and counter passes are separate runs (never --pmc together with a trace domain other than kernel-trace).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r05}"
PART="${2:-a}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
bench_only() {   # name, bench args...
  local w="$1"; shift
  timeout -k 10 420 python3 "$R/bench.py" "$@" > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.log" || { echo "bench $w FAILED" >&2; tail -5 "$OUT/bench_$w.log" >&2; return 1; }
  python3 -c "
import json;d=json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1]);r=d.get('roofline') or {};print('$w:',d['value'],d['unit'],'ms/step',d['ms_per_step'],'kernel',r.get('kernel_ms'),'frac',r.get('frac'),'violations',(d.get('parity') or {}).get('violations'))" >&2
}
trace() {   # name, bench args... (the same command as the bench line, minus the host-side legs)
  local w="$1"; shift
  timeout -k 10 360 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-recall > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || { echo "trace $w FAILED" >&2; return 1; }
  cp "$(ls -t "$OUT"/trace_$w/*/*kernel_stats.csv | head -1)" "$OUT/${w}_kernel_stats.csv"
  head -4 "$OUT/${w}_kernel_stats.csv" | cut -c1-170 >&2
}
pmc() {   # name, counters, bench args...
  local w="$1" c="$2"; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_${c%% *}" -- python3 "$R/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> "$OUT/pmc_${w}_${c%% *}.log" || echo "pmc $w $c failed (non-fatal)" >&2
}
if [ "$PART" = "a" ]; then
  bench_only c3 --workload c3 || exit 1
  trace c3 --workload c3 || exit 1
  pmc c3 FETCH_SIZE --workload c3
  pmc c3 WRITE_SIZE --workload c3
  pmc c3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c3
  bench_only c2 --workload c2 || exit 1
  trace c2 --workload c2 || exit 1
  pmc c2 FETCH_SIZE --workload c2
  pmc c2 WRITE_SIZE --workload c2
  bench_only c2b --workload c2b || exit 1
  trace c2b --workload c2b || exit 1
elif [ "$PART" = "b" ]; then
  bench_only c3q --workload c3q || exit 1
  trace c3q --workload c3q || exit 1
  pmc c3q FETCH_SIZE --workload c3q
  pmc c3q WRITE_SIZE --workload c3q
  bench_only c4_one_gpu --workload c4 --no-cpu-baseline --no-ceiling || echo "c4 failed (non-fatal)" >&2
  bench_only c1 --workload c1 || echo "c1 failed (non-fatal)" >&2
elif [ "$PART" = "c_trace" ]; then
  for sl in 32 128; do for dt in fp32 fp32x3 bf16; do
    trace "c5_bert_${sl}_${dt}" --workload c5 --encoder bert --seq-len $sl --encoder-dtype $dt --no-ceiling --sustained-steps 40 --steps 10 --warmup 3
  done; done
  trace c5_qwen_32_fp32x3 --workload c5 --encoder qwen --seq-len 32 --encoder-dtype fp32x3 --no-ceiling --sustained-steps 40 --steps 10 --warmup 3
  trace c5_gemma_32_fp32x3 --workload c5 --encoder gemma --seq-len 32 --encoder-dtype fp32x3 --no-ceiling --sustained-steps 40 --steps 10 --warmup 3
else
  # part c_bert / c_qwen / c_gemma: the c5 lines of one encoder
  for enc in ${PART#c_}; do for sl in 32 128; do for dt in fp32 fp32x3 bf16; do
    extra="--no-cpu-baseline"
    [ "$enc" = "bert" ] && [ "$sl" = "32" ] && extra=""                 # the BERT lines at 32 tokens carry the CPU baseline (host encode + search)
    bench_only "c5_${enc}_${sl}_${dt}" --workload c5 --encoder $enc --seq-len $sl --encoder-dtype $dt --no-ceiling --sustained-steps 100 $extra
  done; done; done
fi
