#!/bin/bash
# Round 4, on the GPU box (gpurun): the bench lines and rocprofv3 passes whose summaries tools/collect_profiles.py copies into
# profiles/.  Usage: bash tools/run_profiles_r04.sh <out-dir under gpurun_out> <part>
#   part a: c3 (bench, kernel trace, FETCH / WRITE passes, SQ counters), c2, c2b
#   part b: c3q (10M x 1024 bf16), c5 at 32 and 128 tokens, c5 with the Qwen3-shaped encoder, c1;  part c: c3q alone;  part d: c5 at 128 tokens and with the Qwen3-shaped encoder
# Trace and counter passes are separate runs (never --pmc together with a trace domain other than kernel-trace).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r04}"
PART="${2:-a}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
bench_and_trace() {   # name, bench args...
  local w="$1"; shift
  timeout -k 10 420 python3 "$R/bench.py" "$@" > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.log" || { echo "bench $w FAILED" >&2; tail -5 "$OUT/bench_$w.log" >&2; return 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-recall --sustained-steps 20 > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || { echo "trace $w FAILED" >&2; return 1; }
  cp "$(ls -t "$OUT"/trace_$w/*/*kernel_stats.csv | head -1)" "$OUT/${w}_kernel_stats.csv"
  echo "$w done: $(cut -c1-200 "$OUT/bench_$w.json")" >&2
}
pmc() {   # name, counters, bench args...
  local w="$1" c="$2"; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_${c%% *}" -- python3 "$R/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> "$OUT/pmc_${w}_${c%% *}.log" || echo "pmc $w $c failed (non-fatal)" >&2
}
if [ "$PART" = "a" ]; then
  bench_and_trace c3 --workload c3 || exit 1
  pmc c3 FETCH_SIZE --workload c3
  pmc c3 WRITE_SIZE --workload c3
  pmc c3 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload c3
  bench_and_trace c2 --workload c2 || exit 1
  pmc c2 FETCH_SIZE --workload c2
  pmc c2 WRITE_SIZE --workload c2
  bench_and_trace c2b --workload c2b || exit 1
elif [ "$PART" = "e" ]; then
  # configs[2] again, the trace over the bench line's own command (300 sustained steps: the average is the steady state's)
  timeout -k 10 420 python3 "$R/bench.py" --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.log" || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$R/bench.py" --workload c3 --no-cpu-baseline --no-recall > "$OUT/trace_c3.json" 2> "$OUT/trace_c3.log" || exit 1
  cp "$(ls -t "$OUT"/trace_c3/*/*kernel_stats.csv | head -1)" "$OUT/c3_kernel_stats.csv"
  head -3 "$OUT/c3_kernel_stats.csv" | cut -c1-160 >&2
elif [ "$PART" = "g" ]; then
  bench_and_trace c5_qwen_128 --workload c5 --encoder qwen --seq-len 128 || exit 1
elif [ "$PART" = "f" ]; then
  bench_and_trace c5_gemma --workload c5 --encoder gemma || exit 1
elif [ "$PART" = "d" ]; then
  bench_and_trace c5_128 --workload c5 --seq-len 128 || exit 1
  bench_and_trace c5_qwen --workload c5 --encoder qwen || exit 1
elif [ "$PART" = "c" ]; then
  bench_and_trace c3q --workload c3q || exit 1
  pmc c3q FETCH_SIZE --workload c3q
  pmc c3q WRITE_SIZE --workload c3q
else
  bench_and_trace c3q --workload c3q || exit 1
  pmc c3q FETCH_SIZE --workload c3q
  pmc c3q WRITE_SIZE --workload c3q
  bench_and_trace c5 --workload c5 || exit 1
  bench_and_trace c5_128 --workload c5 --seq-len 128 || exit 1
  bench_and_trace c5_qwen --workload c5 --encoder qwen || exit 1
  timeout -k 10 300 python3 "$R/bench.py" --workload c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.log" || echo "c1 bench failed (non-fatal)" >&2
fi
cat "$OUT"/bench_*.json | cut -c1-400
