#!/bin/bash
# Runs on the GPU box (gpurun): the bench lines and rocprofv3 passes whose summaries tools/collect_profiles.py
# copies into profiles/.  Usage: bash tools/run_profiles.sh <out-dir under gpurun_out> [workloads]
# Trace and counter passes are separate runs (never --pmc together with a trace domain other than kernel-trace).
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r03}"
WL="${2:-c3 c2 c2b}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  timeout -k 10 400 python3 "$R/bench.py" --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.log"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -- python3 "$R/bench.py" --workload $w --no-cpu-baseline --no-recall > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_${w}_$c" -- python3 "$R/bench.py" --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-recall > /dev/null 2> "$OUT/pmc_${w}_$c.log"
  done
  echo "$w done" >&2
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_c3_SQ" -- python3 "$R/bench.py" --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-recall > /dev/null 2> "$OUT/pmc_c3_SQ.log" || echo "SQ pass failed (non-fatal)" >&2
# encoder-in-loop: bench line + kernel trace
timeout -k 10 400 python3 "$R/bench.py" --workload c5 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.log" || echo "c5 bench failed (non-fatal)" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c5" -- python3 "$R/bench.py" --workload c5 --no-cpu-baseline --no-recall --sustained-steps 20 > "$OUT/trace_c5.json" 2> "$OUT/trace_c5.log" || echo "c5 trace failed (non-fatal)" >&2
timeout -k 10 300 python3 "$R/bench.py" --workload c1 > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.log" || echo "c1 bench failed (non-fatal)" >&2
# in-kernel clock of the full pass (diagnostic build of the library): >= 2 s of back-to-back launches, then the probe
timeout -k 10 300 python3 "$R/tools/clock_probe.py" > "$OUT/clock_probe.json" 2> "$OUT/clock_probe.log" || echo "clock probe failed (non-fatal)" >&2
# one shard of an 8-way split with the exchange + merge path on: what does not shrink with the shard
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed (non-fatal)" >&2
cat "$OUT"/bench_*.json
