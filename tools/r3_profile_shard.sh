#!/bin/bash
# One box: (1) bench line of configs[2] with every leg on; (2) kernel trace of an eighth of the corpus with the exchange
# path on; (3) the standalone ceiling legs behind 3 s of back-to-back product-like launches, rocm-smi sampled meanwhile.
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3b}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( time timeout -k 10 500 python3 "$R/bench.py" --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.log" ) 2> "$OUT/bench_c3.time"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --sustained-steps 0 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || echo "shard trace failed" >&2
( for i in $(seq 1 40); do date +%s.%N; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" ; sleep 0.25; done > "$OUT/power_clock_ceiling.txt" ) &
SMI=$!
timeout -k 10 200 "$R/tools/microbench/build/mfma_stream_ceiling" 10000000 5 20 4 > "$OUT/ceiling.json" 2> "$OUT/ceiling.log" || echo "ceiling failed" >&2
wait $SMI || true
cat "$OUT/bench_c3.json" "$OUT/ceiling.json"; cat "$OUT/bench_c3.time"
find "$OUT/trace_shard" -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-200 | head -14
