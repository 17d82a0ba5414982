#!/bin/bash
# One box: the time-boxed randomised parity sweep (tests/stress_parity.py), small and --big, on the build in the tree, then
# configs[3]'s corpus (50M x 768 bf16) on ONE GPU.  Usage: bash tools/r04_stress.sh <tag> [seed]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r04stress}"
S="${2:-41}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 330 python3 "$R/tests/stress_parity.py" --seconds 270 --seed $S > "$OUT/stress_small.log" 2>&1 || { echo "stress small FAILED" >&2; tail -5 "$OUT/stress_small.log" >&2; exit 1; }
tail -1 "$OUT/stress_small.log"
timeout -k 10 330 python3 "$R/tests/stress_parity.py" --seconds 240 --seed $((S + 1)) --big > "$OUT/stress_big.log" 2>&1 || { echo "stress big FAILED" >&2; tail -5 "$OUT/stress_big.log" >&2; exit 1; }
tail -1 "$OUT/stress_big.log"
timeout -k 10 500 python3 "$R/bench.py" --workload c4 --no-cpu-baseline --no-ceiling > "$OUT/bench_c4_one_gpu.json" 2> "$OUT/bench_c4_one_gpu.log" || { echo "c4 on one GPU FAILED" >&2; tail -5 "$OUT/bench_c4_one_gpu.log" >&2; exit 1; }
cut -c1-400 "$OUT/bench_c4_one_gpu.json"
