#!/bin/bash
# One box: the long-sequence fp32 forward test, then the time-boxed randomised parity sweep (tests/stress_parity.py), small and --big, on the
# build in the tree.  Usage: bash tools/r05_stress.sh <tag> [seed]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r05stress}"
S="${2:-51}"
mkdir -p "$OUT"
cd "$R" && python -m pytest tests/test_mirrors_gpu.py -x -q -k "long_corpus_texts or float_attention" > "$OUT/long_seq_tests.log" 2>&1; echo "long-sequence tests rc=$?"; tail -4 "$OUT/long_seq_tests.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 330 python3 "$R/tests/stress_parity.py" --seconds 270 --seed $S > "$OUT/stress_small.log" 2>&1 || { echo "stress small FAILED" >&2; tail -5 "$OUT/stress_small.log" >&2; exit 1; }
tail -1 "$OUT/stress_small.log"
timeout -k 10 330 python3 "$R/tests/stress_parity.py" --seconds 240 --seed $((S + 1)) --big > "$OUT/stress_big.log" 2>&1 || { echo "stress big FAILED" >&2; tail -5 "$OUT/stress_big.log" >&2; exit 1; }
tail -1 "$OUT/stress_big.log"
