#!/bin/bash
# One box: shard steps (whole corpus and its 2 / 4 / 8-way shares), the randomised parity sweep (small and big), c4's corpus
# on one GPU.
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-checks}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
bash "$R/tools/shard_steps.sh" "$(basename "$OUT")" > "$OUT/shard_steps.log" 2>&1 || echo "shard steps failed" >&2
timeout -k 10 260 python3 "$R/tests/stress_parity.py" --seconds 200 --seed ${2:-31} > "$OUT/stress_small.log" 2>&1 || echo "stress small FAILED" >&2
timeout -k 10 260 python3 "$R/tests/stress_parity.py" --seconds 150 --seed ${3:-32} --big > "$OUT/stress_big.log" 2>&1 || echo "stress big FAILED" >&2
tail -2 "$OUT/stress_small.log" "$OUT/stress_big.log" 2>/dev/null | cut -c1-200
cat "$OUT/shard_steps.json" 2>/dev/null | head -60
