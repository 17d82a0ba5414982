#!/bin/bash
# One box: the search / API GPU tests, the shard steps (tools/shard_steps.sh) and the kernel trace of a 1.25M-row shard step.
# Usage: bash tools/r04_shard_check.sh <tag> [pytest args]
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r04a}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
if [ -n "$2" ]; then
  timeout -k 10 600 python3 -m pytest $2 -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; echo "pytest FAILED" >&2; exit 1; }
  tail -2 "$OUT/pytest.log"
fi
cd /tmp && export TMPDIR=/tmp
bash "$R/tools/shard_steps.sh" "$TAG" > "$OUT/shard_steps.log" 2>&1 || { tail -20 "$OUT/shard_steps.log"; echo "shard steps FAILED" >&2; exit 1; }
tail -1 "$OUT/shard_steps.log" | cut -c1-1500
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_shard" -- python3 "$R/bench.py" --workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 20 > "$OUT/trace_shard.json" 2> "$OUT/trace_shard.log" || { echo "shard trace FAILED" >&2; exit 1; }
f=$(ls -t "$OUT"/trace_shard/*/*kernel_stats.csv | head -1)
cp "$f" "$OUT/shard_1p25M_kernel_stats.csv"
cut -d, -f1-4,6,7 "$OUT/shard_1p25M_kernel_stats.csv" | head -14
