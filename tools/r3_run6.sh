#!/bin/bash
set -e
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-r3e}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="--workload c3 --rows 1250000 --force-dist --no-cpu-baseline --no-recall --steps 200 --warmup 20 --sustained-steps 200 --no-ceiling"
for s in 1 0 1 0; do
  TS_MFMA_SAMPLE=$s timeout -k 10 200 python3 "$R/bench.py" $B > "$OUT/shard_sample${s}_$RANDOM.json" 2>> "$OUT/shard.log"
done
for d in 384 512; do
  timeout -k 10 300 python3 "$R/tools/ab_shapes.py" --rows 10000000 --dim $d --nq 256 --rounds 3 --steps 10 --variant shape16:TS_MFMA_SHAPE=16 --variant shape32:TS_MFMA_SHAPE=32 --out "$OUT/ab_bf16_$d.json" > "$OUT/ab_bf16_$d.log" 2>&1 || echo "ab $d failed" >&2
done
timeout -k 10 300 python3 "$R/tools/ab_shapes.py" --rows 2000000 --dim 384 --dtype f32 --nq 128 --rounds 3 --steps 10 --variant mfma:TS_MFMA_F32=16 --variant scan:TS_MFMA_F32=0 --out "$OUT/ab_f32_384.json" > "$OUT/ab_f32_384.log" 2>&1 || echo "ab f32 failed" >&2
python3 - "$OUT" <<'PY'
import json, sys, glob
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/shard_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(f.split("/")[-1], "q/s", d["value"], "ms/step", d["ms_per_step"], "sustained", (d.get("sustained") or {}).get("ms_per_step"), "kernel_ms", r.get("kernel_ms"))
for f in sorted(glob.glob(f"{out}/ab_*.log")):
    print(f.split("/")[-1]); print("".join(open(f).readlines()[-6:]))
PY
