#!/bin/bash
# c3q on one box: the bench line (kernel ms, q/s) and the fabric fetch per launch of the paired pass (rocprofv3 --pmc FETCH_SIZE).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/${1:-c3q}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$R/bench.py" --workload c3q --no-cpu-baseline --no-ceiling > "$OUT/bench_c3q.json" 2> "$OUT/bench_c3q.log" || { tail -5 "$OUT/bench_c3q.log"; exit 1; }
python3 -c "
import json;d=json.loads(open('$OUT/bench_c3q.json').read().strip().splitlines()[-1]);r=d['roofline'];print('c3q q/s',d['value'],'ms/step',d['ms_per_step'],'kernel',r['kernel_ms'],'hbm',r['hbm_frac'],'sustained',d['sustained']['kernel_ms'],'violations',d['parity']['violations'])"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH" -- python3 "$R/bench.py" --workload c3q --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> "$OUT/pmc.log"
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/pmc_FETCH/**/*counter_collection.csv",recursive=True))[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "mfma16_topk_kernel<1024" in r["Kernel_Name"]]
print("FETCH_SIZE x2, GB per launch:", [round(2*x*1024/1e9,2) for x in v])
PY
