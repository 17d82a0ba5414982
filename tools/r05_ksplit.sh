#!/bin/bash
# round 5: the k-split form of the paired d = 1024 pass (TS_MFMA_PAIR=2) against the NB = 2 form (TS_MFMA_PAIR=1): parity tests, same-box A/B, fetch
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/${1:-r05k}"; mkdir -p "$OUT"
cd "$R"
TS_MFMA_PAIR=2 timeout -k 10 300 python -m pytest tests/test_search_gpu.py -x -q -k "pair or 1024" > "$OUT/pair_tests_ksplit.log" 2>&1; echo "k-split pair tests rc=$?"; tail -4 "$OUT/pair_tests_ksplit.log"
for m in 1 2 1 2; do
  TS_MFMA_PAIR=$m timeout -k 10 300 python3 bench.py --workload c3q --no-cpu-baseline --no-ceiling --steps 40 --warmup 10 > "$OUT/c3q_pair$m.json" 2> "$OUT/c3q_pair$m.log" || { tail -5 "$OUT/c3q_pair$m.log"; exit 1; }
  python3 -c "
import json;d=json.loads(open('$OUT/c3q_pair$m.json').read().strip().splitlines()[-1]);r=d['roofline'];print('pair form $m: c3q q/s',d['value'],'ms/step',d['ms_per_step'],'kernel',r['kernel_ms'],'hbm',r['hbm_frac'],'sustained',d['sustained']['kernel_ms'],'violations',d['parity']['violations'], (r['power'] or {}).get('sclk_mhz'), d['search_stats'])"
done
cd /tmp && export TMPDIR=/tmp
TS_MFMA_PAIR=2 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH_ksplit" -- python3 "$R/bench.py" --workload c3q --steps 6 --warmup 2 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> "$OUT/pmc_ksplit.log"
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/pmc_FETCH_ksplit/**/*counter_collection.csv",recursive=True))[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "mfma16_topk_kernel<1024" in r["Kernel_Name"]]
print("k-split FETCH_SIZE x2, GB per launch:", [round(2*x*1024/1e9,2) for x in v])
PY
