cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04q
timeout -k 10 300 python3 $R/bench.py --workload c3q --no-cpu-baseline --no-ceiling > $R/gpurun_out/r04q/bench_c3q.json 2> $R/gpurun_out/r04q/bench_c3q.log || { tail -5 $R/gpurun_out/r04q/bench_c3q.log; exit 1; }
python3 -c "
import json;d=json.loads(open('$R/gpurun_out/r04q/bench_c3q.json').read().strip().splitlines()[-1]);r=d['roofline'];print('c3q',d['value'],d['ms_per_step'],r['kernel_ms'],r['hbm_frac'],r.get('power'),d['parity'],d.get('sustained'))"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r04q/pmc_c3q_FETCH_SIZE -- python3 $R/bench.py --workload c3q --steps 3 --warmup 1 --no-cpu-baseline --no-recall --no-ceiling --sustained-steps 0 > /dev/null 2> $R/gpurun_out/r04q/pmc.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/r04q/pmc_c3q_FETCH_SIZE/**/*counter_collection.csv",recursive=True))[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "mfma16_topk_kernel<1024" in r["Kernel_Name"]]
print("FETCH_SIZE x2 GB per launch:", [round(2*x*1024/1e9,2) for x in v])
PY
