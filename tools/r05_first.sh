#!/bin/bash
# round 5, first GPU session: configs[3] at its size, the new pipeline / launch tests, full-depth encoder table, c5 fp32 vs bf16
set -o pipefail
mkdir -p gpurun_out/r05a
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
python -m pytest tests/test_fullsize_gpu.py::test_config3_50m_8_shards -x -q -s > gpurun_out/r05a/c4_test.log 2>&1; echo "c4 test rc=$?"
tail -5 gpurun_out/r05a/c4_test.log
python -m pytest tests/test_api_gpu.py -x -q -k "shards or pipeline or in_flight" > gpurun_out/r05a/api_tests.log 2>&1; echo "api tests rc=$?"
tail -5 gpurun_out/r05a/api_tests.log
python -m pytest tests/test_mirrors_gpu.py -x -q -k "bench" > gpurun_out/r05a/bench_tests.log 2>&1; echo "bench tests rc=$?"
tail -15 gpurun_out/r05a/bench_tests.log
python tests/test_fulldepth_gpu.py > gpurun_out/r05a/fulldepth.txt 2>gpurun_out/r05a/fulldepth.err; echo "fulldepth rc=$?"
cat gpurun_out/r05a/fulldepth.txt
for dt in fp32 bf16; do
  python bench.py --workload c5 --encoder-dtype $dt --seq-len 32 --sustained-steps 100 --no-ceiling > gpurun_out/r05a/c5_bert32_$dt.json 2>gpurun_out/r05a/c5_bert32_$dt.err; echo "c5 $dt rc=$?"
  python -c "import json;d=json.load(open('gpurun_out/r05a/c5_bert32_$dt.json'));print('$dt',d['value'],d['ms_per_step'],d['recall_at_10'],d['cpu_baseline'])"
done
