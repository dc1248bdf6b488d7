#!/bin/bash
# Round 3, job 1: the multi-rank entry points on the one-GPU box + the N = 1 line of the new bench.py.
TAG=${1:-r03a}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_spawn.py -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/${TAG}_tests.log
timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_g2.json 2> gpurun_out/${TAG}_bench_g2.err; echo "bench --gpus 2 rc=$?"
cut -c1-600 gpurun_out/${TAG}_bench_g2.json; tail -3 gpurun_out/${TAG}_bench_g2.err
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_g1.json 2> gpurun_out/${TAG}_bench_g1.err; echo "bench --gpus 1 rc=$?"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open(f'gpurun_out/{tag}_bench_g1.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 1), 'M steps/s', d['ms_per_step'], d['roofline'], d.get('roofline_hbm'), d.get('sustained'))
print(d.get('extra_workloads'))
PY
tail -3 gpurun_out/${TAG}_bench_g1.err
