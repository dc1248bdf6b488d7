#!/bin/bash
# c4 / c3 kernel work (one gpurun call): the conv and LSTM parity tests, per-product A/B of the built igemm variants, then the two
# workload lines.  Outputs under gpurun_out/<tag>_*.
TAG=${1:-r03c4}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_cnn_ppo.py tests/test_gpu_lstm.py -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/${TAG}_tests.log
timeout 600 python tools/igemm_bench.py run 8192 > gpurun_out/${TAG}_igemm_bench.txt 2>&1; echo "igemm bench rc=$?"
cat gpurun_out/${TAG}_igemm_bench.txt
timeout 300 python bench.py --workload c4 --no-cpu-baseline --no-extra --sustained-seconds 0 > gpurun_out/${TAG}_bench_c4.json 2> gpurun_out/${TAG}_bench_c4.err; echo "c4 rc=$?"
timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-extra --sustained-seconds 0 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "c3 rc=$?"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for w in ('c4', 'c3'):
    try:
        d = json.loads(open(f'gpurun_out/{tag}_bench_{w}.json').read().strip().splitlines()[-1])
        print(w, round(d['value'] / 1e6, 4), 'M steps/s', round(d['ms_per_step'], 3), 'ms', {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'achieved', 'avg_launch_ms')})
    except Exception as e:
        print(w, 'no line', e)
PY
