#!/bin/bash
# round 5, job D: DP / GAE tests of the status exchange + reset, then the c3 / c4 legs with their new full-size self-checks
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_gae.py tests/test_gpu_general.py -q -x > gpurun_out/r05d_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r05d_tests.log
for w in c3 c4; do
  ( time timeout 900 python bench.py --no-extra --no-breakdown --sustained-seconds 0 --workload $w --steps 5 --warmup 1 > gpurun_out/r05d_bench_$w.json 2> gpurun_out/r05d_bench_$w.err ) 2>&1 | grep real
  python - $w <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r05d_bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value'] / 1e6, 3), 'M frac', round(d['roofline']['frac'], 4)); print('  self_check', d.get('self_check')); print('  cpu', json.dumps(d.get('cpu_baseline'))[:400])
except Exception as e:
    print(sys.argv[1], 'failed', e); print(open(f'gpurun_out/r05d_bench_{sys.argv[1]}.err').read()[-3000:])
PY
done
