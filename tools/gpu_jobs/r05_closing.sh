#!/bin/bash
# GPU-box half of the closing job of round 5 (run through tools/gpu_jobs/with_reference.sh, which stages the reference):
# the reference's unmodified demo.py end to end on the final build, then the default bench line and the 25 s sustained leg.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
test -f _refstage/demo.py || { echo "no staged reference"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_demo_reference.py -x -q -rs > gpurun_out/r05_demo_tests.log 2>&1; echo "demo tests rc=$?"
tail -3 gpurun_out/r05_demo_tests.log
( time timeout 900 python bench.py > gpurun_out/r05_bench_mlp.json 2> gpurun_out/r05_bench_mlp.err ) 2> gpurun_out/r05_bench_wall.txt; echo "bench rc=$?"
timeout 300 python bench.py --no-extra --no-cpu-baseline --no-breakdown --sustained-seconds 25 > gpurun_out/r05_bench_mlp_sustained25.json 2> gpurun_out/r05_bench_sus.err; echo "sustained rc=$?"
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r05_bench_driver_flags.json 2> /dev/null; echo "driver-flags rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_mlp.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 4), 'ms; sustained', round(d.get('sustained_value', 0) / 1e6, 1), 'deferred', round(d['deferred_readback']['value'] / 1e6, 1))
print({k: d['roofline'][k] for k in ('frac', 'avg_launch_ms', 'launches', 'bracketed', 'traffic')})
print(d.get('self_check'))
print(d.get('kernel_ms_per_step'))
for w in d.get('extra_workloads', []):
    print((w.get('config', {}).get('workload', '')[:50], w.get('value'), (w.get('roofline') or {}).get('frac')) if 'value' in w else w)
s = json.loads(open('gpurun_out/r05_bench_mlp_sustained25.json').read().strip().splitlines()[-1])
print('sustained25', s.get('sustained'))
g = json.loads(open('gpurun_out/r05_bench_driver_flags.json').read().strip().splitlines()[-1])
print('driver flags (--steps 20 --warmup 5):', round(g['value'] / 1e6, 1), 'M', round(g['ms_per_step'], 4), 'ms', g['roofline']['frac'], g['roofline']['launches'])
PY
tail -3 gpurun_out/r05_bench_wall.txt
