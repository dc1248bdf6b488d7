#!/bin/bash
# Round 6, data-parallel readiness on the one-GPU box: the DP tests, bench.py --gpus 2 / 8 on the shared device (affinity + peer-wait in the
# `dist` block), and the jitter sweep at the BASELINE shape.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_spawn.py -x -q > gpurun_out/r06_dp_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r06_dp_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r06_bench_g1.json 2> gpurun_out/r06_bench_g1.err; echo "g1 rc=$?"
cut -c1-300 gpurun_out/r06_bench_g1.json
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r06_bench_g2_shared_device.json 2> gpurun_out/r06_bench_g2.err; echo "g2 rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_g2_shared_device.json') if l.startswith('{')][0])
print('g2 value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps({k: d['dist'][k] for k in ('affinity', 'rank_ms_per_step', 'peer_wait')}))
PY
timeout 600 python bench.py --gpus 8 --steps 10 --warmup 3 --no-extra --no-cpu-baseline --sustained-seconds 0 --no-transport-ab > gpurun_out/r06_bench_g8_shared_device.json 2> gpurun_out/r06_bench_g8.err; echo "g8 rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_g8_shared_device.json') if l.startswith('{')][0])
print('g8 value', d['value'], 'ms', d['ms_per_step'])
print(json.dumps({k: d['dist'][k] for k in ('affinity', 'rank_ms_per_step', 'peer_wait')}))
PY
timeout 600 python tools/dp_jitter.py --world 2 --envs 4096 --horizon 128 --iters 20 --skews 0,50,100,200,500,1000 --out gpurun_out/r06_dp_jitter.json > gpurun_out/r06_dp_jitter.log 2>&1; echo "jitter rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_dp_jitter.json'))
for l in d['legs']:
    print(l['injected_skew_us'], round(l['ms_per_step'], 4), l['added_ms_per_step'], l['wait_us_per_step'], l['small_exchange_wait_us'], l['grad_exchange_wait_us'])
PY
lscpu | grep -i "numa\|model name\|^CPU(s)" ; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head
