#!/bin/bash
# round 5, job C: GPU suite + headline / side-workload benches of the current build (+ the deferred-readback A/B)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05c_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r05c_tests.log
b() { local tag=$1; shift; timeout 600 env "$@" python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r05c_bench_$tag.json 2> gpurun_out/r05c_bench_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r05c_bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 4), 'ms frac', round(d['roofline']['frac'], 4), 'launch us', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'sustained', round(d.get('sustained_value', 0) / 1e6, 1))
    print('   ', {k: v['ms_per_step'] for k, v in d['kernel_ms_per_step'].items()})
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
b base PFA_NOISE_PREFETCH=1
b nonoise PFA_NOISE_PREFETCH=0
b nosums PFA_GAE_SUMS=0
b lazy PFA_LAZY_READBACK=1
b base2 PFA_NOISE_PREFETCH=1
for w in "--workload c3 --steps 6 --warmup 2" "--workload c4 --steps 4 --warmup 1" "--hidden 256 --steps 6 --warmup 2"; do
  timeout 600 python bench.py --no-extra --no-cpu-baseline --no-breakdown --sustained-seconds 0 $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], round(d['value']/1e6,3),'M frac',round(d['roofline']['frac'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1))"
done
