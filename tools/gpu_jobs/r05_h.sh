#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05h_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r05h_tests.log
b() { local tag=$1; shift; timeout 600 env "$@" python bench.py --no-extra --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r05h_bench_$tag.json 2> gpurun_out/r05h_bench_$tag.err; python - $tag <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/r05h_bench_{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 4), 'ms frac', round(d['roofline']['frac'], 4), 'launch us', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'sustained', round(d.get('sustained_value', 0) / 1e6, 1), 'deferred', round(d['deferred_readback']['value'] / 1e6, 1))
    print('   ', {k: v['ms_per_step'] for k, v in d['kernel_ms_per_step'].items()})
except Exception as e:
    print(sys.argv[1], 'failed', e); print(open(f'gpurun_out/r05h_bench_{sys.argv[1]}.err').read()[-1500:])
PY
}
b direct PFA_DIRECT_READBACK=1
b copy PFA_DIRECT_READBACK=0
b direct2 PFA_DIRECT_READBACK=1
