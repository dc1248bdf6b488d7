#!/bin/bash
# round 5, job F: exact GAE — the whole GPU suite, then the headline bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r05f_tests.log; echo "tests rc=${PIPESTATUS[0]}"
grep -n "conv update\|passed\|failed\|FAILED\|Mismatch\|Max abs\|Error" gpurun_out/r05f_tests.log | tail -30
timeout 600 python bench.py --no-extra --no-cpu-baseline --self-check --steps 30 --warmup 5 > gpurun_out/r05f_bench.json 2> gpurun_out/r05f_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05f_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M', round(d['ms_per_step'], 4), 'ms frac', round(d['roofline']['frac'], 4), 'sustained', round(d.get('sustained_value', 0) / 1e6, 1))
print({k: v['ms_per_step'] for k, v in d['kernel_ms_per_step'].items()}); print(d.get('self_check'))
PY
