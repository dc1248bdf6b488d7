#!/bin/bash
# round 5, job G: dX with pixel-uniform tiles — conv tests, then c4 A/B (PFA_IG_DX_TILES=0 / 1), + the tightened-tolerance tests
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_cnn_ppo.py tests/test_gpu_general.py tests/test_gpu_parity_full.py tests/test_gpu_ppo.py -q -s 2>&1 | grep -v "^$" > gpurun_out/r05g_tests.log; echo "tests rc=${PIPESTATUS[0]}"
grep -n "conv update\|passed\|failed\|FAILED\|Mismatch\|Max abs\|Error" gpurun_out/r05g_tests.log | tail -20
for v in 0 1; do
  PFA_IG_DX_TILES=$v timeout 600 python bench.py --no-extra --no-cpu-baseline --sustained-seconds 0 --workload c4 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dx_tiles=$v', round(d['value']/1e6,4),'M frac',round(d['roofline']['frac'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms']*1e3,1), d['kernel_ms_per_step'])"
done
