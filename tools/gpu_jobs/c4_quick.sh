#!/bin/bash
# c4 in both product forms, short: value + the dominant kernels' times (bench breakdown); conv parity tests
for P in fp32 bf16x6; do
timeout 300 python bench.py --workload c4 --products $P --steps 5 --warmup 1 --no-extra --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P', round(d['value']/1e6,4), 'M', round(d['ms_per_step'],1),'ms', {k:round(v['ms_per_step'],2) for k,v in d['kernel_ms_per_step'].items() if v['ms_per_step']>5}, 'frac', round(d['roofline']['frac'],3))"
done
timeout 600 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_cnn_ppo.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
