#!/bin/bash
# rollout / recurrent kernels after a tile-code change: headline and c3 kernel times, the parity tests that pin them
timeout 200 python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mlp', round(d['value']/1e6,1), round(d['ms_per_step'],4), {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items() if v['ms_per_step']>0})"
timeout 200 python bench.py --workload c3 --steps 8 --warmup 2 --no-extra --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', round(d['value']/1e6,2), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['kernel_ms_per_step'].items() if v['ms_per_step']>0.2})"
timeout 200 python bench.py --hidden 256 --steps 8 --warmup 2 --no-extra --no-cpu-baseline --sustained-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('h256', round(d['value']/1e6,2), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['kernel_ms_per_step'].items() if v['ms_per_step']>0.1})"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
