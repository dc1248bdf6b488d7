#!/bin/bash
# c3 (LSTM 128 on 160-float rows, 4096 envs): bench line + per-kernel event breakdown
mkdir -p gpurun_out
TAG=${1:-c3}
timeout 500 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --sustained-seconds 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.loads([l for l in open('gpurun_out/${TAG}_bench.json') if l.startswith('{')][0])
print('value', d['value'], 'ms', d['ms_per_step'])
for k in d:
    if 'break' in k or 'kernel' in k:
        print(k, json.dumps(d[k])[:2500])
PY
