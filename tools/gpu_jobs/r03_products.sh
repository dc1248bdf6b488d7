#!/bin/bash
# The two product forms of the rows kernels (csrc/igemm.hip) on the conv workload: parity tests in both forms, then the c4 line in both.
mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_cnn.py tests/test_gpu_cnn_ppo.py tests/test_gpu_general.py -q -x > gpurun_out/products_tests.log 2>&1; echo "tests rc=$?"
grep -v "^\[W\|amdgpu.ids\|RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|^$\|Gloo" gpurun_out/products_tests.log | tail -4
for P in fp32 bf16x6 fp32 bf16x6; do
  timeout 300 python bench.py --workload c4 --products $P --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-breakdown --sustained-seconds 0 > gpurun_out/products_c4_$P.json 2>/dev/null
  python - $P <<'PY'
import json, sys
P = sys.argv[1]
d = json.loads(open(f'gpurun_out/products_c4_{P}.json').read().strip().splitlines()[-1])
print('c4', P, round(d['value'] / 1e6, 4), 'M steps/s', round(d['ms_per_step'], 1), 'ms', 'rows-form', round(d['roofline']['achieved'], 1), 'TF/s', round(d['roofline']['frac'], 3), '|', d['dtype'][:40])
PY
done
