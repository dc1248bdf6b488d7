#!/bin/bash
# The GEMM-path side workloads in both product forms (no code change: bench.py --products).
mkdir -p gpurun_out
for F in "--workload c4 --policy lstm --steps 2 --warmup 1" "--hidden 256 --steps 5 --warmup 2"; do
  for P in fp32 bf16x6; do
    timeout 300 python bench.py $F --products $P --no-cpu-baseline --no-extra --no-breakdown --sustained-seconds 0 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], '|', sys.argv[2], round(d['value'] / 1e6, 4), 'M steps/s', round(d['ms_per_step'], 2), 'ms', round(d['roofline']['frac'], 3))" "$F" $P
  done
done
