#!/bin/bash
# Other-width update kernel (csrc/ppo_wide.hip): compile-time variants A/B (tools/wide_variant_bench.py build ... ran in the build
# container), the wide-path parity tests on the default build, and the side-workload lines at hidden 64 / 256 / 512.
TAG=${1:-r04_wide}
mkdir -p gpurun_out
timeout 600 python tools/wide_variant_bench.py run 64 256 512 > gpurun_out/${TAG}_variants.log 2>&1; echo "variants rc=$?"
grep -h "/h" gpurun_out/${TAG}_variants.log | cut -c1-260
cp gpurun_out/wide_variant_bench.json gpurun_out/${TAG}_variants.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_dp.py -m gpu -x -q -k "wide or hidden or general" > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/${TAG}_tests.log
for H in 64 256 512; do
  timeout 300 python bench.py --hidden $H --steps 10 --warmup 3 --no-extra --no-cpu-baseline --sustained-seconds 0 > gpurun_out/${TAG}_bench_h$H.json 2> gpurun_out/${TAG}_bench_h$H.err; echo "bench h$H rc=$?"
  python - "$TAG" $H <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/{sys.argv[1]}_bench_h{sys.argv[2]}.json').read().strip().splitlines()[-1])
print('hidden', sys.argv[2], round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 3), 'ms', {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'avg_launch_ms')}, d.get('kernel_ms_per_step'))
PY
done
