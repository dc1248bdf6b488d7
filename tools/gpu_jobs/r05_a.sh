#!/bin/bash
# round 5, job A: demo end-to-end (needs the staged reference) + the GPU suite with durations + host memory
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
free -g | head -2
timeout 900 python -m pytest tests/test_gpu_demo_reference.py -x -q -rs > gpurun_out/r05_demo_tests.log 2>&1; echo "demo tests rc=$?"
tail -5 gpurun_out/r05_demo_tests.log; tail -c 6000 gpurun_out/r05_demo_process_output.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 --deselect tests/test_gpu_demo_reference.py > gpurun_out/r05_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_tests.log
timeout 600 python bench.py --no-extra > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 1), 'M', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print(d['kernel_ms_per_step'])
print(json.dumps(d['cpu_baseline'])[:1500])
PY
