# KS=13 rollout forward + batched stats kernel: tests, headline bench, c3 / c4 lines of the same build
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t6_tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed" gpurun_out/t6_tests.log | tail -3
timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t6_bench_mlp.json 2> gpurun_out/t6_bench_mlp.err; echo "rc=$?"
timeout 200 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/t6_bench_c3.json 2> gpurun_out/t6_bench_c3.err; echo "rc=$?"
timeout 200 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/t6_bench_c4.json 2> gpurun_out/t6_bench_c4.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('mlp','c3','c4'):
    try:
        d=json.loads(open(f'gpurun_out/t6_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value']/1e6,3), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
