mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t1_tests.log 2>&1; echo "tests rc=$?" 
tail -4 gpurun_out/t1_tests.log
PFA_SYNC_READBACK=1 timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t1_bench_sync.json 2> gpurun_out/t1_bench_sync.err; echo "rc=$?"
timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t1_bench_lazy.json 2> gpurun_out/t1_bench_lazy.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('sync','lazy'):
    try:
        d=json.loads(open(f'gpurun_out/t1_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])
    except Exception as e: print(n, 'ERR', e)
PY
