mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t2_tests.log 2>&1; echo "tests rc=$?" 
tail -4 gpurun_out/t2_tests.log
PFA_SYNC_READBACK=1 timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t2_bench_sync.json 2> gpurun_out/t2_bench_sync.err; echo "rc=$?"
timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t2_bench_lazy.json 2> gpurun_out/t2_bench_lazy.err; echo "rc=$?"
PFA_TAPE_PREFETCH=0 timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t2_bench_notape.json 2> gpurun_out/t2_bench_notape.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('sync','lazy','notape'):
    try:
        d=json.loads(open(f'gpurun_out/t2_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['avg_launch_ms'], {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
