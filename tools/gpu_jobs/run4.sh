mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t4_tests.log 2>&1; echo "tests rc=$?" 
grep -n "passed\|failed" gpurun_out/t4_tests.log | tail -3
PFA_BENCH_EVENT_STRIDE=5 timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t4_bench_lazy.json 2> gpurun_out/t4_bench_lazy.err; echo "rc=$?"
PFA_SYNC_READBACK=1 PFA_BENCH_EVENT_STRIDE=5 timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t4_bench_sync.json 2> gpurun_out/t4_bench_sync.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('lazy','sync'):
    try:
        d=json.loads(open(f'gpurun_out/t4_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
