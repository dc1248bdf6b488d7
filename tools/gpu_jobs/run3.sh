mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t3_tests.log 2>&1; echo "tests rc=$?" 
grep -n "passed\|failed" gpurun_out/t3_tests.log | tail -3
for s in 1 5 100000; do
PFA_BENCH_EVENT_STRIDE=$s timeout 120 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/t3_bench_s$s.json 2> gpurun_out/t3_bench_s$s.err; echo "rc=$?"
done
python - <<'PY'
import json
for n in ('s1','s5','s100000'):
    try:
        d=json.loads(open(f'gpurun_out/t3_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, round(d['value']/1e6,1), d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items()})
    except Exception as e: print(n, 'ERR', e)
PY
