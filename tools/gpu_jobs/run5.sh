# final verification of the round-2b state: GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the same command
mkdir -p gpurun_out
ROOT=$PWD
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/t5_tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed" gpurun_out/t5_tests.log | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t5_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/t5_smoke.log
timeout 200 python bench.py > gpurun_out/t5_bench_mlp.json 2> gpurun_out/t5_bench_mlp.err; echo "bench rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/t5_ks_mlp -o ks -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-breakdown > $ROOT/gpurun_out/t5_ks_mlp.log 2>&1; echo "rocprof rc=$?" )
find gpurun_out/t5_ks_mlp -name "*kernel_stats.csv" -exec cp {} gpurun_out/t5_kernel_stats_mlp.csv \;
head -12 gpurun_out/t5_kernel_stats_mlp.csv | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/t5_bench_mlp.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,1), d['ms_per_step'], d['roofline'], {k:v['ms_per_step'] for k,v in d['kernel_ms_per_step'].items()}, d.get('self_check'), d.get('cpu_baseline',{}).get('value'))
PY
