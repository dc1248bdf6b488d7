#!/bin/bash
# One gpurun call that re-verifies a build and collects the evidence committed under profiles/:
#   gpurun --timeout 900 -- 'bash tools/gpu_jobs/verify.sh <tag>'
# GPU test suite, smoke(), the default bench line (with self_check and cpu_baseline), a sustained run (1500 steps) and the
# rocprofv3 kernel statistics of the bench command.  Outputs under gpurun_out/<tag>_*.
TAG=${1:-verify}
mkdir -p gpurun_out
ROOT=$PWD
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed" gpurun_out/${TAG}_tests.log | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 200 python bench.py > gpurun_out/${TAG}_bench_mlp.json 2> gpurun_out/${TAG}_bench_mlp.err; echo "bench rc=$?"
timeout 100 python bench.py --steps 1500 --warmup 20 --no-cpu-baseline --no-breakdown > gpurun_out/${TAG}_bench_mlp_sustained.json 2> gpurun_out/${TAG}_bench_mlp_sustained.err; echo "sustained rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ks_mlp -o ks -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-breakdown > $ROOT/gpurun_out/${TAG}_ks_mlp.log 2>&1; echo "rocprof rc=$?" )
find gpurun_out/${TAG}_ks_mlp -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_mlp.csv \;
rm -rf gpurun_out/${TAG}_ks_mlp
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for n in ('bench_mlp', 'bench_mlp_sustained'):
    d = json.loads(open(f'gpurun_out/{tag}_{n}.json').read().strip().splitlines()[-1])
    print(n, round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 4), 'ms', 'grad', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'us frac',
          round(d['roofline']['frac'], 3), d.get('self_check', {}).get('max_abs_weight_err'), d.get('cpu_baseline', {}).get('value'))
PY
head -6 gpurun_out/${TAG}_kernel_stats_mlp.csv | cut -c1-160
