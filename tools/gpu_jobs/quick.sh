#!/bin/bash
# Short re-check of a build: GPU test suite + rocprofv3 kernel statistics of a 10-step bench run (which prints the bench line too).
TAG=${1:-quick}
mkdir -p gpurun_out
ROOT=$PWD
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
grep -n "passed\|failed" gpurun_out/${TAG}_tests.log | tail -3
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ks_mlp -o ks -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-breakdown > $ROOT/gpurun_out/${TAG}_ks_mlp.log 2>&1; echo "rocprof rc=$?" )
find gpurun_out/${TAG}_ks_mlp -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_mlp.csv \;
rm -rf gpurun_out/${TAG}_ks_mlp
grep -h '"metric"' gpurun_out/${TAG}_ks_mlp.log | cut -c1-330
cut -d, -f1-4 gpurun_out/${TAG}_kernel_stats_mlp.csv | sed 's/(.*)"/"/' | head -16
