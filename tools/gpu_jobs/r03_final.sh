#!/bin/bash
# Round-3 evidence job (one gpurun call): GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the bench command,
# the PMC passes and the stamped pmc_summary.json.  Outputs under gpurun_out/<tag>_*; copy what is judged into profiles/.
TAG=${1:-r03}
mkdir -p gpurun_out
ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/${TAG}_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ks_mlp -o ks -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 > $ROOT/gpurun_out/${TAG}_ks_mlp.log 2>&1; echo "rocprof rc=$?" )
find gpurun_out/${TAG}_ks_mlp -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_mlp.csv \;
rm -rf gpurun_out/${TAG}_ks_mlp
for W in c3 c4; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ks_$W -o ks -- python $ROOT/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 > $ROOT/gpurun_out/${TAG}_ks_$W.log 2>&1; echo "rocprof $W rc=$?" )
  find gpurun_out/${TAG}_ks_$W -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_$W.csv \;
  rm -rf gpurun_out/${TAG}_ks_$W
done
bash profiles/collect_pmc.sh ${TAG} "" "sq1 fetch write" > gpurun_out/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
python profiles/make_pmc_summary.py gpurun_out/pmc_${TAG}/summary.csv > gpurun_out/${TAG}_pmc_summary.log 2>&1; echo "summary rc=$?"
cp profiles/pmc_summary.json gpurun_out/${TAG}_pmc_summary.json
cp gpurun_out/pmc_${TAG}/summary.csv gpurun_out/${TAG}_pmc.csv
rm -rf gpurun_out/pmc_${TAG}/sq1 gpurun_out/pmc_${TAG}/fetch gpurun_out/pmc_${TAG}/write
# the default line LAST: it now finds a summary stamped with this build; shader clock and power sampled next to it (the sustained leg)
( for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/${TAG}_smi_during_bench.txt ) &
SMI=$!
( time timeout 600 python bench.py > gpurun_out/${TAG}_bench_mlp.json 2> gpurun_out/${TAG}_bench_mlp.err ) 2> gpurun_out/${TAG}_bench_wall.txt; echo "bench rc=$?"
kill $SMI 2>/dev/null
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open(f'gpurun_out/{tag}_bench_mlp.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 4), 'ms; sustained', round(d.get('sustained_value', 0) / 1e6, 1))
print({k: d['roofline'][k] for k in ('frac', 'frac_executed', 'frac_useful', 'avg_launch_ms', 'traffic', 'traffic_source')})
print(d.get('roofline_hbm'))
print(d.get('self_check'))
print([(w.get('config', {}).get('workload', '')[:40], w.get('value'), (w.get('roofline') or {}).get('frac')) if 'value' in w else w for w in d.get('extra_workloads', [])])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
cat gpurun_out/${TAG}_bench_wall.txt | tail -4
head -8 gpurun_out/${TAG}_kernel_stats_mlp.csv | cut -c1-170
cat gpurun_out/${TAG}_pmc_summary.log | tail -2
