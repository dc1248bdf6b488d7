#!/bin/bash
# Round-6 evidence job (one gpurun call): GPU suite, smoke, rocprofv3 kernel stats of the bench command for every workload, the PMC
# passes of every workload merged into ONE stamped pmc_summary.json (so the side lines carry `traffic` too), then the default
# bench line (which runs the side workloads as child processes).  Outputs under gpurun_out/<tag>_*; copy what is judged into profiles/.
TAG=${1:-r06}
SKIP_TESTS=${2:-0}
mkdir -p gpurun_out
ROOT=$PWD
if [ "$SKIP_TESTS" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
  grep -E 'passed|failed' gpurun_out/${TAG}_tests.log | tail -2
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
fi
ks() {  # name, bench flags
  local W=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_ks_$W -o ks -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 > $ROOT/gpurun_out/${TAG}_ks_$W.log 2>&1; echo "rocprof $W rc=$?" )
  find gpurun_out/${TAG}_ks_$W -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats_$W.csv \;
  rm -rf gpurun_out/${TAG}_ks_$W
}
ks mlp --steps 10
ks c3 --workload c3 --steps 4 --warmup 1
ks c4 --workload c4 --steps 4 --warmup 1
ks hidden256 --hidden 256 --steps 4 --warmup 1
pmc() {  # name, bench flags (one string)
  bash profiles/collect_pmc.sh ${TAG}_$1 "$2" "sq1 fetch write" > gpurun_out/${TAG}_pmc_$1.log 2>&1; echo "pmc $1 rc=$?"
  cp gpurun_out/pmc_${TAG}_$1/summary.csv gpurun_out/${TAG}_pmc_$1.csv
  rm -rf gpurun_out/pmc_${TAG}_$1/sq1 gpurun_out/pmc_${TAG}_$1/fetch gpurun_out/pmc_${TAG}_$1/write
}
pmc mlp ""
pmc c3 "--workload c3"
pmc c4 "--workload c4"
pmc hidden256 "--hidden 256"
python profiles/make_pmc_summary.py gpurun_out/${TAG}_pmc_mlp.csv gpurun_out/${TAG}_pmc_c3.csv gpurun_out/${TAG}_pmc_c4.csv gpurun_out/${TAG}_pmc_hidden256.csv > gpurun_out/${TAG}_pmc_summary.log 2>&1; echo "summary rc=$?"
cp profiles/pmc_summary.json gpurun_out/${TAG}_pmc_summary.json
# the default line LAST: it now finds a summary stamped with this build; shader clock and power sampled next to it
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/${TAG}_smi_during_bench.txt ) &
SMI=$!
( time timeout 900 python bench.py > gpurun_out/${TAG}_bench_mlp.json 2> gpurun_out/${TAG}_bench_mlp.err ) 2> gpurun_out/${TAG}_bench_wall.txt; echo "bench rc=$?"
kill $SMI 2>/dev/null
# a 25 s leg of the same loop (~19 000 iterations: 10^10 env steps): no drift, no NaN, no tape underrun, generations / sequence numbers far past 2^16
timeout 300 python bench.py --no-extra --no-cpu-baseline --no-breakdown --sustained-seconds 25 > gpurun_out/${TAG}_bench_mlp_sustained25.json 2> gpurun_out/${TAG}_bench_sus.err; echo "sustained rc=$?"
# the deferred-readback mode through a slice of the suite (readback.py's direct buffers in lazy mode)
PFA_LAZY_READBACK=1 timeout 600 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_squared.py tests/test_gpu_learning.py tests/test_gpu_lstm.py -q -x > gpurun_out/${TAG}_tests_lazy.log 2>&1; echo "lazy tests rc=$?"; grep -E 'passed|failed' gpurun_out/${TAG}_tests_lazy.log | tail -1
timeout 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_bench_g2_shared_device.json 2> gpurun_out/${TAG}_bench_g2.err; echo "g2 rc=$?"
timeout 900 python3 bench.py --gpus 8 --steps 5 --warmup 2 --no-extra --no-cpu-baseline --no-breakdown --sustained-seconds 0 > gpurun_out/${TAG}_bench_g8_shared_device.json 2> gpurun_out/${TAG}_bench_g8.err; echo "g8 rc=$?"
timeout 600 python tools/dp_jitter.py --world 2 --envs 4096 --horizon 128 --iters 20 --skews 0,50,100,200,500,1000 --out gpurun_out/${TAG}_dp_jitter.json > gpurun_out/${TAG}_dp_jitter.log 2>&1; echo "jitter rc=$?"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    g8 = json.loads(open(f'gpurun_out/{tag}_bench_g8_shared_device.json').read().strip().splitlines()[-1])
    print('g8 (ranks share the one device: functional, not scaling)', round(g8['value'] / 1e6, 1), 'M', g8['dist']['p2p_selftest_passed'], g8['dist']['allreduce_calls'], json.dumps(g8['dist']['transports']))
except Exception as e:
    print('g8 failed', e)
d = json.loads(open(f'gpurun_out/{tag}_bench_mlp.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 4), 'ms; sustained', round(d.get('sustained_value', 0) / 1e6, 1))
print({k: d['roofline'][k] for k in ('frac', 'frac_executed', 'frac_useful', 'avg_launch_ms', 'traffic', 'traffic_source')})
print(d.get('roofline_hbm'))
print(d.get('self_check'))
print(d.get('profile_ms_per_step'))
for w in d.get('extra_workloads', []):
    print((w.get('config', {}).get('workload', '')[:50], w.get('value'), {k: (w.get('roofline') or {}).get(k) for k in ('frac', 'traffic')}, (w.get('cpu_baseline') or {}).get('value')) if 'value' in w else w)
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
g = json.loads(open(f'gpurun_out/{tag}_bench_g2_shared_device.json').read().strip().splitlines()[-1])
print('g2', round(g['value'] / 1e6, 1), json.dumps(g['dist']['transports']))
PY
cat gpurun_out/${TAG}_bench_wall.txt | tail -4
head -8 gpurun_out/${TAG}_kernel_stats_mlp.csv | cut -c1-170
cat gpurun_out/${TAG}_pmc_summary.log | tail -2
