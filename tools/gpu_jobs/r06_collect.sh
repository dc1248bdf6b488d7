#!/bin/bash
# BUILD-BOX half of tools/gpu_jobs/r06_final.sh: copy what is judged from gpurun_out/ (scratch) into profiles/ (tracked).
TAG=${1:-r06}
cd "$(dirname "$0")/../.." || exit 1
for w in mlp c3 c4 hidden256; do cp gpurun_out/${TAG}_kernel_stats_$w.csv gpurun_out/${TAG}_pmc_$w.csv profiles/; done
cp gpurun_out/${TAG}_pmc_summary.json profiles/pmc_summary.json
cp gpurun_out/${TAG}_bench_mlp.json gpurun_out/${TAG}_bench_mlp_sustained25.json gpurun_out/${TAG}_smi_during_bench.txt gpurun_out/${TAG}_bench_g2_shared_device.json \
   gpurun_out/${TAG}_bench_g8_shared_device.json gpurun_out/${TAG}_dp_jitter.json profiles/
( grep -h "passed" gpurun_out/${TAG}_tests.log gpurun_out/${TAG}_tests_lazy.log | tail -2; grep "^\[smoke\]" gpurun_out/${TAG}_smoke.log | cut -c1-120 ) > profiles/${TAG}_gpu_tests_tail.txt
python - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
d = json.loads([l for l in open(f'profiles/{tag}_bench_mlp.json') if l.startswith('{')][0])
print('headline', round(d['value'] / 1e6, 1), 'M', round(d['ms_per_step'], 4), 'ms; sustained', round(d['sustained']['value'] / 1e6, 1), '; frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'])
print('self_check', json.dumps(d['self_check'])[:400])
for w in d.get('extra_workloads', []):
    r = w.get('roofline') or {}
    print('  ', w['config']['workload'][:44], round(w['value'] / 1e6, 3), 'M', r.get('kernel'), round(r.get('frac') or 0, 3), r.get('traffic'))
s = json.loads([l for l in open(f'profiles/{tag}_bench_mlp_sustained25.json') if l.startswith('{')][0])
print('25 s leg', s['sustained'])
print('pmc build', json.load(open('profiles/pmc_summary.json'))['_build'])
for w in ('mlp', 'c3', 'hidden256'):
    rows = list(csv.reader(open(f'profiles/{tag}_kernel_stats_{w}.csv')))
    print(w, [(r[0].split('(')[0][-36:], r[1], round(float(r[3]) / 1e3, 1)) for r in rows[1:6]])
j = json.load(open(f'profiles/{tag}_dp_jitter.json'))
print('jitter', [(l['injected_skew_us'], round(l['ms_per_step'], 3), l['wait_us_per_step']) for l in j['legs']])
g = json.loads([l for l in open(f'profiles/{tag}_bench_g2_shared_device.json') if l.startswith('{')][0])
print('g2', round(g['value'] / 1e6, 1), g['dist']['peer_wait']['wait_us_per_step'])
PY
