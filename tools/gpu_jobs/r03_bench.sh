#!/bin/bash
# bench lines only: --gpus 2 (plain python, ranks share the device) and the N = 1 default line
TAG=${1:-r03b}
mkdir -p gpurun_out
timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_g2.json 2> gpurun_out/${TAG}_bench_g2.err; echo "bench --gpus 2 rc=$?"
cut -c1-400 gpurun_out/${TAG}_bench_g2.json; grep -v "^\[Gloo\]\|^$" gpurun_out/${TAG}_bench_g2.err | tail -5
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_g1.json 2> gpurun_out/${TAG}_bench_g1.err; echo "bench --gpus 1 rc=$?"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for g in ('g2', 'g1'):
    try:
        d = json.loads(open(f'gpurun_out/{tag}_bench_{g}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(g, 'no line', e); continue
    print(g, round(d['value'] / 1e6, 1), 'M steps/s', round(d['ms_per_step'], 4), {k: d['roofline'][k] for k in ('frac', 'frac_executed', 'avg_launch_ms', 'traffic', 'traffic_source')},
          d.get('roofline_hbm'), d.get('sustained'), d.get('dist'))
    print(d.get('extra_workloads'))
PY
tail -3 gpurun_out/${TAG}_bench_g1.err
