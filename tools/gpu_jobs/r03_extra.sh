#!/bin/bash
# Side evidence on the final build (one gpurun call): PMC passes of the c4 workload (rows-form traffic), the plain-python
# `bench.py --gpus 2` line on this 1-GPU box (ranks share the device: functional, not a scaling number), the long sustained leg.
TAG=${1:-r03}
mkdir -p gpurun_out
bash profiles/collect_pmc.sh ${TAG}_c4 "--workload c4" "sq1 fetch write" > gpurun_out/${TAG}_pmc_c4.log 2>&1; echo "pmc c4 rc=$?"
cp gpurun_out/pmc_${TAG}_c4/summary.csv gpurun_out/${TAG}_pmc_c4.csv
rm -rf gpurun_out/pmc_${TAG}_c4/sq1 gpurun_out/pmc_${TAG}_c4/fetch gpurun_out/pmc_${TAG}_c4/write
timeout 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-extra > gpurun_out/${TAG}_bench_g2_shared_device.json 2> gpurun_out/${TAG}_bench_g2.err; echo "g2 rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-breakdown --sustained-seconds 10 > gpurun_out/${TAG}_bench_mlp_sustained10.json 2>/dev/null; echo "sustained rc=$?"
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for f in (f'gpurun_out/{tag}_bench_g2_shared_device.json', f'gpurun_out/{tag}_bench_mlp_sustained10.json'):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['n_gpus'], round(d['value'] / 1e6, 1), d.get('sustained'), d.get('dist'))
PY
head -12 gpurun_out/${TAG}_pmc_c4.csv | cut -c1-200
