#!/bin/bash
# A/B of the peer all-reduce's chunk size on the one-GPU box (two ranks sharing the device): PFA_P2P_CHUNK_BYTES, alternating.
mkdir -p gpurun_out
for cb in 16384 4096 16384 4096; do
  PFA_P2P_CHUNK_BYTES=$cb timeout 200 python3 bench.py --gpus 2 --steps 40 --warmup 5 --no-extra --sustained-seconds 0 > gpurun_out/p2p_chunk_$cb.json 2>/dev/null
  python - $cb <<'PY'
import json, sys
cb = sys.argv[1]
d = json.loads(open(f'gpurun_out/p2p_chunk_{cb}.json').read().strip().splitlines()[-1])
print('chunk', cb, round(d['value'] / 1e6, 1), 'M', round(d['ms_per_step'], 4), 'ms')
PY
done
