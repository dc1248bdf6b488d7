#!/bin/bash
# kernel timeline of a few bench steps (start/end timestamps per dispatch) -> gpurun_out/<tag>_kernel_trace.csv
TAG=${1:-r03t}
ROOT=$PWD
mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_kt -o kt -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 > $ROOT/gpurun_out/${TAG}_kt.log 2>&1; echo "rocprof rc=$?" )
find gpurun_out/${TAG}_kt -name "*kernel_trace.csv" -exec cp {} gpurun_out/${TAG}_kernel_trace.csv \;
find gpurun_out/${TAG}_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_kt
python - "$TAG" <<'PY'
import csv, sys, re
tag = sys.argv[1]
rows = list(csv.DictReader(open(f'gpurun_out/{tag}_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    m = re.search(r'pfa::([A-Za-z0-9_]+)', n); return m.group(1) if m else n.split('(')[0][:40]
# last ~2 steps: find the last 3 rollout kernels
idx = [i for i, r in enumerate(rows) if 'rollout_mlp_squared' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
busy = 0; prev_end = None; gaps = {}
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    if prev_end is not None:
        gaps.setdefault((short(prev_name), short(r['Kernel_Name'])), []).append(s - prev_end)
    prev_end, prev_name = e, r['Kernel_Name']
total = int(rows[b]['Start_Timestamp']) - t0
print('one step: wall', total / 1e3, 'us; kernels busy (sum)', busy / 1e3, 'us; dispatches', len(seg))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    print(f'  gap {k[0]:28s} -> {k[1]:28s} n={len(v):3d} mean {sum(v)/len(v)/1e3:7.2f} us  total {sum(v)/1e3:8.1f} us')
PY
