#!/bin/bash
# Same-box A/B of the host-side levers of the iteration loop (one gpurun call, alternating runs): the early GAE pass
# (PFA_EARLY_GAE) and the polled readback wait (PFA_SPIN_WAIT_US), each against the default, on the headline workload.
#   gpurun --timeout 600 -- 'bash tools/gpu_jobs/ab_host.sh <tag> [reps]'
TAG=${1:-ab_host}
REPS=${2:-2}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_host.txt
: > $OUT
run() {  # label, env assignments...
  local label=$1; shift
  local line
  line=$(env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 2>/dev/null | tail -1)
  python - "$label" "$line" >> $OUT <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(f"{label:28s} {d['value'] / 1e6:8.1f} M env steps/s   {d['ms_per_step'] * 1e3:8.1f} us/step   grad {d['roofline']['avg_launch_ms'] * 1e3:6.2f} us")
except Exception as e:
    print(f"{label:28s} FAILED {e!r} {line[:200]!r}")
PY
}
for i in $(seq 1 $REPS); do
  run "default" PFA_NOP=1
  run "PFA_EARLY_GAE=0" PFA_EARLY_GAE=0
  run "PFA_SPIN_WAIT_US=0" PFA_SPIN_WAIT_US=0
  run "both off (round-5 loop)" PFA_EARLY_GAE=0 PFA_SPIN_WAIT_US=0
  run "PFA_LAZY_READBACK=1" PFA_LAZY_READBACK=1
done
cat $OUT
