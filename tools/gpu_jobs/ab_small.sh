#!/bin/bash
# Same-box A/B of the two small-launch levers of round 5 (alternating runs on the headline workload): the self-starting GAE window
# (PFA_GAE_SELF) and train()'s report from the update's last launch (PFA_FUSED_LOG), each against the default.
TAG=${1:-ab_small}
REPS=${2:-2}
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_small.txt
: > $OUT
run() {
  local label=$1; shift
  local line
  line=$(env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --sustained-seconds 0 2>/dev/null | tail -1)
  python - "$label" "$line" >> $OUT <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    k = d.get('kernel_ms_per_step', {})
    print(f"{label:28s} {d['value'] / 1e6:8.1f} M env steps/s   {d['ms_per_step'] * 1e3:8.1f} us/step   gae {k.get('gae', {}).get('ms_per_step', 0) * 1e3:5.1f} us  reduce+adam {k.get('ppo_reduce_adam', {}).get('ms_per_step', 0) * 1e3:6.1f} us")
except Exception as e:
    print(f"{label:28s} FAILED {e!r} {line[:200]!r}")
PY
}
for i in $(seq 1 $REPS); do
  run "default" PFA_NOP=1
  run "PFA_GAE_SELF=0" PFA_GAE_SELF=0
  run "PFA_FUSED_LOG=0" PFA_FUSED_LOG=0
  run "both off" PFA_GAE_SELF=0 PFA_FUSED_LOG=0
done
cat $OUT
