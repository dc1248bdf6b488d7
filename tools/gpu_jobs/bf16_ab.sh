#!/bin/bash
# A/B of the bf16-path gradient kernel's versions in ONE gpurun call (same box, same clocks): the tree's own header first, then the
# earlier / alternative versions kept under tools/experiments/ppo_bf16_<v>.hpp (v1_three_barriers, glds_variant), each compiled on the
# box; launch time from tools/bf16_grad_check.py.        bash tools/gpu_jobs/bf16_ab.sh glds_variant v1_three_barriers
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open('gpurun_out/bf16_grad_check_a8.json'))
print(sys.argv[1], d['timing']['bf16x6']['kernels_us'], 'fp32', d['timing']['fp32']['kernels_us']['ppo_mlp_grad'], 'w1 rel', d['grad_diff']['w1']['rel'])
PY
}
cp pufferlib_amd/csrc/ppo_bf16.hpp /tmp/ppo_bf16_tree.hpp
python tools/bf16_grad_check.py 8 > /dev/null 2>&1; show tree
for v in "$@"; do
  cp tools/experiments/ppo_bf16_$v.hpp pufferlib_amd/csrc/ppo_bf16.hpp
  python -c "from pufferlib_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1 || echo "build $v failed"
  python tools/bf16_grad_check.py 8 > /dev/null 2>&1; show $v
done
cp /tmp/ppo_bf16_tree.hpp pufferlib_amd/csrc/ppo_bf16.hpp
python -c "from pufferlib_amd import _lib; _lib.build(force=True)" > /dev/null 2>&1
python tools/bf16_grad_check.py 8 > /dev/null 2>&1; show tree-again
