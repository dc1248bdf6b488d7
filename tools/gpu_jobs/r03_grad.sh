#!/bin/bash
# variant timing of the gradient kernel + the PPO parity tests on the product build
mkdir -p gpurun_out
timeout 300 python tools/variant_bench.py run 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_ppo.py tests/test_gpu_parity_full.py tests/test_gpu_multidiscrete.py tests/test_gpu_learning.py -x -q 2>&1 | tail -15
