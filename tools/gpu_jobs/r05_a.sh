#!/bin/bash
# round 5, job A: demo end-to-end (needs the staged reference) + the GPU suite with durations + host memory
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
free -g | head -2
timeout 900 python -m pytest tests/test_gpu_demo_reference.py -x -q -rs > gpurun_out/r05_demo_tests.log 2>&1; echo "demo tests rc=$?"
tail -5 gpurun_out/r05_demo_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_tests.log
