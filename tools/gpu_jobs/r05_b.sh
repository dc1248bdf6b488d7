#!/bin/bash
# demo end to end, verbose (needs the staged reference)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PYTHONUNBUFFERED=1
rm -f gpurun_out/r05_demo_process_output.txt
mkdir -p /tmp/demo_run && cp _refstage/config.yaml /tmp/demo_run/ && ( cd /tmp/demo_run && PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/tests/shims timeout 300 python -X faulthandler -m pufferlib_amd.demo --reference $GRAFT_REPO_ROOT/_refstage -- --env squared --mode train --vec serial --train.device cuda --train.total-timesteps 200000 > $GRAFT_REPO_ROOT/gpurun_out/r05_demo_direct.log 2>&1; echo "direct rc=$?"; ls -la /tmp/demo_run /tmp/demo_run/experiments/* 2>/dev/null | head -20 )
tail -c 5000 gpurun_out/r05_demo_direct.log
timeout 900 python -m pytest tests/test_gpu_demo_reference.py -x -q -rs > gpurun_out/r05_demo_tests.log 2>&1; echo "demo tests rc=$?"
tail -5 gpurun_out/r05_demo_tests.log; tail -c 6000 gpurun_out/r05_demo_process_output.txt
