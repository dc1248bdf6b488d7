#!/bin/bash
# GPU-box half of tools/gpu_jobs/with_reference.sh.  Expects the staged reference at ./_refstage (git-ignored, shipped by gpurun).
#  (a) the reference's unmodified demo.py end to end on the device engine (tests/test_gpu_demo_reference.py)
#  (b) the reference's own CPU path (Serial and Multiprocessing) timed on THIS box's host cores for C1, C2, the C3-policy and the C4-policy
# Outputs: gpurun_out/r06_demo_end_to_end.json, gpurun_out/r06_reference_cpu_on_gpu_box.jsonl (+ logs)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
test -f _refstage/demo.py || { echo "no staged reference"; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_demo_reference.py -x -q -rs > gpurun_out/r06_demo_tests.log 2>&1; echo "demo tests rc=$?"
tail -5 gpurun_out/r06_demo_tests.log
OUT=gpurun_out/r06_reference_cpu_on_gpu_box.jsonl
rm -f $OUT
PHYS=$(python -c "import psutil; print(psutil.cpu_count(logical=False))")
echo "physical cores $PHYS, logical $(nproc)"
t() { timeout 600 python tools/time_reference.py --out $OUT "$@" 2>> gpurun_out/r06_reference_cpu.err | cut -c1-400; }
# C1 (64 envs): Serial; torch threads 1 / 16 (64 envs x 128-wide GEMMs do not feed a socket)
t --config c1 --iters 5 --threads 16
t --config c1 --iters 5 --threads 1
t --config c1 --backend multiprocessing --workers 8 --iters 5 --threads 16
# C2 (4096 envs, the headline configuration): Serial with 16 / all physical threads; Multiprocessing with 16 / 64 / all physical workers
t --config c2 --iters 3 --threads 16
t --config c2 --iters 3 --threads $PHYS
t --config c2 --backend multiprocessing --workers 16 --iters 3 --threads 16
t --config c2 --backend multiprocessing --workers 64 --iters 3 --threads 16
t --config c2 --backend multiprocessing --workers $PHYS --iters 3 --threads 16
# C4-policy (NatureCNN on uint8 (4,84,84) frames; a bounded sample of the 8192 envs: 256 and 1024 envs x 32 steps)
t --config c4 --envs 256 --iters 2 --threads 16
t --config c4 --envs 1024 --iters 1 --threads 16
t --config c4 --envs 1024 --iters 1 --threads $PHYS
# C3-policy (LSTM 128, 160-byte rows, 4096 envs)
t --config c3 --iters 3 --threads 16
t --config c3 --backend multiprocessing --workers 64 --iters 3 --threads 16
wc -l $OUT
