#!/bin/bash
# whole GPU suite + the side-workload bench lines
TAG=${1:-r03s}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/${TAG}_tests.log
timeout 400 python bench.py --workload c4 --policy lstm --steps 2 --warmup 1 --sustained-seconds 0 --no-breakdown > gpurun_out/${TAG}_c4lstm.json 2> gpurun_out/${TAG}_c4lstm.err; echo "c4 lstm rc=$?"
cut -c1-700 gpurun_out/${TAG}_c4lstm.json; tail -4 gpurun_out/${TAG}_c4lstm.err
