#!/bin/bash
# LDS counters (rocprofv3 --pmc, sq2 pass of profiles/collect_pmc.sh) of one bench workload: gpurun_out/<tag>_lds_<name>.csv
TAG=${1:-r04}; NAME=${2:-c4}; FLAGS=${3:---workload c4}
bash profiles/collect_pmc.sh ${TAG}lds_$NAME "$FLAGS" "sq2" > gpurun_out/${TAG}_lds_$NAME.log 2>&1; echo "lds $NAME rc=$?"
cp gpurun_out/pmc_${TAG}lds_$NAME/summary.csv gpurun_out/${TAG}_lds_$NAME.csv
rm -rf gpurun_out/pmc_${TAG}lds_$NAME
