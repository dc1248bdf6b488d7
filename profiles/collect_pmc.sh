#!/bin/bash
# PMC collection on the GPU box (run through gpurun).  One rocprofv3 pass per counter group, --pmc only
# (never combined with sys/hip/hsa trace domains), outputs under gpurun_out/pmc_<tag>/.
set -u
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
EXTRA=${2:-}     # extra bench.py flags, e.g. "--policy lstm"
PASSES=${3:-"sq1 sq2 fetch write grbm"}
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-extra --sustained-seconds 0 $EXTRA"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmc_${TAG}/$name -o $name -- $CMD > $ROOT/gpurun_out/pmc_${TAG}_$name.log 2>&1
  echo "$name rc=$?"
}
mkdir -p $ROOT/gpurun_out/pmc_${TAG}
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_0-9]+|TCP_[A-Z_0-9]+)\b" | sort -u > $ROOT/gpurun_out/pmc_${TAG}/counters_available.txt
want() { [[ " $PASSES " == *" $1 "* ]]; }
want sq1 && run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
want sq2 && run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
want fetch && run fetch FETCH_SIZE
want write && run write WRITE_SIZE
want grbm && run grbm GRBM_GUI_ACTIVE
python $ROOT/profiles/summarize_pmc.py $ROOT/gpurun_out/pmc_${TAG} > $ROOT/gpurun_out/pmc_${TAG}/summary.csv
head -5 $ROOT/gpurun_out/pmc_${TAG}/summary.csv
