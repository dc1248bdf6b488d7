#!/bin/bash
# PMC passes (rocprofv3 --pmc only) of the gradient kernel alone: tools/grad_loop.py <products>; output gpurun_out/<tag>_pmc_grad_<products>.txt
TAG=${1:-r04}; MODE=${2:-1}
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/pmcg_$name -o $name -- python $ROOT/tools/grad_loop.py $MODE 24 > $ROOT/gpurun_out/pmcg_$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
run c SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_LDS
python - <<PY > $ROOT/gpurun_out/${TAG}_pmc_grad_$MODE.txt
import csv, glob, collections
for name in 'abc':
    for f in glob.glob('$ROOT/gpurun_out/pmcg_%s/**/*counter_collection.csv' % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if 'ppo_mlp_grad' in r['Kernel_Name']:
                k = r['Counter_Name']; acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f'{k:32s} per launch {v / n:16.1f}   ({n} launches)')
PY
cat $ROOT/gpurun_out/${TAG}_pmc_grad_$MODE.txt
rm -rf $ROOT/gpurun_out/pmcg_a $ROOT/gpurun_out/pmcg_b $ROOT/gpurun_out/pmcg_c
