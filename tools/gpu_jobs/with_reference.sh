#!/bin/bash
# BUILD-BOX script: ONE gpurun job that has both a GPU and the reference's files.
#   1. tools/stage_reference.py copies the hot path's reference files (SURVEY 8c's list + demo.py + config.yaml) into the
#      git-ignored ./_refstage (gpurun ships git-ignored files; the GPU box has no /root/reference)
#   2. gpurun runs tools/gpu_jobs/with_reference_job.sh (or "$@") on the GPU box
#   3. the staging directory is removed again, whatever happened — it must not exist at round end and is never committed
# Afterwards: python tools/collect_reference_timing.py  ->  profiles/r06_reference_cpu_on_gpu_box.json
cd "$(dirname "$0")/../.." || exit 1
python tools/stage_reference.py || exit 1
trap 'python tools/stage_reference.py --clean' EXIT
CMD=${*:-bash tools/gpu_jobs/with_reference_job.sh}
/usr/local/graft/bin/gpurun --timeout ${GPURUN_TIMEOUT:-1500} -- "$CMD"
