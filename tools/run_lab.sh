#!/bin/bash
# One gpurun call of the round-3 lab: VALU issue rates, then the traversal's diagnostic variants.
# The lab library and the micro-benchmark are built beforehand (make -C aha_amd/csrc diag; hipcc tools/ubench_valu.hip).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
timeout -k 10 ${T:-500} python3 tools/lab_traverse.py "$@" > gpurun_out/lab/lab_traverse.log 2>&1 || { echo "lab failed"; tail -20 gpurun_out/lab/lab_traverse.log; exit 1; }
tail -60 gpurun_out/lab/lab_traverse.log
timeout -k 10 60 stdbuf -oL tools/ubench_valu > gpurun_out/lab/valu_rate.txt 2>&1 || { echo "ubench_valu failed"; tail -5 gpurun_out/lab/valu_rate.txt; exit 1; }
echo "valu done"
