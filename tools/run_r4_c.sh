#!/bin/bash
# round 4: two-walk kernel after the ordering fix: labs on both kernels, then the SQ/TCC counters of both
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4c gpurun_out/lab
log=gpurun_out/lab/unit_lab_r4c.txt
: > $log
for w in 2 1; do
  for lib in aha_amd/libaha_hip.so aha_amd/libaha_hip_lab*.so; do
    AHA_UNIT_WALKS=$w AHA_LAB_NOTE="walks=$w" AHA_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/lab_unit.py >> $log 2>&1 || { echo "lab $lib failed"; tail -5 $log; exit 1; }
    tail -1 $log
  done
done
AHA_UNIT_WALKS=2 bash tools/pmc_kernel.sh r4c/pmc_w2 ku2_traverse --no-end-to-end > gpurun_out/r4c/pmc_w2.txt 2>&1 || exit 1
AHA_UNIT_WALKS=1 bash tools/pmc_kernel.sh r4c/pmc_w1 ku_traverse --no-end-to-end > gpurun_out/r4c/pmc_w1.txt 2>&1 || exit 1
cat gpurun_out/r4c/pmc_w2.txt gpurun_out/r4c/pmc_w1.txt
