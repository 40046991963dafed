#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/unit_lab_r4f.txt
: > $log
for rep in 1 2; do
for lib in aha_amd/libaha_hip.so aha_amd/libaha_hip_lab7.so aha_amd/libaha_hip_labhead.so; do
  AHA_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/lab_unit.py >> $log 2>&1 || { echo "lab $lib failed"; tail -5 $log; exit 1; }
  tail -1 $log
done
done
