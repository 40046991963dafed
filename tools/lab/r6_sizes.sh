#!/bin/bash
# engine 6 (and engine 4 beside it) on cfg 3's text at several batch sizes: where the per-lane text windows stop fitting the caches
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/sizes_${1:-run}.txt
: > $log
for n in 67108864 268435456 1073741824; do
  for sk in 0 1; do
    AHA_SKIP=$sk AHA_LAB_NOTE="n=$n AHA_SKIP=$sk" timeout -k 10 300 python3 tools/lab_unit.py $n >> $log 2>> gpurun_out/lab/sizes_err.txt || { tail -5 gpurun_out/lab/sizes_err.txt; exit 1; }
    tail -1 $log
  done
done
