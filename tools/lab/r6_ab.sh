#!/bin/bash
# Round 6 A/B leg: tools/lab_unit.py (cfg 3, 1 GiB, HIP events inside the library) over the product library -- engine 4 (ku_traverse),
# the pair engine (AHA_PAIR=1), with AB_SKIP=1 also the skip-ahead traversal (AHA_SKIP=1) -- and every lab build present (aha_amd/libaha_hip_lab_*.so), in
# ONE call: the boxes of the pool differ by up to 12 %.  tools/lab/r6_ab.sh <tag> [reps]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/ab_${1:-run}.txt
: > $log
run() {  # lib, note, env...
  local lib=$1 note=$2; shift 2
  env "$@" AHA_LAB_NOTE="$note" AHA_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 tools/lab_unit.py >> $log 2>> gpurun_out/lab/ab_err.txt || { tail -5 gpurun_out/lab/ab_err.txt; exit 1; }
  tail -1 $log
}
for rep in $(seq 1 ${2:-2}); do
  run aha_amd/libaha_hip.so "engine 4" AHA_PAIR=0 AHA_SKIP=0
  run aha_amd/libaha_hip.so "AHA_PAIR=1" AHA_PAIR=1
  export AHA_PAIR=1
  [ -n "$AB_SKIP" ] && run aha_amd/libaha_hip.so "AHA_SKIP=1" AHA_SKIP=1 AHA_PAIR=0
  for lib in aha_amd/libaha_hip_lab_*.so; do
    [ -f "$lib" ] || continue
    run $lib "" AHA_X=0
  done
done
