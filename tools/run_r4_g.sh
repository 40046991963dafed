#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab gpurun_out/r4g
log=gpurun_out/lab/unit_lab_r4g.txt
: > $log
for rep in 1 2; do
for lib in aha_amd/libaha_hip.so aha_amd/libaha_hip_labhead.so; do
  AHA_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/lab_unit.py >> $log 2>&1 || { echo "lab $lib failed"; tail -5 $log; exit 1; }
  tail -1 $log
done
done
T=900 bash tools/gpu_tests.sh || exit 1
timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.log || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/r4g/bench.json')); print(d['value'], d['roofline']['kernels_ms'])"
