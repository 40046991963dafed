#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4e gpurun_out/lab
log=gpurun_out/lab/unit_lab_r4e.txt
: > $log
for lib in aha_amd/libaha_hip.so aha_amd/libaha_hip_lab7.so aha_amd/libaha_hip.so aha_amd/libaha_hip_lab7.so; do
  AHA_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/lab_unit.py >> $log 2>&1 || { echo "lab $lib failed"; tail -5 $log; exit 1; }
  tail -1 $log
done
T=900 bash tools/gpu_tests.sh || exit 1
run() { local name=$1; shift; timeout -k 10 280 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/r4e/$name.json 2> gpurun_out/r4e/$name.log || { echo "$name failed"; tail -3 gpurun_out/r4e/$name.log; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/r4e/$name.json')); print('$name', d['value'], d['roofline']['kernels_ms'], d.get('end_to_end'))"; }
run bench_cfg3 --steps 10 --warmup 3
run bench_cfg2_64MiB --config 2 --steps 20 --warmup 3 --no-end-to-end
run bench_cfg2_1GiB --config 2 --bytes 1073741824 --steps 10 --warmup 3 --no-end-to-end
AHA_HEADERS_FIRST=0 run bench_cfg2_64MiB_shadow --config 2 --steps 20 --warmup 3 --no-end-to-end
AHA_HEADERS_FIRST=0 run bench_cfg2_1GiB_shadow --config 2 --bytes 1073741824 --steps 10 --warmup 3 --no-end-to-end
