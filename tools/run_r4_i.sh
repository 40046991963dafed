#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4i
run() { local name=$1; shift; timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/r4i/$name.json 2> gpurun_out/r4i/$name.log || { echo "$name failed"; tail -3 gpurun_out/r4i/$name.log; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/r4i/$name.json')); print('$name', d['value'], d['roofline']['kernel'], d['roofline']['kernels_ms'], d['config'].get('compile_s'))"; }
run bench_cfg5_auto --config 5 --steps 5 --warmup 2
AHA_ENGINE=v2 run bench_cfg5_v2 --config 5 --steps 5 --warmup 2
T=1000 bash tools/gpu_tests.sh || exit 1
timeout -k 10 200 python3 tools/exp_overlap_rebuild.py > gpurun_out/r4i/overlap_rebuild.txt 2>&1 || { tail -5 gpurun_out/r4i/overlap_rebuild.txt; exit 1; }
tail -12 gpurun_out/r4i/overlap_rebuild.txt
