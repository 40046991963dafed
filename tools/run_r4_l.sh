#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4l
T=1000 bash tools/gpu_tests.sh || exit 1
for cfg in 5 3; do
timeout -k 10 250 python3 bench.py --config $cfg --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 > gpurun_out/r4l/bench_cfg$cfg.json 2> gpurun_out/r4l/b.log || { tail -3 gpurun_out/r4l/b.log; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r4l/bench_cfg$cfg.json')); print('cfg$cfg', d['value'], d['parity'], d['roofline']['kernels_ms'])"
done
