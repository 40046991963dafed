#!/bin/bash
# round 4, first GPU call: parity suite, bench line, lab variants
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4a
T=1000 bash tools/gpu_tests.sh || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-seconds 3 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.log || { tail -5 gpurun_out/r4a/bench.log; exit 1; }
cat gpurun_out/r4a/bench.json
bash tools/run_lab_unit.sh r4a
