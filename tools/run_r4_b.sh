#!/bin/bash
# round 4: parity suite on the two-walk kernel, bench, lab variants; then the one-walk kernel's bench for comparison
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
T=600 bash tools/gpu_tests.sh || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.log || { tail -5 gpurun_out/r4b/bench.log; exit 1; }
cat gpurun_out/r4b/bench.json
bash tools/run_lab_unit.sh r4b || exit 1
AHA_UNIT_WALKS=1 AHA_LAB_NOTE=one-walk timeout -k 10 120 python3 tools/lab_unit.py >> gpurun_out/lab/unit_lab_r4b.txt 2>&1 || exit 1
tail -1 gpurun_out/lab/unit_lab_r4b.txt
