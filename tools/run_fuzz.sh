#!/bin/bash
# differential fuzzer on the GPU box: tools/run_fuzz.sh <seconds> <seed> <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fuzz
timeout -k 10 $(( ${1:-300} + 120 )) python3 tools/fuzz_gpu.py ${1:-300} ${2:-1} > gpurun_out/fuzz/${3:-fuzz}.log 2>&1
rc=$?
tail -5 gpurun_out/fuzz/${3:-fuzz}.log
exit $rc
