#!/bin/bash
# GPU test runner for gpurun: verbose progress goes to gpurun_out/tests.log (a silent run is taken to be hung)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 ${T:-900} python -m pytest tests -m gpu -x -v --timeout 300 -p no:cacheprovider "$@" > gpurun_out/tests.log 2>&1
rc=$?
tail -15 gpurun_out/tests.log
exit $rc
