#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ovl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl -- python3 tools/exp_overlap_rebuild.py > gpurun_out/ovl/run.log 2>&1
tail -3 gpurun_out/ovl/run.log
