#!/bin/bash
# round 5: parity suite on the current build, the profile collection of the headline + the other configurations, the 8-GPU
# step's kernels on this one GPU, and a fuzz run.  tools/run_r5.sh <tag> <fuzz seed>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r5}
T=1000 bash tools/gpu_tests.sh || exit 1
bash tools/collect_round.sh $tag || exit 1
(timeout -k 10 300 python3 tools/exp_overlap_rebuild.py --separate-pack; timeout -k 10 300 python3 tools/exp_overlap_rebuild.py) > gpurun_out/$tag/overlap_rebuild.txt 2>&1 || { tail -5 gpurun_out/$tag/overlap_rebuild.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/$tag/overlap_rebuild.txt | tail -8
bash tools/run_fuzz.sh 200 ${2:-88001} fuzz_$tag
