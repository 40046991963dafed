#!/bin/bash
# round 4: parity suite on the current build, then the profile collection of the headline + the other configurations
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=1000 bash tools/gpu_tests.sh || exit 1
bash tools/collect_round.sh r4h
