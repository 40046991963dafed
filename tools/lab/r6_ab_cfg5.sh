#!/bin/bash
# cfg 5: the product against the lab builds present (aha_amd/libaha_hip_lab_*.so), one call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/lab
out=gpurun_out/lab/cfg5_ab_${1:-a}.txt; : > $out
for rep in 1 2 3; do
  timeout -k 10 200 python3 tools/lab_cfg.py 5 2>&1 | grep "cfg 5" >> $out || exit 1
  for l in aha_amd/libaha_hip_lab_*.so; do
    AHA_HIP_LIB=$PWD/$l timeout -k 10 200 python3 tools/lab_cfg.py 5 2>&1 | grep "cfg 5" >> $out || exit 1
  done
done
cat $out
