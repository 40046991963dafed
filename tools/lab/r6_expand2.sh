#!/bin/bash
# cfg 5's dense expansion: blocks per launch (AHA_EXPAND_BLOCKS), one call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/lab
out=gpurun_out/lab/expand_blocks_${1:-a}.txt; : > $out
for rep in 1 2; do
  for b in 1000000 81920 40960 20480 10240 5120; do
    AHA_EXPAND_BLOCKS=$b AHA_LAB_NOTE="blocks $b" timeout -k 10 200 python3 tools/lab_cfg.py 5 2>&1 | grep "cfg 5" >> $out || exit 1
  done
  AHA_HIP_LIB=$PWD/aha_amd/libaha_hip_lab_old.so timeout -k 10 200 python3 tools/lab_cfg.py 5 2>&1 | grep "cfg 5" >> $out || exit 1
done
cat $out
