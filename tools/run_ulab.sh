#!/bin/bash
# timing-only variants of the character-level traversal (scan_unit.hip, AHA_UNIT_LAB): what does each part cost?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in ${ULABS:-0 1 2 3 4 5 6 7}; do
  lib=$GRAFT_REPO_ROOT/aha_amd/libaha_hip_ulab$k.so; [ $k = 0 ] && lib=$GRAFT_REPO_ROOT/aha_amd/libaha_hip.so
  AHA_HIP_LIB=$lib AHA_ENGINE=unit timeout 200 python3 tools/exp_unit_uniform.py 2>&1 | grep -a "cfg 3 mix\|rror" | sed "s/^/lab $k: /"
done
for k in ${XGLABS:-}; do
  AHA_HIP_LIB=$GRAFT_REPO_ROOT/aha_amd/libaha_hip_xglab$k.so AHA_ENGINE=unit timeout 200 python3 tools/exp_unit_uniform.py 2>&1 | grep -a "cfg 3 mix\|rror" | sed "s/^/xglab $k: /"
done
