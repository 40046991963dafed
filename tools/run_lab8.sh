#!/bin/bash
# Lab, second question: does occupancy pay?  The traversal kernel held to 64 VGPRs (libaha_hip_diag8.so) with one
# and with two 1024-thread workgroups per CU (AHA_V2_BPC=2: half the LDS each, 8 waves per SIMD).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
export AHA_LAB_LIB=$GRAFT_REPO_ROOT/aha_amd/libaha_hip_diag8.so
run() {  # name, env BPC, args
  local name=$1 bpc=$2; shift; shift
  AHA_V2_BPC=$bpc timeout -k 10 240 python3 tools/lab_traverse.py --out gpurun_out/lab/$name.txt "$@" > gpurun_out/lab/$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/lab/$name.log; exit 1; }
  grep -v "^#" gpurun_out/lab/$name.txt | head -8
}
echo "== cfg 3, 64 VGPRs, 1 workgroup per CU"; run occ_cfg3_bpc1 1 --knobs 0,1,3,4
echo "== cfg 3, 64 VGPRs, 2 workgroups per CU"; run occ_cfg3_bpc2 2 --knobs 0,1,3,4
echo "== cfg 2 with 500 keys (all LDS also at half the LDS), 64 VGPRs, 1 workgroup per CU"; run occ_cfg2_bpc1 1 --config 2 --keys 500 --knobs 0
echo "== cfg 2 with 500 keys (all LDS also at half the LDS), 64 VGPRs, 2 workgroups per CU"; run occ_cfg2_bpc2 2 --config 2 --keys 500 --knobs 0
