#!/bin/bash
# Lab build: the product sources with every `v_cndmask_b32_e32 ..., vcc` of scan_unit.hip's device code re-encoded as VOP3
# (`v_cndmask_b32_e64 ..., vcc`) in the ASSEMBLY -- same instructions, same count, only the encoding of the selects changes.
# Prices the review's "e32 select on an SALU-written VCC costs 12 cycles" on the product trip without the 6 extra instructions
# the source-level form (asm select on a ballot mask) needs.  Output: aha_amd/libaha_hip_lab_e64.so
set -e
cd "$(dirname "$0")/../../aha_amd/csrc"
LLVM=/opt/rocm/lib/llvm/bin
FL="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-parentheses -Wno-bitwise-instead-of-logical $EXTRA"
T=$(mktemp -d)
/opt/rocm/bin/hipcc $FL --offload-arch=gfx950 -I ../../include -S --cuda-device-only -o $T/su.s scan_unit.hip
n=$(grep -c 'v_cndmask_b32_e32 .*, vcc$' $T/su.s || true)
sed -E -i 's/v_cndmask_b32_e32 (.*), vcc$/v_cndmask_b32_e64 \1, vcc/' $T/su.s
echo "re-encoded $n selects"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/su.s -o $T/su.dev.obj
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/su.out $T/su.dev.obj
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/su.out -output=$T/su.hipfb
/opt/rocm/bin/hipcc $FL --offload-arch=gfx950 -I ../../include --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/su.hipfb -c scan_unit.hip -o $T/su_host.o
for f in automaton.cpp cedar_replay.cpp unit.cpp capi.cpp engine.cpp group.cpp kernels.hip scan_v2.hip scan_filter.hip; do
  /opt/rocm/bin/hipcc $FL --offload-arch=gfx950 -I ../../include -c $f -o $T/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=exports.map -o ../libaha_hip_lab_${OUT:-e64}.so $T/*.o -ldl -lpthread
rm -rf $T
