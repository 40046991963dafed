#!/bin/bash
# a lab build of the library: tools/lab/build_lab.sh <name> <extra compiler flags...>  -> aha_amd/libaha_hip_lab_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../../aha_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -w "$@" --offload-arch=gfx950 -shared -Wl,--version-script=exports.map \
  -o ../libaha_hip_lab_$name.so automaton.cpp cedar_replay.cpp unit.cpp capi.cpp engine.cpp group.cpp kernels.hip scan_v2.hip scan_unit.hip scan_filter.hip scan_skip.hip scan_pair.hip -ldl -lpthread
