#!/bin/bash
# round-3 spot check on the GPU box: the parity suite (or a slice: K=...), then bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3
bash tools/gpu_tests.sh ${K:+-k "$K"} || exit 1
b() {  # name, args
  local name=$1; shift
  timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-end-to-end "$@" > gpurun_out/r3/$name.json 2> gpurun_out/r3/$name.log
  python3 -c "
import json; d=json.load(open('gpurun_out/r3/$name.json')); print('$name', d['value'], d['roofline']['engine'], d['roofline']['kernels_ms'])"
}
b bench_cfg3 --steps 10 --warmup 3
AHA_ENGINE=v2 b bench_cfg3_v2 --steps 10 --warmup 3
b bench_cfg5 --config 5 --steps 5 --warmup 2
b bench_cfg2 --config 2 --steps 20 --warmup 3
