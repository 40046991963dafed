#!/bin/bash
# round-3 spot check on the GPU box: a slice of the parity suite, then the cfg 5 and cfg 3 bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3
bash tools/gpu_tests.sh -k "${K:-config5 or chars or nested or random_small or spec or subset}" || exit 1
timeout -k 10 280 python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > gpurun_out/r3/bench_cfg5.json 2> gpurun_out/r3/bench_cfg5.log; python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg5.json')); print('cfg5', d['value'], d['roofline']['kernels_ms'])"
timeout -k 10 280 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/r3/bench_cfg3.json 2> gpurun_out/r3/bench_cfg3.log; python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg3.json')); print('cfg3', d['value'], d['roofline']['kernels_ms'])"
timeout -k 10 280 python3 bench.py --chars --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end > gpurun_out/r3/bench_cfg3_chars.json 2> gpurun_out/r3/bench_cfg3_chars.log; python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg3_chars.json')); print('cfg3 chars', d['value'], d['roofline']['kernels_ms'])"
