#!/bin/bash
# quick perf check of the three configurations (no profiler)
mkdir -p gpurun_out/f3
run() { python3 bench.py "$@" --no-end-to-end --no-cpu-baseline 2> gpurun_out/f3/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d.get('parity'), d['roofline']['kernels_ms'])"; }
run --config 2 --steps 30 --warmup 5
run --config 2 --bytes 1073741824 --steps 20 --warmup 3
run --steps 10 --warmup 3
run --config 5 --steps 5 --warmup 2
