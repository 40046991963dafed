#!/bin/bash
# kf_walk's chunk size: cfg 2 at 64 MiB, 256 MiB and 1 GiB with AHA_FILTER_CHUNK = 4096 .. 32768 and with the library's own choice
mkdir -p gpurun_out/f5
for b in 67108864 268435456 1073741824; do
  for c in 4096 8192 16384 32768 auto; do
    if [ $c = auto ]; then unset AHA_FILTER_CHUNK; else export AHA_FILTER_CHUNK=$c; fi
    python3 bench.py --config 2 --bytes $b --steps 30 --warmup 5 --no-end-to-end --no-cpu-baseline 2> gpurun_out/f5/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bytes $b chunk $c', d['value'], 'GB/s', d['ms_per_step'], 'ms', d.get('parity'), r['kernels_ms'])" | tee -a gpurun_out/f5/chunk_sweep.txt
  done
done
