#!/bin/bash
# the prefix-filter engine against the byte-level engine over the number of keys (cfg 2's generator, 256 MiB)
mkdir -p gpurun_out/f4
for k in 1000 2000 3000 10000 30000; do
  for e in auto v2; do
    if [ $e = v2 ]; then export AHA_ENGINE=v2; else unset AHA_ENGINE; fi
    python3 bench.py --config 2 --keys $k --bytes 268435456 --steps 10 --warmup 2 --no-end-to-end --no-cpu-baseline 2> gpurun_out/f4/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('keys $k', '$e', 'engine', r['engine'], d['value'], 'GB/s', d['ms_per_step'], 'ms', d.get('parity'), 'slots', d['config']['slots'], 'hits', d['config']['hits_per_gpu'], r['kernels_ms'])" | tee -a gpurun_out/f4/keys_sweep.txt
  done
done
