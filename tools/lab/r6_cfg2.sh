#!/bin/bash
# cfg 2 (1 000 ASCII keys, engine 5) at its 64 MiB and at 1 GiB: bench.py lines for the filter's size rule (AHA_FILTER_FILL = the fill
# the size search stops at, as a denominator: 256 = round 5's rule)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/cfg2_${1:-run}.txt
: > $log
for fill in 256 64 32; do
  for bytes in 67108864 1073741824; do
    AHA_FILTER_FILL=$fill timeout -k 10 300 python3 bench.py --config 2 --bytes $bytes --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/lab/_c2.json 2>> gpurun_out/lab/_c2.err || { tail -5 gpurun_out/lab/_c2.err; exit 1; }
    python3 - $fill $bytes >> $log <<'PY'
import json, sys
d = json.loads(open("gpurun_out/lab/_c2.json").read().strip().splitlines()[-1])
print("fill 1/%s bytes %s: %.1f GB/s %.4f ms %s %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["parity"], json.dumps(d["roofline"]["kernels_ms"])))
PY
    tail -1 $log
  done
done
