#!/bin/bash
# cfg 2 at 64 MiB and 1 GiB: the product against the lab builds present, one call
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/cfg2c_${1:-run}.txt
: > $log
run() {
  label=$1; shift
  for bytes in 67108864 1073741824; do
    env "$@" timeout -k 10 300 python3 bench.py --config 2 --bytes $bytes --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/lab/_c2.json 2>> gpurun_out/lab/_c2.err || { tail -5 gpurun_out/lab/_c2.err; exit 1; }
    python3 - "$label" $bytes >> $log <<'PY'
import json, sys
d = json.loads(open("gpurun_out/lab/_c2.json").read().strip().splitlines()[-1])
print("%-28s bytes %s: %.1f GB/s %.4f ms %s %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["parity"], json.dumps(d["roofline"]["kernels_ms"])))
PY
    tail -1 $log
  done
}
for rep in 1 2 3; do
  run "product" AHA_X=1
  for l in aha_amd/libaha_hip_lab_*.so; do run "$(basename $l)" AHA_HIP_LIB=$PWD/$l; done
done
