#!/bin/bash
# cfg 2 at 64 MiB (and 1 GiB): bench.py lines for lab switches of the call's fixed costs, one call
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/cfg2b_${1:-run}.txt
: > $log
run() {  # label, env...
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
for rep in 1 2; do
  run "product" AHA_X=1
  run "memset per call" AHA_CURSOR_MEMSET=1
  run "doc offsets own launch" AHA_DOC_OFFSETS_LAUNCH=1
done
