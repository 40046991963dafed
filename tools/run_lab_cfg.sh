#!/bin/bash
# One gpurun leg: bench.py --config $1 over the product library and every lab build present (aha_amd/libaha_hip_lab*.so):
# value, ms per step and the kernels' times of each.  tools/run_lab_cfg.sh <config> <tag>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lab
log=gpurun_out/lab/cfg${1:-5}_${2:-run}.txt
: > $log
for lib in aha_amd/libaha_hip.so aha_amd/libaha_hip_lab*.so; do
  [ -f "$lib" ] || continue
  AHA_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --config ${1:-5} --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end \
      > gpurun_out/lab/_b.json 2>> gpurun_out/lab/_b.err || echo "($lib: bench.py exit $?: a timing-only build fails the parity gate)"
  [ -s gpurun_out/lab/_b.json ] || { tail -5 gpurun_out/lab/_b.err; exit 1; }
  python3 - "$lib" >> $log <<'PY'
import json, sys
d = json.loads(open("gpurun_out/lab/_b.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["unit"], d["ms_per_step"], "ms", d["parity"], json.dumps(d["roofline"]["kernels_ms"]))
PY
  tail -1 $log
done
