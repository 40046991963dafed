#!/bin/bash
# One gpurun call that gathers a round's evidence (run from the repo root on the GPU box):
#   tools/collect_profiles.sh <tag>           headline config: bench JSON, rocprofv3 kernel stats, PMC passes
#   + the other configurations as plain bench lines: cfg 2 (64 MiB and 1 GiB), cfg 5, String overload, the two-pass engine
# Usage: tools/collect_round.sh <tag>   -> gpurun_out/<tag>/...
set -o pipefail
tag=${1:-round}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
bash tools/collect_profiles.sh "$tag" || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$root"
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 280 python3 bench.py --no-cpu-baseline "$@" > "$out/$name.json" 2> "$out/$name.log" || { echo "$name failed"; tail -3 "$out/$name.log"; exit 1; }
  echo "$name done"
}
run bench_cfg2_64MiB --config 2 --steps 20 --warmup 3 --no-end-to-end
run bench_cfg2_1GiB --config 2 --bytes 1073741824 --steps 10 --warmup 3 --no-end-to-end
run bench_cfg5 --config 5 --steps 5 --warmup 2 --no-end-to-end
run bench_cfg3_chars --chars --steps 10 --warmup 3 --no-end-to-end
AHA_ENGINE=v1 run bench_cfg3_v1 --steps 3 --warmup 1 --no-end-to-end
AHA_ENGINE=v2 run bench_cfg3_v2 --steps 5 --warmup 2 --no-end-to-end
