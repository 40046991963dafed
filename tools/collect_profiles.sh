#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench JSON, rocprofv3 kernel stats of the same command (without its end-to-end legs: three shards of the group match side by
#   side there, which is not what the timed step does), PMC passes (one counter group per pass,
#   --kernel-trace only -- never combined with other trace domains).
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>/...
set -o pipefail
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 280 python3 bench.py --steps 10 --warmup 3 > "$out/bench.json" 2> "$out/bench.log" || exit 1
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > "$out/bench_under_rocprof.json" 2> "$out/stats.log" || exit 1
echo "stats done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> "$out/pmc.log" || exit 1
  echo "pmc $name done"
done
