#!/bin/bash
# cfg 5 (1 M keys, hit-dense 256 MiB): bench line, per-kernel times, and the L2 / HBM counters of its kernels.
set -o pipefail
root=$(pwd); out=$root/gpurun_out/${1:-cfg5}; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 280 python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end > "$out/bench.json" 2> "$out/bench.log" || { tail -5 "$out/bench.log"; exit 1; }
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> "$out/stats.log" || exit 1
echo "stats done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/pmc_$name" -- python3 bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> "$out/pmc.log" || exit 1
  echo "pmc $name done"
done
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "traverse" in k or "expand" in k or "count" in k:
        print(k, {c: v[-1] for c, v in d.items()})
PY
