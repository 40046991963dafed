#!/bin/bash
# bench + rocprofv3 kernel stats of the same command (gpurun): results under gpurun_out/$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-prof}; shift
mkdir -p $out
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-seconds 3 "$@" > $out/bench.json 2> $out/bench.log || { tail -5 $out/bench.log; exit 1; }
python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['roofline']['kernels_ms'], (d.get('cpu_baseline') or {}).get('parity_on_sample'))"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/stats.log || exit 1
python3 - <<PY
import csv, glob
for f in glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:70], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "pct", r["Percentage"])
PY
