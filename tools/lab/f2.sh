#!/bin/bash
# filter engine check: restricted GPU suite + cfg 2 bench at two sizes
mkdir -p gpurun_out/f2
AHA_TEST_ENGINES=f,auto timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/f2/tests.log 2>&1
tail -3 gpurun_out/f2/tests.log
for b in 67108864 1073741824; do
  python3 bench.py --config 2 --bytes $b --steps 20 --warmup 3 --no-end-to-end --no-cpu-baseline > gpurun_out/f2/b_$b.json 2> gpurun_out/f2/b_$b.err &&
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'), d['config'].get('engine'))" gpurun_out/f2/b_$b.json
done
cd /tmp && export TMPDIR=/tmp
for b in 67108864 1073741824; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/f2/prof_$b -- python3 $GRAFT_REPO_ROOT/bench.py --config 2 --bytes $b --steps 20 --warmup 3 --no-end-to-end --no-cpu-baseline > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/f2/prof_$b <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:12]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
