#!/bin/bash
# cfg 5 check: its parity tests + bench + kernel breakdown
mkdir -p gpurun_out/c5
AHA_TEST_ENGINES=v2,auto,uh,ur timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "config5 or dense or chain or nested or full_size" > gpurun_out/c5/tests.log 2>&1
tail -3 gpurun_out/c5/tests.log
python3 bench.py --config 5 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err &&
python3 -c "
import json
d=json.loads(open('gpurun_out/c5/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'), d['roofline']['kernels_ms'])"
