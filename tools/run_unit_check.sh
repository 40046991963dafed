cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r3
bash tools/gpu_tests.sh -k "(malformed or random_small or spec or nested or ragged or config3 or engine_selected) and u" || exit 1
timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 > gpurun_out/r3/bench_cfg3.json 2> gpurun_out/r3/bench_cfg3.log
python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_cfg3.json')); print('cfg3', d['value'], d['roofline']['engine'], d['roofline']['kernels_ms'])"
