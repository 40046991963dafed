#!/bin/bash
# Two bench ranks sharing the one GPU of the development box over gloo (RCCL refuses two ranks on one device):
# exercises the N>1 code path of bench.py end to end -- weak leg, breakdown, strong-scaling leg.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export AHA_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 ${T:-500} python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --bytes $((1 << 28)) "$@" > gpurun_out/two_ranks.json 2> gpurun_out/two_ranks.log
rc=$?
tail -5 gpurun_out/two_ranks.log
cat gpurun_out/two_ranks.json
exit $rc
