#!/bin/bash
# PMC counters of one kernel of the bench (gpurun): tools/pmc_kernel.sh <tag> <kernel substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-pmc}; kern=${2:-k2_traverse}; shift; shift
mkdir -p $out
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>> $out/pmc.log || echo "pmc $name failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, v[-1])
PY
