#!/bin/bash
# runs the K1 prototype (ablations + counters) on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/k1
for a in 0 1 2 3; do
  PK_ABL=$a PK_WAVES=16 PK_POW2=0 timeout -k 10 200 python3 tools/proto_k1.py > gpurun_out/k1/w16_a$a.log 2>&1; echo "abl $a rc=$?"; tail -1 gpurun_out/k1/w16_a$a.log
done
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  name=$(echo $grp | cut -d' ' -f2)
  PK_WAVES=16 PK_POW2=0 timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/k1/pmc_$name -- python3 tools/proto_k1.py > /dev/null 2>> gpurun_out/k1/pmc.log || echo "pmc $name failed"
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/k1/pmc_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k1" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, "last launch:", v[-1], "n", len(v))
PY
