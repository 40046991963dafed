#!/bin/bash
# SQ counters of the prefix-filter engine's kernels on cfg 2 at 1 GiB (one counter group per pass, --kernel-trace only)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/pmc_cfg2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$root"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/$name" -- python3 bench.py --config 2 --bytes 1073741824 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> "$out/pmc.log" || exit 1
  echo "pmc $name done"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for n in ("kf_filter", "kf_walk", "k2d_expand", "k2d_count"):
            if n in k and int(r["Grid_Size"]) >= 65536:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as o:
    for n, cs in acc.items():
        # a kernel dispatch reports one row per counter (summed over XCDs/SEs by rocprofv3); the median over the launches
        line = n + ": " + ", ".join(f"{c} {sorted(v)[len(v)//2]:.4g} (n={len(v)})" for c, v in sorted(cs.items()))
        print(line); o.write(line + "\n")
PY
