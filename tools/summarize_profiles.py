"""Turns gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, re, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]  # e.g. prof_b r01b
src = os.path.join("gpurun_out", tag)
note = ("rocprofv3 --pmc, separate passes with --kernel-trace only, over `python3 bench.py --steps 2 --warmup 1 "
        "--no-cpu-baseline` (cfg3: 100k keys, 1 GiB); median over the FULL-SIZE launches of each kernel (largest grid, at least half the longest duration). FETCH_SIZE/WRITE_SIZE raw units are KB; "
        "gfx950 FETCH_SIZE can under-report wide coalesced streaming reads by 2x (MI355X_MICROARCH.md, HBM) -- "
        "per-lane 16 B strided loads are uncalibrated, so raw values are quoted. SQ_* cycle counters are quad-cycles.")
sys.path.insert(0, os.getcwd())
import subprocess

from bench import lib_fingerprint  # the sources the profiled library was built from

head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "aha_amd/csrc", "include"], capture_output=True,
                            text=True).stdout.strip())
out = {"_note": note, "config": 3, "bytes_per_gpu": 1 << 30, "git_head": head + ("+dirty" if dirty else ""),
       "lib_sha256": lib_fingerprint(), "kernels": {}}


def short(k):
    m = re.search(r"(k[2u]?d?_[a-z_0-9]+)", k)
    return m.group(1) if m else None


for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    # the bench process launches a kernel at several sizes (the timed 1 GiB steps, the 64 MiB ranges of the end-to-end leg,
    # capacity probes): only the FULL-SIZE dispatches -- the largest grid of each kernel -- are averaged
    rows = [r for r in csv.DictReader(open(f)) if "aha::" in r["Kernel_Name"] and short(r["Kernel_Name"])]
    # (a kernel with persistent workgroups has that grid at every size: of the largest-grid dispatches only those that ran
    # at least half as long as the longest one)
    top = collections.defaultdict(int)
    longest = collections.defaultdict(int)
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for r in rows:
        k = short(r["Kernel_Name"])
        top[k] = max(top[k], int(r["Grid_Size"]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if int(r["Grid_Size"]) == top[k]:
            longest[k] = max(longest[k], dur(r))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = short(r["Kernel_Name"])
        if int(r["Grid_Size"]) == top[k] and 2 * dur(r) >= longest[k]:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = sorted(v)
            out["kernels"].setdefault(k, {})[c] = round(v[len(v) // 2], 1)  # median: a capacity probe writes no events
# Calibration on the kernel's own access pattern (tools/calib_fetch.py, profiles/r01f_fetch_calibration.txt): with an
# all-LDS automaton k2_traverse reads exactly the 1 GiB corpus and FETCH_SIZE reports 0.8953e9 bytes (x 0.834: the
# per-lane 16-byte staging loads are tallied partly as half-size requests), WRITE_SIZE 1.06 MB.  The corpus part of
# the traversal's FETCH_SIZE is corrected by that factor; the rest (random 4-byte table loads = 64-byte requests,
# 8-byte event stores) is quoted as counted.
INPUT_REPORTED_FRACTION = 0.8953e9 / (1 << 30)
out["_note"] += (" k2_traverse / ku_traverse (the same 16-byte staging loads): hbm_bytes_per_launch = FETCH_SIZE*1024 + corpus*(1-%.3f) [calibration of the staging "
                 "loads, tools/calib_fetch.py] + WRITE_SIZE*1024." % INPUT_REPORTED_FRACTION)
for k, d in out["kernels"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        corr = out["bytes_per_gpu"] * (1 - INPUT_REPORTED_FRACTION) if k in ("k2_traverse", "ku_traverse") else 0
        d["hbm_bytes_per_launch"] = int((d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 + corr)
json.dump(out, open(f"profiles/{rnd}_pmc.json", "w"), indent=1)
json.dump(out, open("profiles/pmc_traffic_latest.json", "w"), indent=1)
shutil.copy(os.path.join(src, "bench.json"), f"profiles/{rnd}_bench_cfg3.json")
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), f"profiles/{rnd}_bench_under_rocprof.json")
st = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(st[0], f"profiles/{rnd}_rocprofv3_kernel_stats.csv")
# the stats CSV averages over every launch of the process (small parity subsets and capacity probes included): the
# full-size launches -- the timed steps and their warm-ups -- are picked out of the kernel trace by their duration
tr = glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv"))
if tr:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        k = short(r["Kernel_Name"]) if "aha::" in r["Kernel_Name"] else None
        if k:
            per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    u = json.load(open(os.path.join(src, "bench_under_rocprof.json")))
    with open(f"profiles/{rnd}_kernel_trace_full_size.txt", "w") as fo:
        fo.write("# rocprofv3 --kernel-trace of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end`: launches longer than half\n"
                 "# the kernel's longest one (= the full-size launches of the timed steps, warm-ups and parity pass), ms\n"
                 f"# bench.py's own HIP-event average in the same process: {u['roofline']['kernel']} {u['roofline']['avg_ms']} ms\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            big = [x for x in v if x > 0.5 * max(v)]
            fo.write(f"{k:20s} launches {len(v):4d}  full-size {len(big):3d}  mean {sum(big) / len(big):8.4f}  "
                     f"min {min(big):8.4f}  max {max(big):8.4f}\n")
b = json.load(open(os.path.join(src, "bench.json")))
print(b["value"], b["m_hits_per_s"], b["roofline"], b["cpu_baseline"])
print(out["kernels"].get("ku_traverse") or out["kernels"].get("k2_traverse"))
