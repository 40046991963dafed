#!/usr/bin/env python3
"""lab_traverse.py -- where does a wave-trip of k2_traverse spend its time?  (round 3)

Runs the LAB build of the library (`make -C aha_amd/csrc diag` -> aha_amd/libaha_hip_diag.so; the product never
loads it) on the headline workload (cfg 3: 100k keys, 1 GiB) and times the traversal kernel alone, with HIP events
inside the library, for every diagnostic variant of scan_v2.hip (kDg*):

  * subtractive, timing only (the walk is wrong on purpose, nothing reads the results): no far probes, half of the
    far probes, no event stores, far probes answered by L1;
  * additive, exact walk (hits compared with the product variant): +1 far load per far lane, +16 VALU per trip,
    +1 LDS read per trip, plain instead of non-temporal event stores, ds_read + global_load instead of flat_load;
  * s_memtime stamps around the segments of the trip, split by "the wave-trip had a far lane" or not.

Usage (GPU box, repo root):  python3 tools/lab_traverse.py [--bytes N] [--out gpurun_out/lab/trip_anatomy.txt]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AHA_HIP_LIB"] = os.environ.get("AHA_LAB_LIB") or os.path.join(ROOT, "aha_amd", "libaha_hip_diag.so")

import numpy as np  # noqa: E402
import torch  # noqa: E402

KNOBS = [
    (0, "product kernel", "exact"),
    (11, "plain (not non-temporal) event stores", "exact"),
    (16, "event stores really non-temporal (buffer store, aux nt)", "exact"),
    (9, "ds_read (near) + global_load (far) instead of one flat_load", "exact"),
    (12, "split loads, far load non-temporal (nt)", "exact"),
    (13, "split loads, far load sc1 (L2-served, no L1 allocation)", "exact"),
    (14, "split loads, far load 8 bytes wide", "exact"),
    (17, "trip without the NUL contract's instructions (text has no NUL)", "exact"),
    (18, "+ state kept as its whole entry (base + fail-is-root flag)", "exact"),
    (19, "+ ds_read slot[min(idx,T)] and masked global_load", "exact"),
    (5, "+1 independent far load per far lane", "exact"),
    (6, "+16 dependent VALU per trip", "exact"),
    (7, "+1 random ds_read_b32 per trip", "exact"),
    (3, "no event stores", "timing"),
    (1, "no far probes (answered from LDS)", "timing"),
    (15, "40 % of the far probes answered from LDS as misses", "timing"),
    (2, "half of the far probes answered from LDS", "timing"),
    (10, "far probes answered by an 8 KiB window (L1 hits)", "timing"),
    (4, "no far probes, no event stores", "timing"),
    (8, "s_memtime stamps", "exact"),
]
SEG = ["A issue + LDS round trip", "B selects without the probe", "C wait for the probe (vmcnt 0)",
       "D selects on the probe", "E event store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5])
    ap.add_argument("--keys", type=int, default=None)
    ap.add_argument("--knobs", default=None, help="comma-separated variant numbers (default: all)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lab", "trip_anatomy.txt"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    lines = []

    def say(s=""):
        print(s, flush=True)
        lines.append(s)
        with open(args.out, "w") as f:  # partial results survive a killed run
            f.write("\n".join(lines) + "\n")

    from aha_amd import AC, AhaError, synth
    from aha_amd import _native as N

    L = N.lib()
    L.aha_diag_set.argtypes = [C.c_int]
    L.aha_diag_set.restype = None
    L.aha_diag_read_stamps.argtypes = [C.c_void_p, C.c_uint64]
    L.aha_diag_read_stamps.restype = C.c_int
    dev = torch.device("cuda", 0)
    t0 = time.time()
    blob, offs, nf = synth.keys(args.config, K=args.keys)
    corpus, doc = synth.corpus(args.config, blob, offs, nf, n_bytes=args.bytes)
    D = doc.size - 1
    ac = AC.compile_packed(blob, offs, device=0)
    ac.set_profiling(True)
    info = ac.info
    say(f"# cfg {args.config}: {offs.size - 1} keys, {args.bytes} bytes, {D} documents; image {info['n_slots']} slots, "
        f"{info['lds_slots']} in LDS; lib {os.path.basename(os.environ['AHA_HIP_LIB'])}, "
        f"AHA_V2_BPC={os.environ.get('AHA_V2_BPC', '1')}; setup {time.time() - t0:.1f}s")
    d_corpus = torch.from_numpy(corpus).to(dev)
    d_doc = torch.from_numpy(doc.astype(np.int64)).to(dev)
    d_dho = torch.zeros(D + 1, dtype=torch.int64, device=dev)
    L.aha_diag_set(0)
    try:
        n_hits = ac.match_batch_device(d_corpus, d_doc, torch.zeros((1, 3), dtype=torch.int32, device=dev), d_dho)
    except AhaError as e:
        if e.code != N.AHA_E_CAPACITY:
            raise
        n_hits = e.required
    d_out = torch.zeros((n_hits + 1024, 3), dtype=torch.int32, device=dev)
    ref = None
    base_ms = None
    say(f"# {n_hits} hits per pass; traversal time = HIP events inside the library (ms_count), median of {args.steps}")
    say(f"{'variant':<62} {'walk':<7} {'traverse ms':>11} {'vs product':>10}  check")
    want = None if args.knobs is None else [int(x) for x in args.knobs.split(",")]
    for knob, name, kind in KNOBS:
        if want is not None and knob not in want:
            continue
        L.aha_diag_set(knob)
        if kind == "timing":
            os.environ["AHA_DIAG_TRAVERSE_ONLY"] = "1"
        else:
            os.environ.pop("AHA_DIAG_TRAVERSE_ONLY", None)
        ms = []
        n = 0
        for i in range(2 + args.steps):
            d_out.zero_() if (kind == "exact" and i == 0) else None
            n = ac.match_batch_device(d_corpus, d_doc, d_out, d_dho)
            if i >= 2:
                ms.append(ac.last_timing()["ms_count"])
        torch.cuda.synchronize()
        med = float(np.median(ms))
        check = "-"
        if kind == "exact":
            if ref is None:
                ref = d_out[:n].clone()
                check = "reference"
            else:
                check = "identical" if (n == ref.shape[0] and torch.equal(d_out[:n], ref)) else "DIFFERENT"
        if knob == 0:
            base_ms = med
        say(f"{name:<62} {kind:<7} {med:11.3f} {med / base_ms:10.3f}  {check}")
        if knob == 8:
            waves = 256 * 16
            buf = np.zeros(waves * 16, dtype=np.uint32)
            rc = L.aha_diag_read_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
            assert rc == 0, rc
            st = buf.reshape(waves, 16).astype(np.float64)
            say()
            say("# stamps of the last launch: shader cycles per wave-trip, mean over all waves (lane 0's view of its wave); the")
            say("# stamped build is slower than the product (each stamp drains lgkmcnt): read the shares, not the length")
            for cls, o in (("wave-trips without a far lane", 0), ("wave-trips with a far lane", 6)):
                cnt = st[:, o + 5].sum()
                if cnt == 0:
                    say(f"{cls}: none")
                    continue
                tot = st[:, o:o + 5].sum()
                say(f"{cls}: {int(cnt)} wave-trips ({cnt / (st[:, 5].sum() + st[:, 11].sum()) * 100:.1f} %), "
                    f"{tot / cnt:.0f} cycles per wave-trip inside the stamps")
                for k in range(5):
                    v = st[:, o + k].sum()
                    say(f"    {SEG[k]:<34} {v / cnt:8.1f} cycles  {v / tot * 100:5.1f} %")
            say()
    L.aha_diag_set(0)
    os.environ.pop("AHA_DIAG_TRAVERSE_ONLY", None)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
