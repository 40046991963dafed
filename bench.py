#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native Aha::AC#match path.

Metric (BASELINE.json): input GB/s scanned + M-hits/s, 100k-pattern
Aho-Corasick over a 1 GiB UTF-8 corpus (config 3: "100k patterns, 1 GiB
synthetic UTF-8 corpus, 1xMI355X").  A "step" is one pass of the hot path
(aha_ac_match_batch_device: corpus and automaton resident in HBM -> ordered
hit triples + per-document offsets final in HBM) over one batch.

  python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scans
its own 1 GiB corpus (config 4: seed + rank; weak scaling, documents never
span ranks) and the hit buffers are exchanged with an all-gatherv over
RCCL/xGMI (aha_amd/distributed.py).  value = bytes scanned by all ranks / max
over ranks of the timed region.

One JSON line is printed by rank 0; it carries `roofline` (dominant kernel,
HIP-event timed inside the library on the launch stream) and `cpu_baseline`
(the C oracle -- a restatement of the reference's CPU path -- timed on this
box's host cores on a bounded sample, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5], help="BASELINE.json config (3 = headline)")
    ap.add_argument("--bytes", type=int, default=None, help="override corpus bytes per GPU")
    ap.add_argument("--keys", type=int, default=None, help="override number of keys")
    ap.add_argument("--chars", action="store_true", help="String overload (char offsets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-entry / download legs after the timed run")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="bound of the CPU baseline leg (five passes)")
    ap.add_argument("--parity-bytes", type=int, default=32 << 20,
                    help="bytes of the first documents compared with the oracle on every run (the parity gate)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "words", "packed", "triples"],
                    help="N>1 payload of the all-gatherv: the 4-byte stream (value | step of end; default below 2^20 "
                         "keys), {end,value} pairs (8 B per hit), or the 12-byte Hit triples themselves; packed forms "
                         "are rebuilt into triples on arrival, inside the timed step")
    ap.add_argument("--gather", default="allgatherv", choices=["allgatherv", "none"],
                    help="N>1: exchange hit buffers over RCCL inside the timed step")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1: wait for each step's all-gatherv before the next match (default: the exchange of "
                         "step i runs beside the match of step i+1, double-buffered)")
    ap.add_argument("--force-wide", action="store_true")
    ap.add_argument("--no-strong", action="store_true",
                    help="N>1: skip the strong-scaling leg (the cfg 3 corpus split over the ranks) after the timed run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 collective backend; gloo (hits staged through host memory) only to rehearse the "
                         "multi-rank flow on a 1-GPU box together with AHA_BENCH_ONE_DEVICE=1")
    return ap.parse_args()


def lib_fingerprint():
    """sha256 over the sources libaha_hip.so is built from: PMC traffic files are only valid for the build they were
    measured on (a rebuilt binary need not be bit-identical, its sources are)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "aha_amd", "csrc", "*")) + [os.path.join(ROOT, "include", "aha_hip.h")])
    for f in files:
        if os.path.isfile(f) and not f.endswith((".o", ".so")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def parity_sample(blob, offs, corpus, doc, gpu_hits, gpu_dho, chars, max_bytes, log):
    """Compares the GPU hits and per-document offsets of the first documents (up to max_bytes) with the oracle.
    Returns "bit-exact" / "MISMATCH".  The oracle is the checker here, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc

    o = orc.AC.compile_packed(blob, offs)
    D = doc.size - 1
    d1 = max(1, int(np.searchsorted(doc, max_bytes, side="right")) - 1)
    d1 = min(d1, D)
    sub = doc[:d1 + 1] - doc[0]
    seg = corpus[:int(doc[d1])]
    oh, od = o.match_batch(seg, sub, cap=max(1024, seg.size // 4), chars=chars)  # grows on demand
    b = int(gpu_dho[d1])
    ok = b == len(oh) and gpu_hits[:b].tobytes() == oh.tobytes() and np.array_equal(gpu_dho[:d1 + 1], od)
    log(f"parity sample: {d1} documents, {seg.size} bytes, {len(oh)} hits: {'bit-exact' if ok else 'MISMATCH'}")
    return "bit-exact" if ok else "MISMATCH"


def cpu_baseline(blob, offs, corpus, doc, gpu_hits, gpu_dho, seconds, log):
    """Times the oracle (C restatement of ac.cr:176-192,265-286 over reference-layout arrays) on a bounded sample of
    the same workload: the first documents up to 256 MiB (less when `seconds` would not cover five passes), FIVE passes,
    the median (SURVEY.md section 8 d); the first pass also checks the GPU hits on that sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc  # checker / reported baseline only

    t0 = time.time()
    o = orc.AC.compile_packed(blob, offs)
    t_compile = time.time() - t0
    D = doc.size - 1
    # calibration: one document tells the rate; size the sample so that five passes fit `seconds`
    d_cal = 1
    t0 = time.time()
    o.match_batch(corpus[:int(doc[d_cal])], doc[:d_cal + 1] - doc[0], cap=max(1024, int(doc[d_cal]) // 4))
    rate = max(int(doc[d_cal]), 1) / max(time.time() - t0, 1e-6)
    want = int(min(256 << 20, rate * seconds / 5))
    done_docs = max(1, min(D, int(np.searchsorted(doc, want, side="right")) - 1))
    done_bytes = int(doc[done_docs])
    sub = doc[:done_docs + 1] - doc[0]
    seg = corpus[:done_bytes]
    exact, times, n_hits = True, [], 0
    for p in range(5):
        t0 = time.time()
        oh, od = o.match_batch(seg, sub, cap=max(1024, seg.size // 4))
        times.append(time.time() - t0)
        n_hits = len(oh)
        if p == 0 and gpu_hits is not None:
            b = int(gpu_dho[done_docs])
            exact = (b == len(oh)) and gpu_hits[:b].tobytes() == oh.tobytes() and \
                np.array_equal(gpu_dho[:done_docs + 1], od)
    t_match = float(np.median(times))
    log(f"cpu baseline: {done_bytes / 1e6:.1f} MB in {t_match:.2f}s, {n_hits} hits, compile {t_compile:.2f}s")
    # informational: the same loop on every host core, documents statically sharded;
    # ctypes releases the GIL, so plain threads run the C oracle in parallel
    mt = None
    try:
        import concurrent.futures as cf

        cores = max(1, min(16, len(os.sched_getaffinity(0))))  # a 1-GPU box share is 16 host cores
        per = max(1, min(D, 256) // cores)
        shards = [(i * per, min(D, (i + 1) * per)) for i in range(cores) if i * per < D]

        def run(sh):
            lo, hi = sh
            sub = doc[lo:hi + 1] - doc[lo]
            seg = corpus[int(doc[lo]):int(doc[hi])]
            oh, _ = o.match_batch(seg, sub, cap=max(1024, seg.size // 4))
            return seg.size, len(oh)

        t0 = time.time()
        with cf.ThreadPoolExecutor(max_workers=len(shards)) as ex:
            res = list(ex.map(run, shards))
        dt = time.time() - t0
        mt = {"value": round(sum(r[0] for r in res) / dt / 1e9, 4), "unit": "GB/s", "cores": len(shards),
              "sample": f"{sum(r[0] for r in res)} bytes, documents sharded over {len(shards)} threads"}
        log(f"cpu baseline, {len(shards)} threads: {mt['value']} GB/s")
    except Exception as ex:  # informational only
        log(f"multi-thread cpu baseline skipped: {ex}")
    return {
        "value": round(done_bytes / t_match / 1e9, 4),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {done_docs} of {D} documents ({done_bytes} bytes) of the same corpus, 1 thread, median of 5 "
                  f"passes, C restatement of the reference CPU path (oracle/aha_oracle.c), Bytes overload",
        "passes_s": [round(t, 3) for t in times],
        "m_hits_per_s": round(n_hits / t_match / 1e6, 3),
        "parity_on_sample": "bit-exact" if exact else "MISMATCH",
        "all_cores": mt,
    }, exact


def launch_ranks(n):
    """One child process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same command line),
    rendezvous on 127.0.0.1.  Rank 0's stdout (the JSON line) goes to this process's stdout.  Returns the exit code: non-zero
    when a rank fails (the others are stopped) or prints no line."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    deadline = time.time() + 3600
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed or time.time() > deadline:  # a rank died (or the job hangs): the rest would wait for it in a collective
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            rc = 1
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or (p.returncode or 0)
    reader.join(timeout=30)
    out = b"".join(chunks).decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    if not rc and not any(l.startswith("{") for l in out.splitlines()):
        rc = 1
    return rc


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        # plain `python bench.py --gpus N` (no torchrun): this process starts the N ranks itself -- fresh children, before it
        # has made any GPU call -- and passes rank 0's JSON line on
        sys.exit(launch_ranks(args.gpus))
    if os.environ.get("AHA_BENCH_LAUNCH_TEST"):  # (tests/test_host_logic.py: the launcher alone, no GPU)
        if rank == 0:
            print(json.dumps({"launch_test": True, "world": world, "gpus": args.gpus,
                              "master": os.environ.get("MASTER_ADDR", "") + ":" + os.environ.get("MASTER_PORT", "")}), flush=True)
        sys.exit(3 if os.environ.get("AHA_BENCH_LAUNCH_TEST") == f"fail{rank}" else 0)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if os.environ.get("AHA_BENCH_ONE_DEVICE"):  # rehearsal of the N>1 path on a 1-GPU box (ranks share cuda:0)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective payloads live

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    from aha_amd import AC, AhaError, synth
    from aha_amd import _native as N

    cfg = args.config
    if world > 1 and args.gather != "none" and not args.no_overlap:
        # the traversal is a persistent grid that fills every CU's LDS: keep a few CUs free so the RCCL
        # kernels of the overlapped exchange are not locked out until it ends (read at compile time)
        os.environ.setdefault("AHA_RESERVE_CUS", "16")
    t0 = time.time()
    blob, offs, nf = synth.keys(cfg, K=args.keys)
    K = offs.size - 1
    n_bytes = args.bytes if args.bytes else synth.DEFAULT_BYTES[cfg]
    corpus, doc = synth.corpus(cfg, blob, offs, nf, n_bytes=n_bytes, rank=rank)
    D = doc.size - 1
    log(f"generated cfg{cfg}: {K} keys, {n_bytes} bytes, {D} docs in {time.time() - t0:.1f}s")
    t0 = time.time()
    ac = AC.compile_packed(blob, offs, device=local_rank, force_wide=args.force_wide)
    info = ac.info
    compile_s = time.time() - t0
    log(f"compiled in {compile_s:.2f}s: {info}")
    ac.set_profiling(True)

    t0 = time.time()
    d_corpus = torch.from_numpy(corpus).to(dev)
    d_doc = torch.from_numpy(doc.astype(np.int64)).to(dev)
    torch.cuda.synchronize()
    t_upload = time.time() - t0
    d_dho = torch.zeros(D + 1, dtype=torch.int64, device=dev)
    # size the hit buffer with one untimed call
    try:
        n_hits = ac.match_batch_device(d_corpus, d_doc, torch.zeros((1, 3), dtype=torch.int32, device=dev), d_dho,
                                       chars=args.chars)
    except AhaError as e:
        if e.code != N.AHA_E_CAPACITY:
            raise
        n_hits = e.required
    d_out = torch.zeros((n_hits + 1024, 3), dtype=torch.int32, device=dev)
    d_outs = [d_out]
    log(f"{n_hits} hits per pass ({n_hits / n_bytes:.4f} per byte); upload {t_upload:.2f}s "
        f"({n_bytes / t_upload / 1e9:.1f} GB/s PCIe-inclusive)")

    gather = None
    if world > 1 and args.gather == "allgatherv":
        from aha_amd.distributed import HitGatherer

        if args.exchange == "auto":
            args.exchange = "words" if K <= (1 << 20) else "packed"
        xmode = {"words": "words", "packed": "pairs", "triples": "triples"}[args.exchange]
        gather = HitGatherer(dist, cdev, ac=ac, exchange=xmode, chars=args.chars)

    overlap = gather is not None and not args.no_overlap
    if overlap:
        d_outs.append(torch.zeros_like(d_out))
    step_no = [0]

    def step():
        i = step_no[0]
        step_no[0] += 1
        if not overlap:
            n = ac.match_batch_device(d_corpus, d_doc, d_out, d_dho, chars=args.chars)
            if gather is not None:
                gather.all_gatherv(d_out if cdev == dev else d_out[:n].cpu(), n)
            return n
        slot = i & 1
        gather.finish(slot)  # the exchange that last read d_outs[slot] (step i-2) must be done
        n = ac.match_batch_device(d_corpus, d_doc, d_outs[slot], d_dho, chars=args.chars)
        gather.start(d_outs[slot] if cdev == dev else d_outs[slot][:n].cpu(), n, slot)  # beside the next match
        return n

    def drain():
        if overlap:
            gather.finish(0)
            gather.finish(1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    kern = {"ms_total": [], "ms_count": [], "ms_scan": [], "ms_write": [], "ms_aux": []}
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = ac.last_timing()
        for k in kern:
            kern[k].append(t[k])
    drain()  # every exchange issued in the timed region completes inside it
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    tot_bytes = torch.tensor([float(n_bytes)], dtype=torch.float64, device=cdev)
    tot_hits = torch.tensor([float(n_hits)], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_bytes)
        dist.all_reduce(tot_hits)
    elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    gbs = float(tot_bytes.item()) * args.steps / elapsed / 1e9
    mhits = float(tot_hits.item()) * args.steps / elapsed / 1e6

    # ---- the same K steps once more WITHOUT the library's timing events (the product's default; `value` above is measured with
    # them, as the roofline needs their times from the timed region): five event records cost a call ~20 us of wall time,
    # which shows on steps of a tenth of a millisecond (cfg 2 at 64 MiB) and nowhere else.  Reported beside, never as `value`.
    no_events = None
    if world == 1:
        ac.set_profiling(False)
        for _ in range(min(args.warmup, 2)):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        e1 = time.perf_counter() - t1
        ac.set_profiling(True)
        step()  # (last_timing() below is a profiled call's again)
        no_events = {"ms_per_step": round(e1 / args.steps * 1e3, 4), "value": round(n_bytes * args.steps / e1 / 1e9, 3),
                     "note": "the same steps with aha_ac_set_profiling off (no HIP event records inside the call)"}
        log(f"without the library's timing events: {no_events['ms_per_step']} ms per step, {no_events['value']} GB/s")

    # ---- N > 1: the parts of a step on their own (SURVEY.md section 8 d: "gather time broken out"), untimed extras
    breakdown, strong = None, None
    if gather is not None:
        k2 = max(2, min(args.steps, 5))

        def timed(fn):
            fence()
            t0 = time.perf_counter()
            for _ in range(k2):
                fn()
            fence()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item()) / k2 * 1e3

        scan_ms = timed(lambda: ac.match_batch_device(d_corpus, d_doc, d_out, d_dho, chars=args.chars))
        exch_ms = timed(lambda: gather.all_gatherv(d_out if cdev == dev else d_out[:n_hits].cpu(), n_hits))
        breakdown = {"scan_only_ms": round(scan_ms, 4), "exchange_only_ms": round(exch_ms, 4),
                     "step_ms": round(ms_per_step, 4), "overlapped": bool(overlap), "exchange": args.exchange,
                     "wire_bytes_per_hit": round(4.0 * gather.last_payload_elems / max(n_hits, 1), 3),
                     "wire_bytes_per_peer": int(4 * gather.last_payload_elems)}
        log(f"breakdown: scan only {scan_ms:.3f} ms, exchange only {exch_ms:.3f} ms, step {ms_per_step:.3f} ms")
    if gather is not None and not args.no_strong:
        # ---- strong scaling: ONE corpus (rank 0's) cut into contiguous byte-balanced document ranges
        from aha_amd.distributed import HitGatherer, local_shard, stream_digest, strong_scaling_pass

        c0, doc0 = (corpus, doc) if rank == 0 else synth.corpus(cfg, blob, offs, nf, n_bytes=n_bytes, rank=0)
        sub, sub_doc, _lo = local_shard(c0, doc0, rank, world)
        s_corpus = torch.from_numpy(np.ascontiguousarray(sub)).to(dev)
        s_doc = torch.from_numpy(sub_doc.astype(np.int64)).to(dev)
        s_dho = torch.zeros(sub_doc.size, dtype=torch.int64, device=dev)
        try:
            sn = ac.match_batch_device(s_corpus, s_doc, torch.zeros((1, 3), dtype=torch.int32, device=dev), s_dho,
                                       chars=args.chars)
        except AhaError as e:
            if e.code != N.AHA_E_CAPACITY:
                raise
            sn = e.required
        s_out = torch.zeros((sn + 1024, 3), dtype=torch.int32, device=dev)
        sg = HitGatherer(dist, cdev, ac=ac, exchange=xmode, chars=args.chars)

        def match_fn(c, d):
            n = ac.match_batch_device(c, d, s_out, s_dho, chars=args.chars)
            if cdev == dev:
                return s_out, n, s_dho
            return s_out[:n].cpu(), n, s_dho.cpu()

        sync = torch.cuda.synchronize
        allh, alld, _ = strong_scaling_pass(sg, match_fn, s_corpus, s_doc, sync)  # warm-up
        fence()
        t0 = time.perf_counter()
        tm = tx = 0.0
        for _ in range(k2):
            allh, alld, (a, b) = strong_scaling_pass(sg, match_fn, s_corpus, s_doc, sync)
            tm += a
            tx += b
        fence()
        ts = torch.tensor([time.perf_counter() - t0, tm, tx], dtype=torch.float64, device=cdev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        el, tm, tx = [float(x) for x in ts.tolist()]
        verified = None
        if rank == 0:  # the same corpus on one GPU: the gathered stream must be identical
            f_corpus = torch.from_numpy(c0).to(dev)
            f_doc = torch.from_numpy(doc0.astype(np.int64)).to(dev)
            f_dho = torch.zeros(doc0.size, dtype=torch.int64, device=dev)
            f_out = torch.zeros((int(allh.shape[0]) + 1024, 3), dtype=torch.int32, device=dev)
            fn_ = ac.match_batch_device(f_corpus, f_doc, f_out, f_dho, chars=args.chars)
            verified = stream_digest(f_out[:fn_].cpu().numpy(), f_dho.cpu().numpy()) == \
                stream_digest(allh.cpu().numpy(), alld.cpu().numpy())
            del f_corpus, f_out
        strong = {"scaling": "strong", "bytes_total": int(n_bytes), "value": round(n_bytes * k2 / el / 1e9, 3), "unit": "GB/s",
                  "ms_per_step": round(el / k2 * 1e3, 4), "scan_ms": round(tm / k2 * 1e3, 4),
                  "exchange_ms": round(tx / k2 * 1e3, 4), "steps": k2, "hits_total": int(allh.shape[0]),
                  "identical_to_one_gpu": verified}
        log(f"strong scaling: {strong}")

    # roofline of the dominant kernel (HIP events inside the library, launch stream)
    avg = {k: float(np.mean(v)) for k, v in kern.items()}
    engine = ac.last_timing()["engine"]
    A = info["image_bytes"]
    if engine == 4:
        # ku_traverse (character-level image): corpus + doc offsets + the unit image in (8-byte slots -- the fail headers
        # are among them since round 4 --, the root table and the decode tables); its event records are scratch
        dom, dom_ms = "ku_traverse", avg["ms_count"]
        A = info["unit_slots"] * 8 + info["unit_syms"] * 4 + 11264
        alg_bytes = n_bytes + 8 * (D + 1) + A
    elif engine == 5:
        # prefix-filter engine: kf_filter reads every byte of the corpus once (+ its Bloom filter) and is the longest kernel of
        # the step; the bitmap it writes (1 bit per byte) is scratch like the other engines' event records.  ms_scan = the walks.
        dom, dom_ms = "kf_filter", avg["ms_count"]
        alg_bytes = n_bytes + 4 * info["filter_words"]
    elif engine == 2:
        # k2_traverse: corpus + doc offsets + automaton image in; its event records are scratch
        dom, dom_ms = "k2_traverse", avg["ms_count"]
        alg_bytes = n_bytes + 8 * (D + 1) + A
    elif avg["ms_write"] >= avg["ms_count"]:  # two-pass engine: ordered-write pass
        dom, dom_ms = "k_write", avg["ms_write"]
        alg_bytes = n_bytes + 12 * n_hits + 16 * (D + 1) + A
    else:
        dom, dom_ms = "k_count", avg["ms_count"]
        alg_bytes = n_bytes + 8 * (D + 1) + A
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    traffic, traffic_src = None, None
    try:  # PMC HBM bytes per launch, measured in separate rocprofv3 --pmc passes of THIS build (tools/collect_profiles.sh)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
        if pmc.get("config") == cfg and pmc.get("bytes_per_gpu") == n_bytes and pmc.get("lib_sha256") == lib_fingerprint():
            traffic = pmc["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
            traffic_src = {"file": "profiles/pmc_traffic_latest.json", "git_head": pmc.get("git_head"),
                           "lib_sha256": pmc.get("lib_sha256")[:16]}
        else:
            log("profiles/pmc_traffic_latest.json was not measured on this build/config: roofline.traffic = null")
    except Exception:
        pass
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "traffic_source": traffic_src,
        "alg_bytes_per_launch": int(alg_bytes), "avg_ms": round(dom_ms, 4), "engine": engine,
        "whole_path_frac": round((n_bytes + 12 * n_hits + 16 * (D + 1) + A) / (avg["ms_total"] * 1e-3) / 1e9
                                 / HBM_PEAK_GBS, 4),
        "kernels_ms": {k: round(v, 4) for k, v in avg.items()},
        # SURVEY.md section 8 d: "also report against the 6.29 TB/s measured-copy ceiling" (MI355X_MICROARCH.md: float4 copy)
        "frac_of_measured_copy": round(achieved / 6290.0, 4),
    }

    # ---- end to end, outside the timed region (SURVEY.md section 8 d: "incl. PCIe upload and incl. D2H of hits as
    # separate lines"): what the drop-in caller of the host-buffer entry point sees, next to the PCIe rate it is bound by
    end_to_end = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        from aha_amd import DeviceBuffer

        def best_of(fn, k=3):
            ts = []
            for _ in range(k):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return min(ts)

        probe = DeviceBuffer(local_rank, n_bytes)
        probe.upload(corpus)  # warm: first touch of the pageable pages, stream creation
        t_h2d = best_of(lambda: probe.upload(corpus))
        del probe
        # the C entry point itself on buffers the caller already owns (allocating and first-touching a 442 MB numpy
        # array per call would be timed otherwise)
        import ctypes as C

        host_out = np.ones(n_hits + 1024, dtype=np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")]))
        host_dho = np.ones(D + 1, dtype=np.uint64)
        doc_u64 = np.ascontiguousarray(doc, dtype=np.uint64)
        prm = N.aha_match_params()
        prm.struct_size = C.sizeof(N.aha_match_params)
        prm.char_offsets = 1 if args.chars else 0
        got = C.c_uint64(0)

        def host_entry():
            rc = N.lib().aha_ac_match_batch(ac._h, corpus.ctypes.data, doc_u64.ctypes.data, D, C.byref(prm),
                                            host_out.ctypes.data, host_out.size, host_dho.ctypes.data, C.byref(got))
            assert rc == 0 and got.value == n_hits, (rc, got.value)

        host_entry()  # warm: staging buffers and streams of the handle
        t_host = best_of(host_entry)

        def dev_plus_download():
            n = ac.match_batch_device(d_corpus, d_doc, d_out, d_dho, chars=args.chars)
            N.lib().aha_buffer_download(local_rank, host_out.ctypes.data, d_out.data_ptr(), 12 * n)

        dev_plus_download()
        t_dl = best_of(dev_plus_download)
        # the group API's host path on this one device: three shards of the batch, each through the same pipelined host
        # entry (aha_ac_match_batch_keep), then the exchange by device-to-device copies (device ids repeat)
        group_gbs = group_ms = group_res_ms = group_res_parts = None
        try:
            from aha_amd import ACGroup as Group
            grp = Group.compile_packed(blob, offs, [local_rank] * 3)
            gout = np.ones(n_hits + 1024, dtype=np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")]))

            def group_entry():
                rc = N.lib().aha_group_match_batch(grp._h, corpus.ctypes.data, doc_u64.ctypes.data, D, C.byref(prm),
                                                   gout.ctypes.data, gout.size, host_dho.ctypes.data, C.byref(got))
                assert rc == 0 and got.value == n_hits, (rc, got.value)

            group_entry()
            t_grp = best_of(group_entry)
            assert gout[:n_hits].tobytes() == host_out[:n_hits].tobytes(), "group hits differ from the single handle's"
            group_gbs, group_ms = round(n_bytes / t_grp / 1e9, 2), round(t_grp * 1e3, 2)
            # ... and the resident entry: the three ranges stay on the device, every shard matches its own, the same exchange;
            # nothing of the batch crosses PCIe (what a one-process caller reaches the metric's "corpus resident" with at N > 1)
            res = grp.upload_corpus(corpus, doc_u64)

            def group_resident():
                n, _ = grp.match_corpus(res)
                assert n == n_hits, n

            group_resident()
            t_res = best_of(group_resident)
            gt = grp.last_timing()
            assert grp.download_shard(1)[:n_hits].tobytes() == host_out[:n_hits].tobytes(), "resident group hits differ"
            group_res_ms = round(t_res * 1e3, 2)
            group_res_parts = {"match_ms": round(gt["ms_match"], 2), "exchange_ms": round(gt["ms_exchange"], 2)}
            del res
            del grp
        except Exception as e:  # the leg is informational
            log(f"group host entry leg failed: {e!r}")
        end_to_end = {"host_entry_gbs": round(n_bytes / t_host / 1e9, 2), "host_entry_ms": round(t_host * 1e3, 2),
                      "with_download_gbs": round(n_bytes / t_dl / 1e9, 2), "with_download_ms": round(t_dl * 1e3, 2),
                      "pcie_h2d_gbs": round(n_bytes / t_h2d / 1e9, 2),
                      "host_entry_vs_pcie": round(t_h2d / t_host, 3),
                      "group_host_entry_gbs": group_gbs, "group_host_entry_ms": group_ms,
                      "group_resident_ms": group_res_ms, "group_resident_parts": group_res_parts,
                      "note": "host_entry = aha_ac_match_batch on pageable host buffers (upload, match and download "
                              "pipelined over document ranges); with_download = device-resident match + D2H of the "
                              "hits; pcie_h2d = one blocking upload of the same corpus; group_host_entry = aha_group_match_batch over "
                              "three shards on this one device (shards that share a device run one after the other: upload, "
                              "match and the copy of each shard's hits to its place in the caller's buffer pipelined per shard; "
                              "the gathered list is also built on the device); group_resident = aha_group_corpus_upload once, then "
                              "aha_group_match_batch_device over the same three shards (host wall clock of the call: three matches side "
                              "by side on one device + the exchange); best of 3; never `value`"}
        log(f"end to end: {end_to_end}")

    # ---- parity gate: no throughput figure without a bit-exact comparison on this run's hits.
    # Rank 0 checks its own shard once, outside the timed region, on every run (N > 1 and profiler runs included).
    cpu = None
    parity = "unchecked"
    if rank == 0:
        hits_h = d_out[:n_hits].cpu().numpy().view(np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")])).reshape(-1)
        dho_h = d_dho.cpu().numpy().astype(np.uint64)
        parity = parity_sample(blob, offs, corpus, doc, hits_h, dho_h, args.chars, args.parity_bytes, log)
        if world == 1 and not args.no_cpu_baseline and parity == "bit-exact":
            cpu, exact = cpu_baseline(blob, offs, corpus, doc, None if args.chars else hits_h, dho_h, args.cpu_seconds, log)
            if args.chars:
                cpu["parity_on_sample"] = "unchecked (the timed CPU leg is the Bytes overload)"
            if not exact:
                parity = "MISMATCH"
        if parity != "bit-exact":
            log("PARITY MISMATCH against the oracle: no throughput is reported")

    if rank == 0:
        ok = parity == "bit-exact"
        line = {
            "metric": "input GB/s scanned + M-hits/s, 100k-pattern AC over 1 GiB UTF-8 corpus",
            "value": round(gbs, 3) if ok else None, "unit": "GB/s", "m_hits_per_s": round(mhits, 2) if ok else None,
            "parity": parity,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"cfg{cfg if world == 1 else 4}: {K} patterns, {n_bytes} B synthetic "
                                   f"{'UTF-8' if cfg != 2 else 'ASCII'} corpus per GPU, {D} docs",
                       "keys": K, "bytes_per_gpu": n_bytes, "docs_per_gpu": D, "hits_per_gpu": n_hits,
                       "slots": info["n_slots"], "slot_bytes": info["slot_bytes"], "max_key_len": info["max_key_len"],
                       "offsets": "chars" if args.chars else "bytes",
                       "compile_s": round(compile_s, 3), "unit_slots": info.get("unit_slots"),
                       "parallelism": f"doc-sharded x{world}" + (f" + {args.gather} ({args.exchange})" if world > 1 else "")
                                      + (" (overlapped)" if overlap else "")},
            "roofline": roofline,
        }
        if no_events is not None and ok:
            line["without_timing_events"] = no_events
        if breakdown is not None:
            line["breakdown"] = breakdown
        if strong is not None:
            line["strong_scaling"] = strong
            if strong["identical_to_one_gpu"] is False:
                parity = "MISMATCH"
                line["parity"], line["value"], line["m_hits_per_s"] = parity, None, None
        if end_to_end is not None:
            line["end_to_end"] = end_to_end
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity != "bit-exact":
        sys.exit(1)


if __name__ == "__main__":
    main()
