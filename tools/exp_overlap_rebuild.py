"""Experiment: what does an 8-GPU step cost ONE GPU in kernels -- the match of the next step beside the pack of the own
hits and the rebuild of the seven peers' exchange streams (one launch over a segment table since round 3; pass
--launches 7 for the round-2 form, one unpack launch per peer)?  Two host threads, two streams, one GPU.
Round 5: the match call leaves the exchange stream as well (aha_ac_match_batch_device_stream: the pack kernels behind the match
on ITS stream); --separate-pack runs round 4's form (the pack on the second stream, beside the next match).
(profiles/r05_fused_exchange_stream.txt: the expansion writing the words itself was built, measured slower and removed.)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from aha_amd import AC, synth

n_bytes = 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
ac = AC.compile_packed(blob, offs)
dev = torch.device("cuda:0")
dc = torch.from_numpy(corpus).to(dev)
dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
out = torch.zeros((n_bytes // 16, 3), dtype=torch.int32, device=dev)
dho = torch.zeros(doc.size, dtype=torch.int64, device=dev)
n = ac.match_batch_device(dc, dd, out, dho)
words = torch.zeros(2 * n + n // 1024 + 64, dtype=torch.int32, device=dev)
nw = torch.zeros(1, dtype=torch.int64, device=dev)
ac.hits_pack4_device(out, n, words, nw)
allh = torch.zeros((8 * n, 3), dtype=torch.int32, device=dev)
one_launch = "--launches" not in sys.argv
fused = "--separate-pack" not in sys.argv
capo = out.shape[0]
words2 = torch.zeros(2 * capo + capo // 1024 + 2, dtype=torch.int32, device=dev)
nw2 = torch.zeros_like(nw)
nwords = None
out2 = torch.zeros_like(out)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
K = 10


def match_loop():
    for _ in range(K):
        if fused:
            ac.match_batch_device(dc, dd, out2, dho, stream=sa.cuda_stream, words=words2, n_words=nw2)
        else:
            ac.match_batch_device(dc, dd, out2, dho, stream=sa.cuda_stream)


def rebuild_loop():
    for _ in range(K):
        if not fused:
            ac.hits_pack4_device(out, n, words, nw, stream=sb.cuda_stream)
        if one_launch:  # the seven peers' streams (here: seven times the own one) rebuilt by one launch
            ac.hits_unpack4_segs_device(words, [(0, n, p * n) for p in range(1, 8)], allh, stream=sb.cuda_stream)
        else:
            for p in range(1, 8):
                ac.hits_unpack4_device(words, n, allh[p * n:(p + 1) * n], stream=sb.cuda_stream)
    sb.synchronize()


def run(fns):
    ths = [threading.Thread(target=f) for f in fns]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


print("match + pack in one call, one stream" if fused else "separate pack kernels (round 4)", flush=True)
for fns, name in (([match_loop], "match (+ stream) alone" if fused else "match alone"),
                  ([rebuild_loop], "7 rebuilds alone" if fused else "pack + 7 rebuilds alone"),
                  ([match_loop, rebuild_loop], "both, two streams")):
    run(fns)
    print(f"{name}: {run(fns):.3f} ms per step", flush=True)
