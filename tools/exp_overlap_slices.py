"""Experiment: does the post pipeline of one sub-batch hide under the traversal of another?  One handle, T host
threads, one stream each (per-call scratch sets), each thread matching its own slice of the cfg 3 corpus."""
import sys
import threading
import time

import numpy as np
import torch

from aha_amd import AC, synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
ac = AC.compile_packed(blob, offs)
dev = torch.device("cuda:0")
D = doc.size - 1
for T in (1, 2, 4):
    parts = []
    for r in range(T):
        lo, hi = D * r // T, D * (r + 1) // T
        c = torch.from_numpy(corpus[int(doc[lo]):int(doc[hi])]).to(dev)
        d = torch.from_numpy((doc[lo:hi + 1] - doc[lo]).astype(np.int64)).to(dev)
        out = torch.zeros((c.numel() // 16, 3), dtype=torch.int32, device=dev)
        dho = torch.zeros(d.numel(), dtype=torch.int64, device=dev)
        parts.append((c, d, out, dho, torch.cuda.Stream()))
    steps = 10

    def work(r, k):
        c, d, out, dho, st = parts[r]
        for _ in range(k):
            ac.match_batch_device(c, d, out, dho, stream=st.cuda_stream)

    def run(k):
        ths = [threading.Thread(target=work, args=(r, k)) for r in range(T)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(3)
    el = run(steps)
    print(f"{T} concurrent slices: {el / steps * 1e3:.3f} ms per {n_bytes >> 20} MiB = {n_bytes * steps / el / 1e9:.1f} GB/s", flush=True)
