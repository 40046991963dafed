"""Traversal on stream A beside (a) a one-block spin kernel, (b) a streaming copy on stream B."""
import os, sys, threading, time
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
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 27, dtype=torch.int32, device=dev)
y = torch.zeros_like(x)
K = 10


def match_loop():
    for _ in range(K):
        ac.match_batch_device(dc, dd, out, dho, stream=sa.cuda_stream)


def spin_loop():
    with torch.cuda.stream(sb):
        torch.cuda._sleep(int(2.4e9 * 0.030))  # ~30 ms
    sb.synchronize()


def copy_loop():
    with torch.cuda.stream(sb):
        for _ in range(K * 4):
            y.copy_(x)  # 1 GiB of traffic each
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
    return (time.perf_counter() - t0) * 1e3


for fns, name in (([match_loop], "10 matches"), ([spin_loop], "spin"), ([copy_loop], "40 copies of 512 MiB"),
                  ([match_loop, spin_loop], "matches + spin"), ([match_loop, copy_loop], "matches + copies")):
    run(fns)
    print(f"{name}: {run(fns):.2f} ms", flush=True)
