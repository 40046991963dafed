"""Cost of the exchange formats on one MI355X (test tooling): pack / rebuild kernels over the hits of the cfg 3 batch.
python tools/bench_exchange.py [bytes]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from aha_amd import AC, synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
ac = AC.compile_packed(blob, offs)
dev = torch.device("cuda:0")
dc = torch.from_numpy(corpus).to(dev)
dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
out = torch.zeros((n_bytes // 16, 3), dtype=torch.int32, device=dev)
n = ac.match_batch_device(dc, dd, out, None)
words = torch.zeros(2 * n + n // 1024 + 64, dtype=torch.int32, device=dev)
pairs = torch.zeros((n, 2), dtype=torch.int32, device=dev)
back = torch.zeros((n, 3), dtype=torch.int32, device=dev)
nw = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t1 = timed(lambda: ac.hits_pack4_device(out, n, words, nw))
t2 = timed(lambda: ac.hits_unpack4_device(words, n, back))
assert torch.equal(back, out[:n])
t3 = timed(lambda: ac.hits_pack_device(out, n, pairs))
t4 = timed(lambda: ac.hits_unpack_device(pairs, n, back))
assert torch.equal(back, out[:n])
print(f"{n} hits ({n * 12 / 1e6:.0f} MB of triples): words stream {int(nw[0]) * 4 / 1e6:.1f} MB "
      f"({int(nw[0]) * 4 / n:.3f} B per hit), pack {t1:.3f} ms, rebuild {t2:.3f} ms "
      f"({n * 16 / t2 / 1e6:.0f} GB/s of read+write); pairs {n * 8 / 1e6:.0f} MB, pack {t3:.3f} ms, rebuild {t4:.3f} ms")
