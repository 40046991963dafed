import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from aha_amd import AC, synth
n_bytes = 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
ac = AC.compile_packed(blob, offs); ac.set_profiling(True)
dc = torch.from_numpy(corpus).cuda(); dd = torch.from_numpy(doc.astype(np.int64)).cuda()
cap = n_bytes // 16
out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
words = torch.zeros(2 * cap + cap // 1024 + 2, dtype=torch.int32, device="cuda"); nw = torch.zeros(1, dtype=torch.int64, device="cuda")
for fused in (False, True, False, True):
    ts = []
    for _ in range(6):
        if fused: ac.match_batch_device(dc, dd, out, None, words=words, n_words=nw)
        else: ac.match_batch_device(dc, dd, out, None)
        t = ac.last_timing(); ts.append((t["ms_total"], t["ms_count"], t["ms_write"]))
    ts.sort(); print("fused" if fused else "plain", ts[len(ts)//2], flush=True)
