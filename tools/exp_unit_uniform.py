"""Experiment: the traversal engines on a text of three-byte characters only (every unit has the same length, so the
lanes of a wave stay in step) against the cfg 3 mix.  AHA_ENGINE=unit / v2 from the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, synth

n_bytes = 1 << 30
blob, offs, nf = synth.keys(3)
ac = AC.compile_packed(blob, offs)
ac.set_profiling(True)
rng = np.random.default_rng(1)
cp = rng.integers(0x4E00, 0x9FA6, size=n_bytes // 3, dtype=np.uint32)
t = np.empty((cp.size, 3), dtype=np.uint8)
t[:, 0] = 0xE0 | (cp >> 12)
t[:, 1] = 0x80 | ((cp >> 6) & 0x3F)
t[:, 2] = 0x80 | (cp & 0x3F)
uni = t.reshape(-1)
mix, doc_mix = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
for name, corpus in (("three-byte characters only", uni), ("cfg 3 mix", mix)):
    n = corpus.size
    doc = np.linspace(0, n, 1025).astype(np.int64) // 3 * 3
    doc[-1] = n
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc).cuda()
    out = torch.zeros((n // 16, 3), dtype=torch.int32, device="cuda")
    for _ in range(3):
        h = ac.match_batch_device(dc, dd, out, None)
    tm = ac.last_timing()
    print(f"{os.environ.get('AHA_ENGINE')}: {name}: engine {tm['engine']} traverse {tm['ms_count']:.3f} ms total {tm['ms_total']:.3f} ms (aux {tm['ms_aux']:.3f}, write {tm['ms_write']:.3f}) hits {h}", flush=True)
    del dc, out
