"""Throughput with many small documents (the reference's typical input is short strings)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aha_amd import AC, AhaError, synth
blob, offs, nf = synth.keys(3)
ac = AC.compile_packed(blob, offs)
ac.set_profiling(True)
for docb in (1 << 20, 4096, 256, 64, 16):
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 28, doc_bytes=docb)
    dc = torch.from_numpy(corpus).cuda(); dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
    try:
        n = ac.match_batch_device(dc, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"), dho)
    except AhaError as e:
        n = e.required
    out = torch.zeros((n + 16, 3), dtype=torch.int32, device="cuda")
    for _ in range(2): ac.match_batch_device(dc, dd, out, dho)
    t = ac.last_timing()
    print(f"doc~{docb:8d} B: {doc.size - 1:9d} docs, {n:9d} hits, total {t['ms_total']:.3f} ms "
          f"({corpus.size / t['ms_total'] / 1e6:.1f} GB/s), traverse {t['ms_count']:.3f} ms, post {t['ms_aux'] + t['ms_write']:.3f} ms")
