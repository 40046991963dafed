"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on k2_traverse's own access pattern (MI355X_MICROARCH.md, HBM:
"calibrate on a known byte count in your own access pattern"): an automaton with one key that never occurs walks
1 GiB of text entirely in LDS, so the kernel's only HBM reads are the per-lane 2 x 16-byte staging loads of the
corpus (+ 8 bytes per document offset) and it writes nothing but per-chunk counters.
Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/calib_fetch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aha_amd import AC

n = 1 << 30
rng = np.random.default_rng(3)
corpus = rng.integers(97, 123, size=n, dtype=np.uint8)          # a..z only
doc = np.arange(0, n + 1, 1 << 20, dtype=np.int64)
ac = AC.compile([b"\xff\xfe"])                                   # never occurs
dc = torch.from_numpy(corpus).cuda(); dd = torch.from_numpy(doc).cuda()
out = torch.zeros((1024, 3), dtype=torch.int32, device="cuda")
dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
for _ in range(3):
    assert ac.match_batch_device(dc, dd, out, dho) == 0
torch.cuda.synchronize()
print("corpus bytes", n)
