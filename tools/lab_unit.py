"""Lab: the character-level traversal of one library build on cfg 3 (or cfg 3's keys on another text): traversal and total
time from the HIP events inside the library.  AHA_HIP_LIB selects the build (product or a timing-only lab variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
blob, offs, nf = synth.keys(3)
ac = AC.compile_packed(blob, offs)
ac.set_profiling(True)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
dc = torch.from_numpy(corpus).cuda()
dd = torch.from_numpy(doc.astype(np.int64)).cuda()
out = torch.zeros((n_bytes // 16, 3), dtype=torch.int32, device="cuda")
ts = []
for _ in range(7):
    h = ac.match_batch_device(dc, dd, out, None)
    tm = ac.last_timing()
    ts.append((tm["ms_count"] + tm["ms_scan"], tm["ms_total"], tm["ms_write"], tm["ms_count"], tm["ms_scan"]))
ts.sort()
m = ts[len(ts) // 2]
print(f"{os.path.basename(os.environ.get('AHA_HIP_LIB', 'libaha_hip.so'))} {os.environ.get('AHA_LAB_NOTE', '')}: engine {tm['engine']} "
      f"traverse {m[0]:.3f} ms (kernels {m[3]:.3f} + {m[4]:.3f}) total {m[1]:.3f} ms expansion {sorted(t[2] for t in ts)[len(ts) // 2]:.3f} ms hits {h}", flush=True)
