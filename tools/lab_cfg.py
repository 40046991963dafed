"""Lab: one library build (AHA_HIP_LIB) on a BASELINE config: the library's own HIP-event times per call, median of 7.
python3 tools/lab_cfg.py <config> [bytes] -- prints engine, ms_count (traversal / filter), ms_scan, ms_aux (regroup etc.), ms_write
(expansion) and ms_total.  Timing-only lab builds may produce wrong hits: no comparison here (tests and bench.py do that)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, AhaError, synth
from aha_amd import _native as N

cfg = int(sys.argv[1])
n_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else synth.DEFAULT_BYTES[cfg]
blob, offs, nf = synth.keys(cfg)
ac = AC.compile_packed(blob, offs)
ac.set_profiling(True)
corpus, doc = synth.corpus(cfg, blob, offs, nf, n_bytes=n_bytes)
dc = torch.from_numpy(corpus).cuda()
dd = torch.from_numpy(doc.astype(np.int64)).cuda()
dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
try:
    n = ac.match_batch_device(dc, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"), dho)
except AhaError as e:
    if e.code != N.AHA_E_CAPACITY:
        raise
    n = e.required
out = torch.zeros((n + 1024, 3), dtype=torch.int32, device="cuda")
ts = []
for _ in range(7):
    h = ac.match_batch_device(dc, dd, out, dho)
    ts.append(ac.last_timing())
med = lambda k: sorted(t[k] for t in ts)[len(ts) // 2]
print(f"{os.path.basename(os.environ.get('AHA_HIP_LIB', 'libaha_hip.so'))} {os.environ.get('AHA_LAB_NOTE', '')}: cfg {cfg} engine {ts[-1]['engine']} "
      f"count {med('ms_count'):.3f} scan {med('ms_scan'):.3f} aux {med('ms_aux'):.3f} write {med('ms_write'):.3f} total {med('ms_total'):.3f} ms hits {h}", flush=True)
