"""Driver + numpy model for the position-parallel shallow-classify prototype
(tools/proto_pp.hip).  Prototype only: times the idea for DESIGN.md's roadmap
and checks its counts; not part of the product path."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ["AHA_FILTER"] = "1"
from aha_amd import AC, synth  # noqa: E402

so = os.path.join(ROOT, "tools", "libproto_pp.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "proto_pp.hip")])
L = C.CDLL(so)
L.proto_pp_run.restype = C.c_int
L.proto_pp_run.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int,
                           C.c_void_p]

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
ac = AC.compile_packed(blob, offs, host_only=True)
info = ac.info
assert info["filter_d0"] == 2, info
slots = ac.export(0, np.uint32)
T = info["lds_slots"]
rows = slots[:T].copy()
# rebuild the Bloom filter at 64 KiB from the exact set (same hash as automaton.hpp) so that rows + filter +
# the per-wave staging of 16 waves fit the 160 KiB of LDS
xs = ac.export(6, np.uint64)
xs = xs[xs != 0]
WORDS = 16384
Bv = (xs >> np.uint64(32)).astype(np.uint64)
wv = (xs & np.uint64(0xFFFFFFFF)).astype(np.uint64)
M32 = np.uint64(0xFFFFFFFF)
h = ((Bv * np.uint64(0x9E3779B1)) & M32) ^ ((wv * np.uint64(0x85EBCA6B)) & M32)
h ^= h >> np.uint64(15)
h = (h * np.uint64(0x2C1B3C6D)) & M32
h ^= h >> np.uint64(13)
g = (h * np.uint64(0x297A2D39)) & M32
mask = (np.uint64(1) << (g >> np.uint64(27))) | (np.uint64(1) << ((g >> np.uint64(22)) & np.uint64(31))) | \
       (np.uint64(1) << ((g >> np.uint64(17)) & np.uint64(31)))
widx = ((h * np.uint64(WORDS)) >> np.uint64(32)).astype(np.int64)
bloom = np.zeros(WORDS, dtype=np.uint32)
np.bitwise_or.at(bloom, widx, mask.astype(np.uint32))
print("t_rows", T, "bloom words", bloom.size, "entries", xs.size)

dev = torch.device("cuda:0")
d_rows = torch.from_numpy(rows.view(np.int32)).to(dev)
d_bloom = torch.from_numpy(bloom.view(np.int32)).to(dev)
d_text = torch.from_numpy(corpus).to(dev)
d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)
grid = torch.cuda.get_device_properties(0).multi_processor_count


def run():
    d_cnt.zero_()
    rc = L.proto_pp_run(d_rows.data_ptr(), T, d_bloom.data_ptr(), bloom.size, d_text.data_ptr(), corpus.size,
                        d_cnt.data_ptr(), grid, None)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(5):
    run()
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 5
cnt = d_cnt.cpu().numpy()
print(f"k_pp: {ms:.3f} ms per pass over {corpus.size} bytes = {corpus.size / ms / 1e6:.1f} GB/s; "
      f"shallow END events {cnt[0]}, boundary positions {cnt[1]} ({cnt[1] / corpus.size:.3f}/byte), "
      f"suspects {cnt[2]} ({cnt[2] / corpus.size:.4f}/byte)")

# ---- numpy model on a prefix (single document semantics, like the kernel) ----
m = min(corpus.size, 1 << 22)
t = corpus[:m].astype(np.uint32)
prev = np.concatenate([[0], t[:-1]])
e1 = rows[t]
ok1 = (t != 0) & ((e1 & 0xFF) == t)
e1p = rows[prev]
ok1p = (prev != 0) & ((e1p & 0xFF) == prev)
idx2 = ((e1p >> 8) & 0x3FFFFF) ^ t
e2 = np.where(ok1p & (idx2 < T), rows[np.minimum(idx2, T - 1)], 0)
ok2 = ok1p & (t != 0) & ((e2 & 0xFF) == t)
ex = np.where(ok2, e2, np.where(ok1, e1, 0))
print("model (first %d bytes): END events %d, boundary positions %d" % (m, int(np.count_nonzero(ex & 0x80000000)),
                                                                       int(np.count_nonzero(ok2))))
d_cnt.zero_()
L.proto_pp_run(d_rows.data_ptr(), T, d_bloom.data_ptr(), bloom.size, d_text.data_ptr(), m, d_cnt.data_ptr(), grid, None)
torch.cuda.synchronize()
c2 = d_cnt.cpu().numpy()
print("kernel on the same prefix: END events %d, boundary positions %d" % (c2[0], c2[1]))
assert c2[0] == np.count_nonzero(ex & 0x80000000) and c2[1] == np.count_nonzero(ok2)
print("counts agree")
