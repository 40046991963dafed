"""Lab: the start-parallel formulation on the GPU -- pair filter (tools/lab/proto_k1.hip) + candidate walks (tools/lab/proto_k2.hip)
over the hash image of tools/lab/hash_engine (built by the lab library of commit fc6578a) -- on cfg 3's text as ONE document:
time of both kernels, events and hits against the product engine."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
blob, offs, nf = synth.keys(3)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes, doc_bytes=n_bytes)
assert doc.size == 2, doc
# ---- the hash image, from the lab library (host only)
os.environ["AHA_ENGINE"] = "hash"
H = C.CDLL(os.path.join(ROOT, "tools/lab/hash_engine/libaha_hip_hash.so"))
class Opt(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]
o = Opt(16, -1, 1, 0)
h = C.c_void_p(); ek = C.c_uint32(0)
H.aha_ac_compile.restype = C.c_int32
H.aha_ac_compile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
H.aha_ac_export.restype = C.c_int64
H.aha_ac_export.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
assert H.aha_ac_compile(blob.ctypes.data, offs.ctypes.data, offs.size - 1, C.byref(o), C.byref(h), C.byref(ek)) == 0
def export(which, dtype):
    n = H.aha_ac_export(h, which, None, 0)
    buf = np.zeros(int(n) // np.dtype(dtype).itemsize, dtype=dtype)
    H.aha_ac_export(h, which, buf.ctypes.data, int(n))
    return buf
par = export(14, np.uint32)
k1, n_groups, pair_log2, deep_log2 = (int(x) for x in par[:4])
bloom, disp, pairs, deep = export(10, np.uint32), export(11, np.uint8), export(12, np.uint32), export(13, np.uint32)
print("hash image:", par, flush=True)
del os.environ["AHA_ENGINE"]
# ---- reference: the product engine on the same text
ac = AC.compile_packed(blob, offs)
ac.set_profiling(True)
dc = torch.from_numpy(corpus).cuda()
dd = torch.from_numpy(doc.astype(np.int64)).cuda()
out = torch.zeros((n_bytes // 16, 3), dtype=torch.int32, device="cuda")
n = ac.match_batch_device(dc, dd, out, None)
for _ in range(3):
    ac.match_batch_device(dc, dd, out, None)
t = ac.last_timing()
ends = out[:n, 1]
n_events = int(torch.unique_consecutive(ends).numel())
print(f"product: {n} hits, {n_events} events; traverse {t['ms_count']:.3f} ms, total {t['ms_total']:.3f} ms", flush=True)
max_len = ac.info["max_key_len"]
del out
# ---- K1: the pair filter
L1 = C.CDLL(os.path.join(ROOT, "tools/lab/libproto_k1.so"))
L2 = C.CDLL(os.path.join(ROOT, "tools/lab/libproto_k2.so"))
L3 = C.CDLL(os.path.join(ROOT, "tools/lab/libproto_k3.so"))
lentab = np.full(256, 8, dtype=np.uint8); lentab[0xC0:0xE0] = 16; lentab[0xE0:0xF0] = 24
db = torch.from_numpy(bloom.view(np.int32)).cuda(); dl = torch.from_numpy(lentab).cuda()
bitmap = torch.zeros(n_bytes // 64 + 64, dtype=torch.int64, device="cuda")
tab = torch.zeros(16, dtype=torch.int32, device="cuda"); sink = torch.zeros(4, dtype=torch.int32, device="cuda")
ms1 = C.c_float(0)
assert k1 == 0x9E3779, hex(k1)  # (proto_k1's multiplier)
rc = L1.proto_k1_run(C.c_void_p(dc.data_ptr()), C.c_uint64(n_bytes), C.c_void_p(db.data_ptr()), C.c_uint32(14), C.c_void_p(dl.data_ptr()),
                     C.c_void_p(bitmap.data_ptr()), C.c_void_p(tab.data_ptr()), C.c_uint32(0), C.c_void_p(sink.data_ptr()), 0, 5, C.byref(ms1))
torch.cuda.synchronize()
print(f"K1 filter rc {rc}: {ms1.value:.3f} ms", flush=True)
# ---- K2: the walks
dd_ = torch.from_numpy(disp).cuda(); dp = torch.from_numpy(pairs.view(np.int32)).cuda(); de = torch.from_numpy(deep.view(np.int32)).cuda()
cnt = torch.zeros(8, dtype=torch.int64, device="cuda")
ms2 = C.c_float(0)
for grid, steps in ((1024, 40),):
    rc = L2.proto_k2_run(C.c_void_p(dc.data_ptr()), C.c_uint64(n_bytes), C.c_void_p(bitmap.data_ptr()), C.c_void_p(dd_.data_ptr()),
                         C.c_void_p(dp.data_ptr()), C.c_void_p(de.data_ptr()), C.c_uint32(n_groups), C.c_uint32(pair_log2),
                         C.c_uint32(deep_log2), C.c_uint32(k1), C.c_uint32(max_len), C.c_void_p(cnt.data_ptr()), grid, 5, C.byref(ms2), steps)
    torch.cuda.synchronize()
    c = cnt.cpu().numpy()
    print(f"K2 walks rc {rc} grid {grid} deep steps <= {steps}: {ms2.value:.3f} ms; events {c[0]} (product {n_events}), hits {c[1]} (product {n}), "
          f"starts with more than 4 END steps {c[3]}, candidates {c[4]}, pair hits {c[5]}", flush=True)
ms3 = C.c_float(0)
for grid in (512, 1024):
    rc = L3.proto_k3_run(C.c_void_p(dc.data_ptr()), C.c_uint64(n_bytes), C.c_void_p(bitmap.data_ptr()), C.c_void_p(dd_.data_ptr()),
                         C.c_void_p(dp.data_ptr()), C.c_void_p(de.data_ptr()), C.c_uint32(n_groups), C.c_uint32(pair_log2),
                         C.c_uint32(deep_log2), C.c_uint32(k1), C.c_uint32(max_len), C.c_void_p(cnt.data_ptr()), grid, 5, C.byref(ms3))
    torch.cuda.synchronize()
    c = cnt.cpu().numpy()
    print(f"K3 walks (lane per piece, two pair probes + a walker step in flight) rc {rc} grid {grid}: {ms3.value:.3f} ms; events {c[0]} "
          f"(product {n_events}), hits {c[1]} (product {n}), overflows {c[2]}, candidates {c[3]}, walkers {c[4]}, wave-iterations {c[5]}", flush=True)
print(f"K1 + K3: {ms1.value + ms3.value:.3f} ms against the product traversal's {t['ms_count']:.3f} ms", flush=True)
