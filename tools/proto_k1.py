"""Lab: pair-filter prototype (tools/lab/proto_k1.hip) on cfg 3: time per GiB, candidate density, no false negatives."""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
so = os.path.join(ROOT, "tools/lab/libproto_k1.so")
L = C.CDLL(so)

blob, offs, nf = synth.keys(3)
keys = [bytes(blob[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]

def ulen(b):
    return 1 if b < 0x80 else 2 if 0xC0 <= b < 0xE0 else 3 if 0xE0 <= b < 0xF0 else 0

pairs = set()
for k in keys:
    l1 = ulen(k[0]); l2 = ulen(k[l1])
    pairs.add(k[:l1 + l2])
print("pairs", len(pairs), flush=True)

def clen(b):
    return 2 if 0xC0 <= b < 0xE0 else 3 if 0xE0 <= b < 0xF0 else 1

def hash_pairs(ps):
    c1 = np.zeros(len(ps), dtype=np.uint64); c2 = np.zeros(len(ps), dtype=np.uint64)
    for i, p in enumerate(ps):
        l1 = clen(p[0])
        c1[i] = int.from_bytes(p[:l1], "little"); c2[i] = int.from_bytes(p[l1:], "little")
    M = np.uint64(0xFFFFFFFF)
    g = (c1 * np.uint64(0x85EBCB)) & M
    gp = ((g >> np.uint64(11)) | (g << np.uint64(21))) & M
    h = (c2 * np.uint64(0x9E3779) + gp) & M
    h ^= h >> np.uint64(16)
    return h

LOG2W = int(sys.argv[2]) if len(sys.argv) > 2 else 13
bw = 1 << LOG2W
h = hash_pairs(sorted(pairs))
wi = h >> np.uint64(32 - LOG2W)
m = (np.uint64(1) << (h & np.uint64(31))) | (np.uint64(1) << ((h >> np.uint64(5)) & np.uint64(31)))
bloom = np.zeros(bw, dtype=np.uint32)
np.bitwise_or.at(bloom, wi.astype(np.int64), m.astype(np.uint32))
print("bloom fill", np.unpackbits(bloom.view(np.uint8)).mean(), flush=True)
lentab = np.full(256, 8, dtype=np.uint8)
lentab[0xC0:0xE0] = 16; lentab[0xE0:0xF0] = 24

corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
dc = torch.from_numpy(corpus).cuda()
db = torch.from_numpy(bloom.view(np.int32)).cuda()
dl = torch.from_numpy(lentab).cuda()
out = torch.zeros(n_bytes // 64 + 64, dtype=torch.int64, device="cuda")
tab = torch.randint(0, 1 << 30, (1 << 19,), dtype=torch.int32, device="cuda")  # 2^17 entries of 16 bytes
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
ms = C.c_float(0)
for variant in (0, 1, 2, 4, 6, 8, 0x200, 0x201, 0x202, 0x206):
    out.zero_()
    rc = L.proto_k1_run(C.c_void_p(dc.data_ptr()), C.c_uint64(n_bytes), C.c_void_p(db.data_ptr()), C.c_uint32(LOG2W),
                        C.c_void_p(dl.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(tab.data_ptr()), C.c_uint32((1 << 17) - 1),
                        C.c_void_p(sink.data_ptr()), variant, 5, C.byref(ms))
    torch.cuda.synchronize()
    bits = out[: n_bytes // 64].cpu().numpy().view(np.uint64)
    cnt = int(sum(bin(int(x)).count("1") for x in bits[: 1 << 14]))
    print(f"variant {variant} rc {rc}: {ms.value:.3f} ms for {n_bytes} B = {n_bytes / ms.value / 1e6:.1f} GB/s; "
          f"candidates per byte (first MiB) {cnt / (64 << 14):.4f}", flush=True)
    # exactness of the filter on the first 2 MiB: every true pair start is flagged
    nb = min(n_bytes, 1 << 21)
    t = bytes(corpus[:nb + 8])
    miss = 0; true = 0; flagged = 0
    i = 0
    while i < nb - 8:
        l1 = ulen(t[i])
        if l1 == 0:
            i += 1
            continue
        l2 = ulen(t[i + l1])
        f = (int(bits[i >> 6]) >> (i & 63)) & 1
        flagged += f
        if l2 and t[i:i + l1 + l2] in pairs and variant in (0, 2, 0x200, 0x202):
            true += 1
            if not f:
                miss += 1
        i += l1
    print(f"  true pair starts {true}, flagged {flagged}, false negatives {miss}", flush=True)
