"""CPU model: far (non-LDS) lookups per byte of the k2 trip on cfg 3 when part of the LDS prefix is traded for a
Bloom filter over the goto transitions that leave depth-3 states (a probe the filter rejects is not issued)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aha_amd import AC, synth

kb, ko, nf = synth.keys(3)
ac = AC.compile_packed(kb, ko, host_only=True)
info = ac.info
slots = ac.export(0, np.uint32)
n = slots.size
label = slots & 0xFF
base = (slots >> 8) & 0x3FFFFF
depth = np.full(n, -1, dtype=np.int32)
depth[0] = 0
level = np.array([0], dtype=np.int64)
labs = np.arange(1, 256, dtype=np.int64)
d = 0
trans = {}  # depth -> (B array, b array)
while level.size:
    nxt, tb, tl = [], [], []
    for i in range(0, level.size, 1 << 16):
        Bs = level[i:i + (1 << 16)]
        idx = Bs[:, None] ^ labs[None, :]
        ok = label[idx] == labs[None, :]
        nxt.append(base[idx[ok]].astype(np.int64))
        bb, ll = np.nonzero(ok)
        tb.append(Bs[bb]); tl.append(labs[ll])
    trans[d] = (np.concatenate(tb), np.concatenate(tl))
    level = np.concatenate(nxt) if nxt else np.array([], dtype=np.int64)
    d += 1
    depth[level] = d
    if d > 6: break
dep = depth.tolist()
sl = slots.tolist()
NB = 1 << 20
corpus, doc = synth.corpus(3, kb, ko, nf, n_bytes=NB)
text = corpus.tobytes()
FR = 0x40000000

def h32(B, b):
    x = (B * 0x9E3779B1 + b * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x2C1B3C6D) & 0xFFFFFFFF; x ^= x >> 13
    return x

def run(filter_kb, k, depths):
    T = ((124 * 1024 - filter_kb * 1024) // 4) & ~3
    mbits = filter_kb * 1024 * 8
    bits = None
    if filter_kb:
        bits = np.zeros(mbits, dtype=bool)
        for dd in depths:
            Bs, ls = trans[dd]
            for B_, l_ in zip(Bs.tolist(), ls.tolist()):
                x = h32(B_, l_)
                for j in range(k):
                    bits[((x >> (j * 11)) * 2654435761 & 0xFFFFFFFF) % mbits] = True
        bl = bits.tolist()
    far = 0; farhdr = 0; rejected = 0; fp = 0
    B = 0; fr = 0; hdr = False; i = 0; trips = 0
    while i < NB:
        b = text[i]; trips += 1
        if hdr:
            if B >= T: far += 1; farhdr += 1
            en = sl[B]; B = (en >> 8) & 0x3FFFFF; fr = en & FR; hdr = False
            continue
        idx = B ^ b
        skip = False
        if filter_kb and dep[B] in depths:
            x = h32(B, b)
            ok = all(bl[((x >> (j * 11)) * 2654435761 & 0xFFFFFFFF) % mbits] for j in range(k))
            if not ok: skip = True; rejected += 1
        if skip:
            m = False
        else:
            en = sl[idx]
            if idx >= T: far += 1
            m = (en & 0xFF) == b
            if filter_kb and dep[B] in depths and not m: fp += 1
        if m:
            B = (en >> 8) & 0x3FFFFF; fr = en & FR; i += 1; continue
        if B == 0 or fr:
            e0 = sl[b]
            if (e0 & 0xFF) == b: B = (e0 >> 8) & 0x3FFFFF; fr = e0 & FR
            else: B = 0; fr = 0
            i += 1
        else:
            hdr = True
    print("filter %2d KB k=%d depths=%s T=%5d: far lookups/byte %.3f (headers %.3f), rejected %.3f, false positives %.3f"
          % (filter_kb, k, depths, T, far / NB, farhdr / NB, rejected / NB, fp / NB), flush=True)

run(0, 0, ())
for kbs, k in ((24, 2), (32, 2), (32, 3), (48, 3), (64, 3)):
    run(kbs, k, (3,))
run(64, 3, (3, 4))
