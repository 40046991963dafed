"""CPU model of k2_traverse's lookups on a cfg-3 sample: where do the lookups
that leave LDS go (by state depth, probe hit/miss, fail header)?"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from aha_amd import AC, synth

kb, ko, nf = synth.keys(3)
ac = AC.compile_packed(kb, ko, host_only=True)
info = ac.info
slots = ac.export(0, np.uint32)
T = info["lds_slots"]
n = slots.size
label = slots & 0xFF
base = (slots >> 8) & 0x3FFFFF
# depth of every state (keyed by base) via level-wise BFS
depth = np.full(n, -1, dtype=np.int32)
depth[0] = 0
level = np.array([0], dtype=np.int64)
labs = np.arange(1, 256, dtype=np.int64)
d = 0
counts = []
while level.size:
    counts.append(level.size)
    nxt = []
    for i in range(0, level.size, 1 << 16):
        Bs = level[i:i + (1 << 16)]
        idx = Bs[:, None] ^ labs[None, :]
        ok = label[idx] == labs[None, :]
        nxt.append(base[idx[ok]].astype(np.int64))
    level = np.concatenate(nxt) if nxt else np.array([], dtype=np.int64)
    d += 1
    depth[level] = d
print("states per depth", counts[:10], "T", T, "slots", n)
for dd in range(1, 7):
    b = np.nonzero(depth == dd)[0]
    print("depth", dd, "header slots: min", b.min(), "max", b.max(), "in LDS", int((b < T).sum()), "of", b.size)

NB = 1 << 21
corpus, doc = synth.corpus(3, kb, ko, nf, n_bytes=NB)
sl = slots.tolist()
dep = depth.tolist()
cat = collections.Counter()
B = 0; fr = 0; hdr = False
i = 0
text = corpus.tobytes()
trips = 0
FR = 0x40000000
while i < NB:
    b = text[i]
    trips += 1
    idx = B if hdr else (B ^ b)
    en = sl[idx]
    where = "lds" if idx < T else "l2"
    if hdr:
        cat[("hdr", where, dep[B])] += 1
        B = (en >> 8) & 0x3FFFFF; fr = en & FR; hdr = False
        continue
    m = (en & 0xFF) == b
    cat[("probe", where, dep[B], "hit" if m else "miss")] += 1
    if m:
        B = (en >> 8) & 0x3FFFFF; fr = en & FR; i += 1
        continue
    if B == 0 or fr:
        e0 = sl[b]
        if (e0 & 0xFF) == b:
            B = (e0 >> 8) & 0x3FFFFF; fr = e0 & FR
        else:
            B = 0; fr = 0
        i += 1
    else:
        hdr = True
print("trips/byte %.3f" % (trips / NB))
tot_l2 = sum(v for k, v in cat.items() if k[1] == "l2")
print("lookups leaving LDS per byte %.3f" % (tot_l2 / NB))
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print("%-34s %.4f /byte" % (str(k), v / NB))
