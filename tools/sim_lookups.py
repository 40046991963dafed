"""CPU model of k2_traverse's lookups on a cfg-3 sample: where do the lookups
that leave LDS go (by state depth, probe hit/miss, fail header)?  Follows the
kernel's trip incl. the shadow fail links (include/aha_hip.h, aha_ac_info_t)."""
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
S1, S2, HDR = info["fail_s1_lo"], info["fail_s2_lo"], info["fail_hdr_lo"]
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
print("states per depth", counts[:8], "T", T, "slots", n, "fail ranges", S1, S2, HDR)

NB = 1 << 21
INTRIP = os.environ.get('INTRIP', '1') == '1'
corpus, doc = synth.corpus(3, kb, ko, nf, n_bytes=NB)
sl = slots.tolist()
dep = depth.tolist()
cat = collections.Counter()
B = 0; fr = 0; hdr = False
r1 = 0; s2 = 0
i = 0
text = corpus.tobytes()
trips = 0
FR = 0x40000000
while i < NB:
    b = text[i]
    trips += 1
    idx = B if hdr else (B ^ b)
    en = sl[idx]
    where = "lds" if idx < T else "far"
    if hdr:
        cat[("hdr", where, dep[B])] += 1
        B = (en >> 8) & 0x3FFFFF; fr = en & FR; hdr = False
        continue
    m = (en & 0xFF) == b
    cat[("probe", where, dep[B], "hit" if m else "miss")] += 1
    e0 = sl[b]
    mr = (e0 & 0xFF) == b
    e2 = sl[((r1 >> 8) & 0x3FFFFF) ^ b]
    consumed = False
    if m:
        B = (en >> 8) & 0x3FFFFF; fr = en & FR; consumed = True
    elif B == 0 or fr:
        if mr: B = (e0 >> 8) & 0x3FFFFF; fr = e0 & FR
        else: B = 0; fr = 0
        consumed = True
    elif S1 <= B < HDR:
        sx = r1 if B < S2 else s2
        i3 = ((sx >> 8) & 0x3FFFFF) ^ b
        if INTRIP and i3 < T:  # resolve the rest of the fail chain in this trip (all rows in LDS)
            e3 = sl[i3]
            X = e3 if (e3 & 0xFF) == b else (e2 if (e2 & 0xFF) == b else (e0 if mr else 0))
            B = (X >> 8) & 0x3FFFFF; fr = X & FR; consumed = True
            cat[("intrip", "lds", dep[B], "")] += 1
        else:
            B = (sx >> 8) & 0x3FFFFF; fr = sx & FR
            if INTRIP: cat[("e3far", "", 0, "")] += 1
    else:
        hdr = True
    if consumed:
        s2 = e2 if (e2 & 0xFF) == b else (e0 if mr else 0)
        r1 = e0 if mr else 0
        i += 1
print("trips/byte %.3f" % (trips / NB))
tot_far = sum(v for k, v in cat.items() if k[1] == "far")
print("lookups leaving LDS per byte %.3f" % (tot_far / NB))
for k, v in sorted(cat.items(), key=lambda kv: -kv[1])[:18]:
    print("%-34s %.4f /byte" % (str(k), v / NB))
