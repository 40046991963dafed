"""Lab: engines 4 and 6 on cfg 3's KEYS over texts with different densities of two-character paths (marks): cfg 3's own text
(0.10 marks per byte: every pair of Latin / Cyrillic letters is a path), and a text of CJK characters only with the same share
of key tokens.  tools/lab_skip_text.py [bytes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aha_amd import AC, synth

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
blob, offs, nf = synth.keys(3)
rng = np.random.default_rng(5)


def cjk_text(n):
    """tokens of 1..8 random CJK characters, a key with p = 1/32, a space behind a token with p = 1/2; documents of ~1 MiB"""
    out = bytearray()
    keys = [bytes(blob[offs[i]:offs[i + 1]]) for i in rng.integers(0, len(offs) - 1, 4096)]
    cps = rng.integers(0x4E00, 0x9FA6, size=n // 3 + 16)
    enc = np.empty((cps.size, 3), dtype=np.uint8)
    enc[:, 0] = 0xE0 | (cps >> 12)
    enc[:, 1] = 0x80 | ((cps >> 6) & 63)
    enc[:, 2] = 0x80 | (cps & 63)
    i = 0
    while len(out) < n:
        k = int(rng.integers(1, 9))
        if rng.random() < 1 / 32:
            out += keys[int(rng.integers(0, len(keys)))]
        else:
            out += enc[i:i + k].tobytes()
            i += k
        if rng.random() < 0.5:
            out += b" "
    t = np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()
    # cut at a character boundary: blank out a partial character at the end
    while (t[-1] & 0xC0) == 0x80 or t[-1] >= 0xC0:
        t[-1] = 0x20
        if (t[-2] & 0xC0) != 0x80 and t[-2] < 0xC0:
            break
        t = np.concatenate([t[:-2], np.array([0x20, 0x20], dtype=np.uint8)]) if False else t
        t[-2] = 0x20 if (t[-2] & 0xC0) == 0x80 or t[-2] >= 0xC0 else t[-2]
        break
    doc = np.arange(0, n + 1, 1 << 20, dtype=np.uint64)
    if doc[-1] != n:
        doc = np.append(doc, np.uint64(n))
    return t, doc


texts = {"cfg3 text": synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)}
# (the generator above is slow in Python: a 16 MiB block repeated)
blk, _ = cjk_text(min(n_bytes, 1 << 24))
rep = np.tile(blk, n_bytes // blk.size)
texts["CJK-only text"] = (rep, np.append(np.arange(0, rep.size, 1 << 20, dtype=np.uint64), np.uint64(rep.size)))
for name, (corpus, doc) in texts.items():
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    out = torch.zeros((corpus.size // 8, 3), dtype=torch.int32, device="cuda")
    res = {}
    for sk in ("0", "1"):
        os.environ["AHA_SKIP"] = sk
        ac = AC.compile_packed(blob, offs)
        ac.set_profiling(True)
        ts = []
        for _ in range(5):
            h = ac.match_batch_device(dc, dd, out, None)
            tm = ac.last_timing()
            ts.append((tm["ms_total"], tm["ms_count"], tm["ms_scan"], tm["engine"]))
        ts.sort()
        m = ts[len(ts) // 2]
        res[sk] = (h, out[:h].cpu().numpy().tobytes())
        print(f"{name}, {corpus.size >> 20} MiB, AHA_SKIP={sk}: engine {m[3]} total {m[0]:.3f} ms (kernels {m[1]:.3f} + {m[2]:.3f}) hits {h}", flush=True)
    assert res["0"] == res["1"], "engines disagree"
