"""CPU twin of the position-parallel engine (aha_amd/csrc/pp.hpp, scan_pp.hip): tests only.

PpSim interprets the pair table and the Bloom filter exactly as exported by the library (AHA_IMG_PP_T2 /
AHA_IMG_PP_BLOOM) for pass 1, and re-states pass 2 (exact walks of the items, prefix maximum of the reaches, exact
check of the boring starts in front of a candidate) on the independent dict trie of tests/pymodel.py.  Its hits must
equal the model's sequential automaton: that pins the engine's algorithm without a GPU."""
import numpy as np

from aha_amd import _native as N

K1, K2, K3 = 0x9E3779, 0x85EBCB, 0xC2B2AF
GUARD = 5
M32 = 0xFFFFFFFF


def pp_hash(lo, b4):
    m1 = ((lo & 0xFFFFFF) * K1) & M32
    h3 = ((((lo >> 8) & 0xFFFF) * K2) + m1) & M32
    h4 = (((lo >> 8) * K2) + m1) & M32
    h5 = (((b4 & 0xFF) * K3) + h4) & M32
    return m1, h3, h4, h5


def pp_word(h, words):
    return ((h >> 8) * ((words << 8) & 0xFFFFFF)) >> 32


def pp_mask(m1):
    g = m1 ^ (m1 >> 11)
    return (1 << (g & 7)) | (0x100 << ((g >> 8) & 7)) | (0x10000 << ((g >> 16) & 7)) | (0x1000000 << ((g >> 24) & 7))


class PpSim:
    def __init__(self, ac, model):
        self.info = ac.info
        assert self.info["pp_enabled"], "automaton does not meet the engine's preconditions"
        self.t2 = ac.export(N.AHA_IMG_PP_T2, np.uint32)
        self.bloom = ac.export(N.AHA_IMG_PP_BLOOM, np.uint32)
        self.m = model

    # ---- pass 1: which starts become items
    def items(self, t):
        n = len(t)
        pad = bytes(t) + b"\0" * 8
        out = {}
        words = self.bloom.size
        for j in range(n):
            b0, b1 = pad[j], pad[j + 1]
            code = (int(self.t2[b0 | ((b1 & 15) << 8)]) >> ((b1 >> 4) * 2)) & 3
            if not code:
                continue
            flags = 0x8000 if code & 2 else 0
            if code & 1:
                lo = int.from_bytes(pad[j:j + 4], "little")
                m1, h3, h4, h5 = pp_hash(lo, pad[j + 4])
                bm = pp_mask(m1)
                for bit, h in ((0x1000, h3), (0x2000, h4), (0x4000, h5)):
                    if (int(self.bloom[pp_word(h, words)]) & bm) == bm:
                        flags |= bit
            if flags:
                out[j] = flags
        return out

    def walk(self, t, j, end):
        """(L, [(depth, state)] of END nodes) of the trie walk from the root along t[j:end]."""
        s, L, ends = 0, 0, []
        for d in range(1, end - j + 1):
            b = t[j + d - 1]
            nxt = self.m.children[s].get(b) if b else None
            if nxt is None:
                break
            s, L = nxt, d
            if self.m.key_of[s] >= 0:
                ends.append((d, s))
        return L, ends

    def check_boring(self, t, items):
        """pass 1's guarantee: a start that is no item has no END node on its walk and a walk shorter than GUARD"""
        for j in range(len(t)):
            if j in items:
                continue
            L, ends = self.walk(t, j, len(t))
            assert not ends and L < GUARD, (j, L, ends)

    # ---- pass 2 on one document
    def match(self, text):
        t = bytes(text)
        items = self.items(t)
        self.check_boring(t, items)
        pos = sorted(items)
        cov, run, reach, cands = {}, 0, {}, []
        for j in pos:
            L, ends = self.walk(t, j, len(t))
            cov[j] = run
            run = max(run, j + L)
            reach[j] = L
            for d, s in ends:
                cands.append((j + d - 1, j, s))
        events = {}
        for i, j, s in cands:
            ok = cov[j] <= i
            back = 1
            while ok and back < GUARD - 1:
                jj = j - back
                ln = i - jj + 1
                if jj < 0 or ln > GUARD - 1:
                    break
                if jj not in items:
                    L, _ = self.walk(t, jj, i + 1)
                    if L == ln:
                        ok = False
                back += 1
            if ok:
                assert i not in events, "two states for one position"
                events[i] = s
        out = []
        for i in sorted(events):
            u = events[i]
            while True:
                k = self.m.key_of[u]
                out.append((i + 1 - len(self.m.keys[k]), i + 1, k))
                u = self.m.fail[u]
                if self.m.key_of[u] < 0:
                    break
        return out
