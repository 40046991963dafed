"""Tests-only interpreter of the exported automaton image (aha_ac_export):
walks the XOR double array exactly as kernels.hip does, on the CPU, so the
host logic (trie build, fail links, placement, encoding) can be checked
against the oracle without a GPU.  Never imported by the product package."""
import numpy as np

W_END, W_FAILROOT, W_BASE_MASK = 0x80000000, 0x40000000, 0x3FFFFFFF
C_END, C_BASE_SHIFT, C_BASE_MASK = 0x80000000, 8, 0x3FFFFF


class ImageSim:
    def __init__(self, ac):
        info = ac.info
        self.compact = info["slot_bytes"] == 4
        self.slots = ac.export(0, np.uint32 if self.compact else np.uint64)
        self.end_key = ac.export(1, np.int32) if self.compact else None
        ln = ac.export(2, np.uint32).reshape(-1, 2)
        self.key_len = ln[:, 0].astype(np.int64)
        self.key_next = ln[:, 1].astype(np.int32)
        self.key_cnt = ac.export(3, np.uint32)
        self.root = 0
        assert self.slots.size == info["n_slots"]
        # shadow fail links (include/aha_hip.h, aha_ac_info_t): all 0 = every state has a header
        self.s1_lo, self.s2_lo, self.hdr_lo = info["fail_s1_lo"], info["fail_s2_lo"], info["fail_hdr_lo"]

    def _probe(self, B, b):
        e = int(self.slots[B ^ b])
        if self.compact:
            if (e & 0xFF) == b:
                nB = (e >> C_BASE_SHIFT) & C_BASE_MASK
                return nB, (int(self.end_key[nB]) if e & C_END else -1)
            return None
        lo, hi = e & 0xFFFFFFFF, e >> 32
        if (hi & 0xFF) == b:
            return lo & W_BASE_MASK, ((hi >> 8) if lo & W_END else -1)
        return None

    def _child_or_root(self, B, b):
        r = self._probe(B, b)
        return self.root if r is None else r[0]

    def _fail(self, B, last2=(0, 0)):
        """fails[nid] of a non-root state; last2 = the two bytes consumed before the current one."""
        if B >= self.hdr_lo:  # header slot (every state when the ranges are all 0)
            e = int(self.slots[B])
            return ((e >> C_BASE_SHIFT) & C_BASE_MASK) if self.compact else (e & W_BASE_MASK)
        if B < self.s1_lo:
            return self.root
        x, y = last2
        if B >= self.s2_lo:
            s1 = self._child_or_root(self.root, x)
            if s1 != self.root:
                s2 = self._child_or_root(s1, y)
                if s2 != self.root:
                    return s2
        return self._child_or_root(self.root, y)

    def match(self, text):
        out = []
        B = self.root
        t = bytes(text)
        for i, b in enumerate(t):
            if b == 0:
                B = self.root
                continue
            key = -1
            while True:
                r = self._probe(B, b)
                if r is not None:
                    B, key = r
                    break
                if B == self.root:
                    break
                B = self._fail(B, (t[i - 2] if i >= 2 else 0, t[i - 1] if i >= 1 else 0))
            if key >= 0:
                k = key
                n = 0
                while k >= 0:
                    out.append((i + 1 - int(self.key_len[k]), i + 1, k))
                    n += 1
                    k = int(self.key_next[k])
                assert n == int(self.key_cnt[key])
        return out


# ---------------------------------------------------------------------------
# CPU twin of k3_traverse (scan_v2.hip, filter mode): one lane walking a whole
# sequence with the FAST / PEND / EXACT modes, on the exported Bloom filter and
# exact set.  Checks the filter construction and the mode logic without a GPU.
def _fhash(B, w):
    h = ((B * 0x9E3779B1) ^ (w * 0x85EBCA6B)) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 13
    return h


def _fmask(h):
    g = (h * 0x297A2D39) & 0xFFFFFFFF
    return (1 << (g >> 27)) | (1 << ((g >> 22) & 31)) | (1 << ((g >> 17) & 31))


def _fkey(n, x1, x2, x3):
    return (n << 24) | x1 | (x2 << 8) | (x3 << 16)


class FilterSim(ImageSim):
    def __init__(self, ac):
        super().__init__(ac)
        info = ac.info
        self.d0 = info["filter_d0"]
        assert self.d0 > 0, "filter mode is off for this automaton"
        self.T = info["lds_slots"]
        self.TB = info["boundary_end"]
        self.bloom = ac.export(5, np.uint32)
        self.xset = ac.export(6, np.uint64)
        self.stats = {"fast": 0, "pend": 0, "pend_fp": 0, "exact": 0, "global": 0}

    def _bloom(self, B, w):
        h = _fhash(B, w)
        m = _fmask(h)
        return (int(self.bloom[(h * self.bloom.size) >> 32]) & m) == m

    def _xhas(self, B, w):
        k = (B << 32) | w
        mask = self.xset.size - 1
        i = _fhash(B, w) & mask
        while True:
            v = int(self.xset[i])
            if v == k:
                return True
            if v == 0:
                return False
            i = (i + 1) & mask

    def _flags(self, e):
        if self.compact:
            return bool(e & 0x40000000)
        return bool((e & 0xFFFFFFFF) & W_FAILROOT)

    def _entry(self, idx):
        return int(self.slots[idx])

    def _dec(self, e, b):
        """-> (match, base, end, failroot, key)"""
        if self.compact:
            if (e & 0xFF) != b:
                return False, 0, False, False
            return True, (e >> C_BASE_SHIFT) & C_BASE_MASK, bool(e & C_END), bool(e & 0x40000000)
        lo, hi = e & 0xFFFFFFFF, e >> 32
        if (hi & 0xFF) != b:
            return False, 0, False, False
        return True, lo & W_BASE_MASK, bool(lo & W_END), bool(lo & W_FAILROOT)

    def _key_at(self, B, e):
        return int(self.end_key[B]) if self.compact else (e >> 32) >> 8

    def match(self, text):
        t = bytes(text)
        n = len(t)
        out = []
        root, T, TB, d0 = 0, self.T, self.TB, self.d0
        B, fr, mode, pos = root, False, 2, 0
        hist, since, replay, skipf = [], 0, 0, False
        st = self.stats

        def emit(i, key):
            k = key
            while k >= 0:
                out.append((i + 1 - int(self.key_len[k]), i + 1, k))
                k = int(self.key_next[k])

        def hdr(Bx):
            e = self._entry(Bx)
            if self.compact:
                return (e >> C_BASE_SHIFT) & C_BASE_MASK, bool(e & 0x40000000)
            return (e & 0xFFFFFFFF) & W_BASE_MASK, bool((e & 0xFFFFFFFF) & W_FAILROOT)

        guard = 0
        while pos < n:
            guard += 1
            assert guard < 40 * n + 1000, "no progress"
            consumed = False
            b = 0
            if mode == 0:
                st["fast"] += 1
                b = t[pos]
                if b == 0:
                    B, fr, consumed = root, False, True
                elif B < T:
                    e = self._entry(B ^ b)
                    m, nB, end, nfr = self._dec(e, b)
                    if m:
                        B, fr, consumed = nB, nfr, True
                        if end:
                            emit(pos, self._key_at(nB, e))
                    elif B == root:
                        consumed = True
                    elif fr:
                        B, fr = root, False
                    else:
                        assert B < T
                        B, fr = hdr(B)
                else:
                    assert B < TB, "deep state in FAST mode"
                    suspect = False
                    if not skipf:
                        room = n - pos
                        x1 = b
                        suspect = self._bloom(B, _fkey(1, x1, 0, 0))
                        if room >= 2:
                            suspect = suspect or self._bloom(B, _fkey(2, x1, t[pos + 1], 0))
                            if room >= 3:
                                suspect = suspect or self._bloom(B, _fkey(3, x1, t[pos + 1], t[pos + 2]))
                    if suspect:
                        mode = 1
                    else:
                        skipf = False
                        if fr:
                            B, fr = root, False
                        else:
                            nb_, nfr_ = root, False
                            for L in range(d0 - 1, 0, -1):
                                sB, sfr, ok = root, False, True
                                for j in range(L, 0, -1):
                                    hb = hist[-j]
                                    e = self._entry(sB ^ hb)
                                    m, nB, _, nfr = self._dec(e, hb)
                                    if m:
                                        sB, sfr = nB, nfr
                                    else:
                                        ok = False
                                        break
                                if ok:
                                    nb_, nfr_ = sB, sfr
                                    break
                            B, fr = nb_, nfr_
            elif mode == 1:
                st["pend"] += 1
                room = n - pos
                x1 = t[pos]
                real = self._xhas(B, _fkey(1, x1, 0, 0))
                if room >= 2:
                    real = real or self._xhas(B, _fkey(2, x1, t[pos + 1], 0))
                    if room >= 3:
                        real = real or self._xhas(B, _fkey(3, x1, t[pos + 1], t[pos + 2]))
                if not real:
                    mode, skipf = 0, True
                    st["pend_fp"] += 1
                else:
                    mode, replay, B, fr = 2, min(since, d0 + 2), root, False
            else:
                st["exact"] += 1
                b = hist[-replay] if replay else t[pos]
                took = False
                if b == 0:
                    B, fr, took = root, False, True
                else:
                    idx = B ^ b
                    if idx >= T:
                        st["global"] += 1
                    e = self._entry(idx)
                    m, nB, end, nfr = self._dec(e, b)
                    if m:
                        B, fr, took = nB, nfr, True
                        if end and not replay:
                            emit(pos, self._key_at(nB, e))
                    elif B == root:
                        took = True
                    elif fr:
                        B, fr = root, False
                    else:
                        if B >= T:
                            st["global"] += 1
                        B, fr = hdr(B)
                if took:
                    if replay:
                        replay -= 1
                    else:
                        consumed = True
                        if B < TB:
                            mode = 0
            if consumed:
                hist.append(b)
                if len(hist) > 8:
                    hist.pop(0)
                since = min(since + 1, 15)
                pos += 1
        return out
