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
