"""CPU twin of the hash image (aha_amd/csrc/hash.hpp): interprets the tables the library built (aha_ac_export) the way
the kernels do -- raw UTF-8 characters, the pair filter, the pairs' perfect hash, the cuckoo table of the deeper
transitions and headers -- in two ways:
  * match():          the sequential walk of scan_hash.hip kh_traverse (fail links, headers, child filters)
  * match_parallel(): every character start on its own -- the pair filter picks the candidates, each candidate walks its
                      goto path alone, a hit of a later start is dropped when an earlier start's walk is still alive at
                      its end (the reference's state is the LONGEST suffix that is a trie path: src/aha/ac.cr:176-192)
Test infrastructure: checks the image builder and both formulations against the oracle without a GPU."""
import numpy as np

from aha_amd import _native as N

M32 = 0xFFFFFFFF
TAG, HDR, K2, MIX, MIX2, SALT = 1 << 24, 0xFFFFFF, 0x85EBCB, 0x2545F491, 0x9E3779B1, 0x5BD1E995


def mul24(a, b):
    return ((a & 0xFFFFFF) * (b & 0xFFFFFF)) & M32


def rot(g):
    return ((g >> 11) | (g << 21)) & M32


def cls(c):
    return mul24(c, K2) >> 27


def part_char(c):
    return rot(mul24(c, K2))


def part_base(b):
    return rot(mul24(b, K2)) ^ SALT


class HashSim:
    def __init__(self, ac):
        par = ac.export(N.AHA_IMG_HASH_PARAMS, np.uint32)
        assert par.size == 8, "the key set has no hash image"
        self.k1, self.n_groups, self.pair_log2, self.deep_log2 = (int(x) for x in par[:4])
        self.bloom = ac.export(N.AHA_IMG_HASH_BLOOM, np.uint32)
        self.disp = ac.export(N.AHA_IMG_HASH_DISP, np.uint8)
        self.pairs = ac.export(N.AHA_IMG_HASH_PAIRS, np.uint32).reshape(-1, 4)
        self.deep = ac.export(N.AHA_IMG_HASH_DEEP, np.uint32).reshape(-1, 4)
        self.end_key = ac.export(N.AHA_IMG_UNIT_END_KEY, np.int32)
        self.key_ln = ac.export(N.AHA_IMG_KEY_LN, np.uint32).reshape(-1, 2)
        assert self.bloom.size == 1 << 14 and self.disp.size == self.n_groups
        assert self.pairs.shape[0] == 1 << self.pair_log2 and self.deep.shape[0] == 1 << self.deep_log2
        self.probes = 0

    def key(self, part, c):
        h = (mul24(c, self.k1) + part) & M32
        return h ^ (h >> 16)

    def bloom_pass(self, h):
        m = (1 << (h & 31)) | (1 << ((h >> 5) & 31))
        return (int(self.bloom[h >> 18]) & m) == m

    def pair(self, c1, c2):
        """entry of the transition of c1's one-character state on c2, or None (behind the filter, like the kernel)"""
        h = self.key(part_char(c1), c2)
        if not self.bloom_pass(h):
            return None
        self.probes += 1
        t = (h * MIX) & M32
        d = int(self.disp[(h >> 7) & (self.n_groups - 1)])
        sl = ((t >> (32 - self.pair_log2)) + d * (((t << 1) | 1) & M32)) & ((1 << self.pair_log2) - 1)
        e = self.pairs[sl]
        return e if int(e[0]) == (TAG | c1) and (int(e[1]) & 0xFFFFFF) == c2 else None

    def deep_entry(self, base, c):
        h = self.key(part_base(base), c)
        t = (h * MIX) & M32
        self.probes += 1
        for sl in (t >> (32 - self.deep_log2), ((t * MIX2) & M32) >> (32 - self.deep_log2)):
            e = self.deep[sl]
            if int(e[0]) == base and (int(e[1]) & 0xFFFFFF) == c:
                return e
        return None

    @staticmethod
    def unit_at(t, p, end):
        """(character as a little-endian integer, length) of the unit at t[p]; a malformed or truncated unit is its first byte"""
        b0 = t[p]
        want = 2 if (b0 & 0xE0) == 0xC0 else 3 if (b0 & 0xF0) == 0xE0 else 1
        ok = p + want <= end and all((t[p + i] & 0xC0) == 0x80 for i in range(1, want))
        if not ok:
            return b0, 1
        return int.from_bytes(t[p:p + want], "little"), want

    def expand(self, out, E, p):
        k = int(self.end_key[E & 0x3FFFFF])
        assert k >= 0
        while k >= 0:
            ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
            out.append((p - ln, p, k))
            k = nxt

    def match(self, text):
        """The sequential walk (kh_traverse): one document."""
        t = bytes(text)
        n = len(t)
        out = []
        E, CF, pc = 0, 0, 0
        p = 0
        self.trips = 0
        while p < n:
            c, L = self.unit_at(t, p, n)
            while True:
                self.trips += 1
                B = E & 0x3FFFFF
                deep = B != 0
                hdr = ((E >> 29) & 3) == 1
                nfr, f1 = bool((E >> 30) & 1), bool((E >> 29) & 1)
                dpass = deep and not hdr and bool((CF >> cls(c)) & 1)
                fall = deep and not hdr and not dpass
                pairmode = (not deep) or (fall and nfr and f1)
                needhdr = hdr or (fall and nfr and not f1)
                if pairmode:
                    e = self.pair(pc, c)
                elif needhdr:
                    e = self.deep_entry(B, HDR)
                    assert e is not None, "missing header"
                elif dpass:
                    e = self.deep_entry(B, c)
                else:
                    e = None
                if e is not None and needhdr:
                    E, CF = int(e[2]), int(e[3])
                    continue
                if e is not None:
                    E, CF = int(e[2]), int(e[3])
                    consumed = True
                elif pairmode or not nfr:
                    E, consumed = 0, True
                elif f1:
                    E, consumed = 0, False
                else:
                    E, consumed = B | (1 << 29), False
                if consumed:
                    break
            pc = c
            p += L
            if E >> 31:
                self.expand(out, E, p)
        return out

    def match_parallel(self, text):
        """Every start on its own: candidates by the pair filter, goto walks, blocking by the earlier starts' reach."""
        t = bytes(text)
        n = len(t)
        walks = []  # (start, [(end position, state word)], reach)
        p = 0
        while p < n:
            if (t[p] & 0xC0) == 0x80:  # a stray continuation byte starts nothing a key holds
                p += 1
                continue
            c1, L1 = self.unit_at(t, p, n)
            if p + L1 < n:
                c2, L2 = self.unit_at(t, p + L1, n)
                e = self.pair(c1, c2)
                if e is not None:
                    q = p + L1 + L2
                    ends = []
                    E, CF = int(e[2]), int(e[3])
                    if E >> 31:
                        ends.append((q, E))
                    while q < n:
                        c, L = self.unit_at(t, q, n)
                        if not (CF >> cls(c)) & 1:
                            break
                        e = self.deep_entry(E & 0x3FFFFF, c)
                        if e is None:
                            break
                        E, CF = int(e[2]), int(e[3])
                        q += L
                        if E >> 31:
                            ends.append((q, E))
                    walks.append((p, ends, q))
            p += L1
        out = []
        reach = 0  # furthest position an earlier start's walk is alive at
        for _, ends, r in walks:
            for q, E in ends:
                if q > reach:
                    self.expand(out, E, q)
            reach = max(reach, r)
        return out
