"""CPU twin of the skip-ahead traversal (aha_amd/csrc/scan_skip.hip; unit.hpp, MARKS): the marking pass and the walk that
starts only at marked positions, over the images the library built (aha_ac_export) -- the same rules in the same order
as the two kernels: ks_mark's per-piece chain over raw UTF-8 units with the pair filter, ks_traverse's jump / stay /
early-fail decisions, the pseudo jumps at the end of a lane's bitmap window, document boundaries, chunks with their
warm-up.  Test infrastructure: checks the formulation (src/aha/ac.cr:176-192 visited only where a two-character path can
start) against the oracle without a GPU."""
import numpy as np

from aha_amd import _native as N
from unitsim import UnitSim

KA, KB = 0x9E3779, 0x85EBCB
PIECE, ROW = 64, 80
M32 = 0xFFFFFFFF


def mul24(a, b):
    return ((a & 0xFFFFFF) * (b & 0xFFFFFF)) & M32


def sk_part(c0):
    g = mul24(c0, KB)
    return ((g >> 11) | (g << 21)) & M32


def sk_hash(part, c1, k1=KA):
    h = (mul24(c1, k1) + part) & M32
    return h ^ (h >> 16)


class SkipSim(UnitSim):
    def __init__(self, ac):
        super().__init__(ac)
        info = ac.info
        assert info["skip_filter_words"], "the key set has no mark filter (a one-character key?)"
        self.bloom = ac.export(N.AHA_IMG_UNIT_MARKS, np.uint32)
        self.log2 = int(self.bloom.size).bit_length() - 1
        assert self.bloom.size == info["skip_filter_words"] == 1 << self.log2
        assert self.bb == 22
        self.k1 = info["pair_hash_k1"] or KA
        self.warm = max(int(info["max_key_len"]) - 1, 0)

    # ---- ks_mark: one bit per byte position
    def marks(self, text):
        t = bytes(text)
        n = len(t)
        out = bytearray((n + 63) // 64 * 64 + 192)
        for g0 in range(0, n, PIECE):
            row = t[g0:g0 + ROW].ljust(ROW, b"\0")
            o, po, gp = 0, PIECE, 0
            while po < PIECE or o == 0:
                at = min(o, ROW - 8)
                x = int.from_bytes(row[at:at + 4], "little")
                b0 = x & 0xFF
                s = 2 if (b0 & 0xE0) == 0xC0 else (3 if (b0 & 0xF0) == 0xE0 else 1)
                cm = (0xC0C000 if s == 3 else (0xC000 if s == 2 else 0))
                if (x & cm) != (cm & 0x808080):  # a lead byte without its continuation bytes: a one-byte unit
                    s = 1
                c = x & ((1 << (8 * s)) - 1)
                h = sk_hash(gp, c, self.k1)
                gp = sk_part(c)
                w = int(self.bloom[h >> (32 - self.log2)])
                m = (1 << (h & 31)) | (1 << ((h >> 5) & 31))
                if po < PIECE and (w & m) == m:
                    out[g0 + po] = 1
                po = o
                o += s
        return out

    # ---- ks_traverse, one lane: the chunk [a, e) of a batch (doc = document offsets), events as (doc, end, state base)
    def walk_chunk(self, t, doc, mk, a, e, warm, trips):
        n = len(t)
        D = len(doc) - 1
        dn = int(np.searchsorted(doc, a, side="left"))  # first boundary at or behind a
        nb = int(doc[dn])
        pos = a
        doc_start = a
        if nb != a:
            doc_start = int(doc[dn - 1])
            pos = a - min(a - doc_start, warm)
        events = []
        E = pc = 0
        d1 = False
        mql = False  # mark at the start of the last consumed unit
        needpos = True

        def nextmark(frm, bw):
            """first marked position in [frm, 64 * bw + 124), else that limit (a pseudo jump); clamped to the document and
            the chunk"""
            lim = 64 * bw + 124
            p = frm
            while p < lim and not mk[p]:
                p += 1
            return min(p, nb, e)

        cur = None  # (code, L) of the unit at pos
        while pos < e:
            if pos == nb:  # a document starts here
                while nb == pos and dn <= D:
                    dn += 1
                    nb = int(doc[dn]) if dn <= D else 1 << 62
                doc_start = pos
                E = pc = 0
                d1 = False
                needpos = True
                if pos >= e:
                    break
            if needpos:  # (re)positioning from the root: the chunk's start, a document's start
                tgt = nextmark(pos, pos >> 6)
                pos = tgt
                needpos = False
                cur = None
                if pos >= e or pos == nb:
                    continue
                # jump: consume the unit at the target from the root
                c0, l0 = self.unit_at(t, pos, min(nb, n))
                E = int(self.root[c0])
                pc = c0
                d1 = E != 0
                mql = bool(mk[pos])
                pos += l0
                if pos >= e or pos == nb:
                    continue
                cur = self.unit_at(t, pos, min(nb, n))
            if cur is None:
                cur = self.unit_at(t, pos, min(nb, n))
            code, L = cur
            trips[0] += 1
            good = code != 0
            B = E & self.bmask
            hdr = ((E >> 29) & 3) == 1
            grp = B >= self.big_lo and code >= self.n_low and not hdr
            se = 0 if hdr else (self.g0 + (code >> 5) if grp else code)
            probe = good and B != 0 and bool((((E >> self.bb) | (1 << self.nf)) >> min(code & 7, self.nf)) & 1)
            lo = hi = 0
            if probe:
                ent = int(self.slots[B ^ se])
                lo, hi = ent & M32, ent >> 32
            symhit = probe and not grp and (hi & 0xFFFF) == se
            hit = symhit and not hdr
            redir = grp and probe and bool((lo >> (code & 31)) & 1)
            nfr, f1 = bool((E >> 30) & 1), bool((E >> 29) & 1)
            # a miss in a state that fails to a one-character state whose own start is NOT marked: that state has no
            # transition on this unit either (the mark filter has no false negatives), so the root's table answers now
            viaroot = not symhit and not redir and (not nfr or not good or (f1 and not mql))
            rt = int(self.root[code])
            consumed = hit or viaroot
            p0 = pos
            if symhit:
                E = lo
            elif redir:
                child_slot = hi + bin(lo & ((1 << (code & 31)) - 1)).count("1")
                E = (child_slot ^ code) | (((1 << self.nf) - 1) << self.bb)
            elif viaroot:
                E = rt
            elif f1:
                E = int(self.root[pc]) & 0x7FFFFFFF
                d1 = True
            else:
                E = B | (((1 << self.nf) - 1) << self.bb) | (1 << 29)  # header pending
            if hdr:
                d1 = False  # a header's fail state is neither the root nor a one-character state
            if consumed:
                pc = code
                pos = p0 + L
                mql = bool(mk[p0])
                d1 = viaroot and E != 0
                if (E >> 31) and a <= pos - 1 < e:
                    events.append((pos, E & self.bmask))
                # the next unit (the kernel decodes it while the probe is in flight)
                ncur = self.unit_at(t, pos, min(nb, n)) if pos < min(nb, n) else (0, 1)
                # early fail: the state just entered fails to a one-character state and its filter says the next unit does
                # not continue it
                if hit and pos < nb and ((E >> 29) & 3) == 3:
                    nfc = ncur[0] & 7
                    if nfc < 7 and not (E >> (self.bb + nfc)) & 1:
                        E = rt & 0x7FFFFFFF
                        d1 = True
                cur = ncur
                shallow = E == 0 or d1
                if shallow and not (d1 and mql):  # jump to the next mark at or behind pos
                    tgt = nextmark(pos, p0 >> 6)
                    pos = tgt
                    cur = None
                    if pos >= e or pos == nb:
                        E = pc = 0
                        d1 = False
                        continue
                    c0, l0 = self.unit_at(t, pos, min(nb, n))
                    E = int(self.root[c0])
                    pc = c0
                    d1 = E != 0
                    mql = bool(mk[pos])
                    pos += l0
                    if pos < e and pos != nb:
                        cur = self.unit_at(t, pos, min(nb, n))
            # (not consumed: the same unit is tried again in the fail state, under the header, or as the group's child)
        return events

    def match_batch(self, text, doc, S=4096, trips=None):
        """the batch as the kernels walk it -- marks, then chunks of S bytes, each by its own lane; returns the hits as one
        list of (document, start, end, value) in the reference's order"""
        t = bytes(text)
        n = len(t)
        doc = np.asarray(doc, dtype=np.int64)
        mk = self.marks(t)
        trips = trips if trips is not None else [0]
        hits = []
        for a in range(0, n, S):
            for end_abs, base in self.walk_chunk(t, doc, mk, a, min(a + S, n), self.warm, trips):
                d = int(np.searchsorted(doc, end_abs - 1, side="right")) - 1
                k = int(self.end_key[base])
                assert k >= 0
                endd = end_abs - int(doc[d])
                while k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    hits.append((d, endd - ln, endd, k))
                    k = nxt
        return hits
