"""CPU twin of the pair engine (aha_amd/csrc/scan_pair.hip; unit.hpp, PAIR TABLE): the stateless pair pass, the deep walks,
the voiding and filling of the event regions -- the same rules in the same order as the kernels, over the tables the library
built (aha_ac_export).  src/aha/ac.cr:176-192 says: the state after a character is the longest suffix that is a trie path, i.e.
the goto walk of the earliest start still alive; an END state reports (ac.cr:183-185).  With no one-character key:
  * a position whose two characters are a path with an END state is an event (kp_pairs: filter, ONE table probe, key check) --
    unless a walk of three characters or more from an earlier start is still alive there;
  * such DEEP walks start where the two-character state has a transition on the third character (the table's child filter
    lets a superset through); kp_walk follows each from the root: its reach, its END states of depth >= 3;
  * kp_void removes the events a deep walk covers (everything that ends behind its pair and at or before its reach), kp_fill
    puts a walk's own END events -- those behind the reach of every earlier walk -- into the slots kp_pairs left behind the
    pair's event.
Test infrastructure: checks the formulation against the oracle without a GPU."""
import numpy as np

from aha_amd import _native as N
from skipsim import M32, mul24, sk_hash, sk_part
from unitsim import UnitSim

PIECE, ROW, TILE = 32, 44, 2048
KC, MIX = 0xC2B2AF, 0x2545F491
NPLACE = 3  # slots behind a deep candidate's pair event
NULL = 0x00FFFFFF  # a record that stands for no hit (count 0)


def pt_cls(raw):
    g = mul24(raw, KC)
    return (1 << (g >> 27)) | (1 << ((g >> 22) & 31))


class PairSim(UnitSim):
    def __init__(self, ac):
        super().__init__(ac)
        info = ac.info
        assert info["pair_table_log2"], "the key set has no pair table"
        self.lg = info["pair_table_log2"]
        self.G = info["pair_groups"]
        self.k1 = info["pair_hash_k1"]
        self.ptab = ac.export(N.AHA_IMG_UNIT_PAIRS, np.uint32).reshape(-1, 4)
        self.disp = ac.export(N.AHA_IMG_UNIT_PAIR_DISP, np.uint8)
        self.bloom = ac.export(N.AHA_IMG_UNIT_MARKS, np.uint32)
        self.blg = int(self.bloom.size).bit_length() - 1
        assert self.ptab.shape[0] == 1 << self.lg and self.disp.size == self.G
        self.key_cnt = ac.export(N.AHA_IMG_KEY_CNT, np.uint32)
        self.max_len = int(info["max_key_len"])

    def probe_pair(self, h, raw0, raw1):
        """the pair table's entry for (raw0, raw1) behind hash h, or None (kp_pairs: one load + the key check)"""
        d = int(self.disp[(h >> 7) & (self.G - 1)])
        t = (h * MIX) & M32
        sl = ((t >> (32 - self.lg)) + d * (((t << 1) | 1) & M32)) & ((1 << self.lg) - 1)
        e = self.ptab[sl]
        if (int(e[0]) & 0xFFFFFF) == raw0 and int(e[1]) == raw1:
            return int(e[0]) >> 24, np.int32(e[2]), int(e[3])  # c4, key of the END state (or -1), child filter
        return None

    # ---- kp_pairs: per tile the records in position order [x, end, kind] and the deep candidates (start, pair end, slot)
    def pairs(self, t, doc):
        n = len(t)
        tiles = []
        for t0 in range(0, n, TILE):
            recs, cands = [], []
            for g0 in range(t0, min(t0 + TILE, n), PIECE):
                row = t[g0:g0 + ROW].ljust(ROW, b"\0")
                dn = int(np.searchsorted(doc, g0, side="left"))
                nb = int(doc[dn]) if dn < len(doc) else 1 << 62
                o, po1, po2, gp, c1, c0, it = 0, PIECE, PIECE, 0, 0, 0, 0
                pend = None  # (hash, raw0, raw1) of the pair asked for in the iteration before
                while it == 0 or po1 < PIECE or po2 < PIECE:
                    it += 1
                    if g0 + o == nb:  # a document starts here: no pair across it (the pending one lies before it)
                        while dn < len(doc) and int(doc[dn]) == g0 + o:
                            dn += 1
                        nb = int(doc[dn]) if dn < len(doc) else 1 << 62
                        bnd = True
                    else:
                        bnd = False
                    at = min(o, ROW - 5)  # (two aligned dwords from at & ~3 stay inside the row)
                    x = int.from_bytes(row[at:at + 4], "little")
                    b0 = x & 0xFF
                    s = 2 if (b0 & 0xE0) == 0xC0 else (3 if (b0 & 0xF0) == 0xE0 else 1)
                    cm = (0xC0C000 if s == 3 else (0xC000 if s == 2 else 0))
                    if (x & cm) != (cm & 0x808080) or g0 + o + s > min(nb, n):  # no continuation bytes, or not inside the document
                        s = 1
                    c = x & ((1 << (8 * s)) - 1)
                    if pend is not None:
                        e = self.probe_pair(*pend)
                        if e is not None:
                            c4, key, cf = e
                            end = g0 + o
                            if key >= 0:
                                recs.append([int(key), end, "pair"])
                            if not bnd and (cf & pt_cls(c)) == pt_cls(c):
                                cands.append([g0 + po2, end, len(recs)])
                                for _ in range(NPLACE):
                                    recs.append([-1, end, "slot"])
                        pend = None
                    if bnd:
                        po1 = po2 = PIECE
                    h = sk_hash(gp, c, self.k1)
                    gp = sk_part(c)
                    w = int(self.bloom[h >> (32 - self.blg)])
                    m = (1 << (h & 31)) | (1 << ((h >> 5) & 31))
                    if po1 < PIECE and (w & m) == m:
                        pend = (h, c1, c)
                    po2, po1 = po1, o
                    c0, c1 = c1, c
                    o += s
            tiles.append((recs, cands))
        return tiles

    # ---- kp_walk: the goto walk from the root at `start` (ac.cr:176-192 without the fail links): its reach and END states
    def walk(self, t, start, dend):
        p, E, depth = start, 0, 0
        ends = []
        reach = start
        while p < dend:
            code, L = self.unit_at(t, p, dend)
            if code == 0:
                break
            if E == 0:
                nE = int(self.root[code])
                if nE == 0:
                    break
            else:
                nE = self._goto(E, code)
                if nE is None:
                    break
            E = nE
            p += L
            depth += 1
            reach = p
            if depth >= 3 and E >> 31:
                ends.append((p, E & self.bmask))
        return depth, reach, ends

    def _goto(self, E, code):
        B = E & self.bmask
        if B >= self.big_lo and code >= self.n_low:  # a big state's group record, then the child's own slot
            ent = int(self.slots[B + self.g0 + (code >> 5)])
            lo, hi = ent & M32, ent >> 32
            if not (lo >> (code & 31)) & 1:
                return None
            slot = hi + bin(lo & ((1 << (code & 31)) - 1)).count("1")
            ent = int(self.slots[slot])
            assert (ent >> 32) & 0xFFFF == code
            return ent & M32
        if not (((E >> self.bb) | (1 << self.nf)) >> min(code & 7, self.nf)) & 1:
            return None
        ent = int(self.slots[B ^ code])
        if (ent >> 32) & 0xFFFF != code:
            return None
        return ent & M32

    def match_batch(self, text, doc, stats=None):
        t = bytes(text)
        n = len(t)
        doc = np.asarray(doc, dtype=np.int64)
        tiles = self.pairs(t, doc)
        # kp_walk
        walks = []  # (tile, start, pair end, slot, depth, reach, ends)
        for ti, (recs, cands) in enumerate(tiles):
            for start, pend, slot in cands:
                d = int(np.searchsorted(doc, start, side="right")) - 1
                depth, reach, ends = self.walk(t, start, int(doc[d + 1]))
                assert len(ends) <= NPLACE, "more deep END states on one path than the pair engine has room for"
                walks.append((ti, start, pend, slot, depth, reach, ends))
        if stats is not None:
            stats["cands"] = len(walks)
            stats["deep"] = sum(1 for w in walks if w[4] >= 3)
            stats["records"] = sum(len(r) for r, _ in tiles)
        # kp_void: everything that ends behind a deep walk's pair and at or before its reach
        for ti, start, pend, slot, depth, reach, ends in walks:
            if depth < 3:
                continue
            for tj in (ti, ti + 1):
                if tj >= len(tiles):
                    break
                for r in tiles[tj][0]:
                    if pend < r[1] <= reach and r[2] == "pair":
                        r[0] = -1
        # kp_fill: a walk's own END events behind the reach of every earlier walk
        for wi, (ti, start, pend, slot, depth, reach, ends) in enumerate(walks):
            if depth < 3:
                continue
            R = 0
            for wj in range(wi - 1, -1, -1):
                if walks[wj][1] < start - self.max_len:
                    break
                if walks[wj][4] >= 3:
                    R = max(R, walks[wj][5])
            k = 0
            for end, base in ends:
                if end > R:
                    key = int(self.end_key[base])
                    assert key >= 0
                    tiles[ti][0][slot + k] = [key, end, "deep"]
                    k += 1
        # expansion (k2d_expand): the records in region order, NULL ones give nothing
        hits = []
        for recs, _ in tiles:
            for key, end, kind in recs:
                if key < 0:
                    continue
                d = int(np.searchsorted(doc, end - 1, side="right")) - 1
                endd = end - int(doc[d])
                k = key
                while k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    hits.append((d, endd - ln, endd, k))
                    k = nxt
        return hits
