"""CPU twin of the character-level engine (aha_amd/csrc/unit.hpp, scan_unit.hip): interprets the unit image the library
built (aha_ac_export) exactly as the kernel does -- table-driven unit decoding into the dense alphabet, one probe per
unit behind the root entry's filter, fail links carried by the entries, headers for fail targets, root table -- and
expands the events with the key tables.  Test infrastructure: checks the image builder and the unit-level algorithm
against the oracle without a GPU."""
import numpy as np

from aha_amd import _native as N

BIAS, POISON = 1 << 17, 1 << 24
T0A, T0B = 0, 1024  # word offsets of the decode tables (unit.hpp)


class UnitSim:
    def __init__(self, ac):
        info = ac.info
        assert info["unit_enabled"], "the key set is not eligible for the unit image"
        self.slots = ac.export(N.AHA_IMG_UNIT_SLOTS, np.uint64)
        self.root = ac.export(N.AHA_IMG_UNIT_ROOT, np.uint32)
        self.end_key = ac.export(N.AHA_IMG_UNIT_END_KEY, np.int32)
        self.tab = ac.export(N.AHA_IMG_UNIT_TABLES, np.uint32)
        self.key_ln = ac.export(N.AHA_IMG_KEY_LN, np.uint32).reshape(-1, 2)
        self.n_slots = info["unit_slots"]
        assert self.slots.size == self.n_slots and self.n_slots % (1 << 16) == 0
        assert self.root.size == info["unit_syms"] and int(self.root[0]) == 0

    def unit_at(self, t, p, end):
        """(symbol, length) of the unit that starts at t[p]; end = end of the document.  Symbol 0: matches nothing."""
        b0 = t[p]
        b1 = t[p + 1] if p + 1 < len(t) else 0  # the kernel reads whatever follows; the document check comes after
        b2 = t[p + 2] if p + 2 < len(t) else 0
        a1, a2, base, lo = (int(x) for x in self.tab[T0A + 4 * b0:T0A + 4 * b0 + 4])
        span, want = int(self.tab[T0B + 2 * b0]), int(self.tab[T0B + 2 * b0 + 1])
        s = (base + int(self.tab[a1 // 4 + b1]) + int(self.tab[a2 // 4 + b2])) & 0xFFFFFFFF
        in_doc = p + want <= end
        whole = in_doc and s < POISON
        good = whole and ((s - lo) & 0xFFFFFFFF) < span
        return (s - BIAS if good else 0), (want if whole else 1)

    def match(self, text):
        """One document: list of (start, end, value) in the reference's order."""
        t = bytes(text)
        n = len(t)
        out = []
        B, fb, ffr = 0, 0, True
        flt = 0xFF  # filter of the current state (depth-1 states carry one, others pass everything)
        p = 0
        trips = 0
        self.probes = 0
        while p < n:
            code, L = self.unit_at(t, p, n)
            good = code != 0
            while True:  # the trips of this unit
                trips += 1
                lo = hi = 0
                hit = False
                if B != 0 and good and (flt >> (code & 7)) & 1:
                    self.probes += 1
                    e = int(self.slots[B ^ code])
                    lo, hi = e & 0xFFFFFFFF, e >> 32
                    hit = (hi & 0xFFFF) == code
                if hit:
                    B = lo & 0x1FFFFF
                    fb = ((lo >> 21) & 0x3FF) | (((hi >> 16) & 0x7FF) << 10)
                    ffr = bool((hi >> 27) & 1)
                    end = bool(lo >> 31)
                    flt = 0xFF
                    break
                if not good or B == 0 or fb == 0:  # the fail link is the root (or nothing matches): its table
                    r = int(self.root[code])
                    B, fb, ffr = r & 0x1FFFFF, 0, True
                    flt = (r >> 21) & 0xFF
                    end = good and bool(r >> 31)
                    break
                flt = 0xFF
                if ffr:  # fall to the fail state, whose own fail link is the root
                    B, fb, ffr = fb, 0, True
                    continue
                e = int(self.slots[fb])  # header of the fail state
                lo, hi = e & 0xFFFFFFFF, e >> 32
                assert (hi & 0xFFFF) == 0 and e != 0, "missing header"
                B = fb
                fb = ((lo >> 21) & 0x3FF) | (((hi >> 16) & 0x7FF) << 10)
                ffr = bool((hi >> 27) & 1)
            p += L
            if end:
                k = int(self.end_key[B])
                assert k >= 0
                while k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    out.append((p - ln, p, k))
                    k = nxt
        self.trips = trips
        return out
