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
        self.fail_tab = ac.export(N.AHA_IMG_UNIT_FAIL, np.uint32)
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
        E = 0  # the state as one word: base | filter (7 bits) << 22 | F1 << 29 | NFR << 30 | END << 31; 0 = the root
        pc = 0  # the symbol that led to it
        p = 0
        trips = 0
        self.probes = 0
        while p < n:
            code, L = self.unit_at(t, p, n)
            good = code != 0
            while True:  # the trips of this unit
                trips += 1
                hit = False
                B = E & 0x3FFFFF
                if good and B != 0 and ((((E >> 22) & 0x7F) | 0x80) >> (code & 7)) & 1:
                    self.probes += 1
                    e = int(self.slots[B ^ code])
                    lo, hi = e & 0xFFFFFFFF, e >> 32
                    hit = (hi & 0xFFFF) == code
                if hit:
                    E = lo
                    break
                if not good or not (E >> 30) & 1:  # the fail link is the root (or nothing matches): its table
                    E = int(self.root[code])
                    break
                if (E >> 29) & 1:  # F1: the fail state is the one-character state of the symbol that led here
                    E = int(self.root[pc]) & 0x7FFFFFFF
                else:
                    E = int(self.fail_tab[B])  # fall to the fail state, try the unit again there
                assert E & 0x3FFFFF, "missing fail link"
            pc = code
            p += L
            if E >> 31:
                k = int(self.end_key[E & 0x3FFFFF])
                assert k >= 0
                while k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    out.append((p - ln, p, k))
                    k = nxt
        self.trips = trips
        return out
