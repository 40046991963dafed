"""CPU twin of the character-level engine (aha_amd/csrc/unit.hpp, scan_unit.hip): interprets the unit image the library
built (aha_ac_export) exactly as the kernel does -- table-driven unit decoding into the dense alphabet, one probe per
unit behind the state word's filter, fail links carried by the entries or fetched from headers in a trip of their own,
group records of the big states, root table -- and
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
        self.big_lo, self.n_low = info["unit_big_lo"], info["unit_n_low"]
        self.g0 = self.n_low - (self.n_low >> 5)
        self.bb = info["unit_base_bits"]  # 22, or 23 for a large image: base | filter (29 - bb bits) | F1 | NFR | END
        self.bmask, self.nf = (1 << self.bb) - 1, 29 - self.bb
        assert self.slots.size == self.n_slots and self.n_slots % (1 << 15) == 0 and self.n_low % 32 == 0
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
                B = E & self.bmask
                hdr = ((E >> 29) & 3) == 1  # header pending (F1 without NFR: no entry holds that)
                grp = B >= self.big_lo and code >= self.n_low and not hdr
                se = 0 if hdr else (self.g0 + (code >> 5) if grp else code)  # what the probe is keyed by
                lo = hi = 0
                # (the classes beyond the filter's stored bits always probe: 7, and 6 with 23-bit bases)
                probe = good and B != 0 and bool((((E >> self.bb) | (1 << self.nf)) >> min(code & 7, self.nf)) & 1)
                if probe:
                    self.probes += 1
                    e = int(self.slots[B ^ se])
                    lo, hi = e & 0xFFFFFFFF, e >> 32
                symhit = probe and not grp and (hi & 0xFFFF) == se
                if hdr:  # the header: the fail state's word; the unit is tried again there
                    assert symhit and lo & self.bmask, "missing header"
                    E = lo
                    continue
                if symhit:
                    E = lo
                    break
                if grp and probe and (lo >> (code & 31)) & 1:  # a big state continues on this high symbol: its child's slot
                    child_slot = hi + bin(lo & ((1 << (code & 31)) - 1)).count("1")
                    E = (child_slot ^ code) | (((1 << self.nf) - 1) << self.bb)
                    continue
                if not good or not (E >> 30) & 1:  # the fail link is the root (or nothing matches): its table
                    E = int(self.root[code])
                    break
                if (E >> 29) & 1:  # F1: the fail state is the one-character state of the symbol that led here
                    E = int(self.root[pc]) & 0x7FFFFFFF
                    assert E & self.bmask, "missing fail link"
                else:
                    E = B | (((1 << self.nf) - 1) << self.bb) | (1 << 29)  # fetch the header in the next trip
            pc = code
            p += L
            if E >> 31:
                k = int(self.end_key[E & self.bmask])
                assert k >= 0
                while k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    out.append((p - ln, p, k))
                    k = nxt
        self.trips = trips
        return out
