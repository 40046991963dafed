"""CPU twin of the character-level engine (aha_amd/csrc/unit.hpp, scan_unit.hip): interprets the unit image the library
built (aha_ac_export) exactly as the kernel does -- unit decoding, one probe per unit, fail links carried by the
entries, headers for fail targets, root table -- and expands the events with the key tables.  Test infrastructure:
checks the image builder and the unit-level algorithm against the oracle without a GPU."""
import numpy as np

from aha_amd import _native as N

U2, U3, BAD = 0x80, 0x880, 0x1FFFF


class UnitSim:
    def __init__(self, ac):
        info = ac.info
        assert info["unit_enabled"], "the key set is not eligible for the unit image"
        self.slots = ac.export(N.AHA_IMG_UNIT_SLOTS, np.uint64)
        self.root = ac.export(N.AHA_IMG_UNIT_ROOT, np.uint32)
        self.end_info = ac.export(N.AHA_IMG_UNIT_END_INFO, np.uint32)
        self.key_ln = ac.export(N.AHA_IMG_KEY_LN, np.uint32).reshape(-1, 2)
        self.n_slots = info["unit_slots"]
        assert self.slots.size == self.n_slots and self.n_slots % (1 << 17) == 0

    @staticmethod
    def unit_at(t, p, end):
        """(code, length) of the unit that starts at t[p]; end = end of the document."""
        b0 = t[p]
        if b0 == 0:
            return BAD, 1
        if b0 < 0x80:
            return b0, 1
        if (b0 & 0xE0) == 0xC0 and p + 1 < end and (t[p + 1] & 0xC0) == 0x80:
            return U2 + (((b0 & 0x1F) << 6) | (t[p + 1] & 0x3F)), 2
        if (b0 & 0xF0) == 0xE0 and p + 2 < end and (t[p + 1] & 0xC0) == 0x80 and (t[p + 2] & 0xC0) == 0x80:
            return U3 + (((b0 & 0x0F) << 12) | ((t[p + 1] & 0x3F) << 6) | (t[p + 2] & 0x3F)), 3
        return BAD, 1

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
            if code == BAD:
                B, fb, ffr, flt = 0, 0, True, 0xFF
                p += L
                continue
            while True:  # the trips of this unit
                trips += 1
                lo = hi = 0
                hit = False
                if B != 0 and (flt >> ((code ^ (code >> 4) ^ (code >> 9)) & 7)) & 1:
                    self.probes += 1
                    e = int(self.slots[B ^ code])
                    lo, hi = e & 0xFFFFFFFF, e >> 32
                    hit = (hi & 0x1FFFF) == code
                if hit:
                    B = lo & 0x1FFFFF
                    fb = ((lo >> 21) & 0x3FF) | (((hi >> 17) & 0x7FF) << 10)
                    ffr = bool((hi >> 28) & 1)
                    end = bool(lo >> 31)
                    flt = 0xFF
                    break
                if B == 0 or fb == 0:  # the fail link is the root: its table
                    r = int(self.root[code])
                    B, fb, ffr = r & 0x1FFFFF, 0, True
                    flt = (r >> 21) & 0xFF
                    end = bool(r >> 31)
                    break
                flt = 0xFF
                if ffr:  # fall to the fail state, whose own fail link is the root
                    B, fb, ffr = fb, 0, True
                    continue
                e = int(self.slots[fb])  # header of the fail state
                lo, hi = e & 0xFFFFFFFF, e >> 32
                assert (hi & 0x1FFFF) == 0 and e != 0, "missing header"
                B = fb
                fb = ((lo >> 21) & 0x3FF) | (((hi >> 17) & 0x7FF) << 10)
                ffr = bool((hi >> 28) & 1)
            p += L
            if end:
                x = int(self.end_info[B])
                assert x != 0xFFFFFFFF
                k = x & 0xFFFFFF
                while k != 0xFFFFFFFF and k >= 0:
                    ln, nxt = int(self.key_ln[k][0]), int(np.int32(self.key_ln[k][1]))
                    out.append((p - ln, p, k))
                    k = nxt
                    if k < 0:
                        break
        self.trips = trips
        return out
