"""CPU twin of the prefix-filter engine's formulation (aha_amd/csrc/scan_filter.hip): src/aha/ac.cr:176-192 without a state
carried through the text.  The reference's state after byte j is the longest suffix of the text that is a trie path, i.e.
the goto walk of the EARLIEST start that is still alive at j.  So:
  1. kf_filter: the positions whose next D bytes pass the Bloom filter over the keys' first D bytes (no false negatives);
  2. kf_walk:   the goto walk from each of them over the byte-level image, stopped at its document's end;
  3. an END step of a walk is an event of the reference iff no EARLIER start's walk is still alive at that byte -- the
     exclusive prefix maximum of the walks' reaches (starts the filter rejects die within D - 1 bytes: they end no key, keys
     have at least D bytes, and they outlive no later start's END).
Events in (start, step) order are in position order.  The walks use the exported image (tests/imgsim.py: the same probes the
kernel makes); the filter is rebuilt from the keys with the kernel's hash (image.hpp kFilterMul).  Test infrastructure."""
import numpy as np

from imgsim import ImageSim

FILTER_MUL = 0x9E3779B1


class FilterSim(ImageSim):
    def __init__(self, ac, keys):
        super().__init__(ac)
        info = ac.info
        self.D = info["filter_prefix_bytes"]
        assert self.D, "the key set has no prefix filter"
        self.words = info["filter_words"]
        self.lg = self.words.bit_length() - 1
        self.bloom = [0] * self.words
        for k in keys:
            w, m = self._word_mask(int.from_bytes(bytes(k[:self.D]), "little"))
            self.bloom[w] |= m
        self.max_len = info["max_key_len"]
        self.keys = list(keys)

    def _word_mask(self, w4):
        h = (w4 * FILTER_MUL) & 0xFFFFFFFF
        return h >> (32 - self.lg), (1 << ((h >> (32 - self.lg - 5)) & 31)) | (1 << ((h >> (32 - self.lg - 10)) & 31))

    def candidates(self, t):
        """kf_filter: positions whose D bytes (zero-padded behind the text) pass the filter"""
        out = []
        pad = bytes(t) + b"\0" * 4
        for p in range(len(t)):
            w, m = self._word_mask(int.from_bytes(pad[p:p + self.D], "little"))
            if self.bloom[w] & m == m:
                out.append(p)
        return out

    def match_batch(self, text, doc):
        """hits of the whole batch as (document, start, end, value), by the start-parallel rule"""
        t = bytes(text)
        doc = [int(x) for x in doc]
        hits = []
        for d in range(len(doc) - 1):
            ds, de = doc[d], doc[d + 1]
            seg = t[ds:de]
            reach = -1  # furthest byte (inclusive, document offset) an earlier start's walk is alive at
            for q in self.candidates(seg):
                B, j, ends = self.root, q, []
                while j < len(seg) and seg[j] != 0:  # (NUL: the reference's state falls to the root; keys hold none)
                    r = self._probe(B, seg[j])
                    if r is None:
                        break
                    B, key = r
                    if key >= 0:
                        ends.append((j, key))
                    j += 1
                last_alive = j - 1  # the walk is alive at q .. j - 1
                for end_pos, key in ends:
                    if end_pos > reach:  # no earlier start alive at this byte: the reference's state IS this walk's
                        k = key
                        while k >= 0:
                            ln = int(self.key_len[k])
                            hits.append((d, end_pos + 1 - ln, end_pos + 1, k))
                            k = int(self.key_next[k])
                reach = max(reach, last_alive)
        return hits

    # ---- char offsets (kf_walk<.., CHARS> + k2d_expand<.., true>): a character = a byte outside 0x80..0xBF.  Per chunk of S
    # bytes the continuation bytes as a 64-bit mask and a running count per 64 bytes; an event carries the characters counted
    # from its document's start if that lies inside the chunk ("exact"), else from the chunk's start, and the expansion adds
    # what lies between the document's start and the chunk: lead_base (scan of the chunks' counts), chunk_doc0, doc_lead_rank.
    def match_batch_chars(self, text, doc, S=4096):
        t = bytes(text)
        n = len(t)
        doc = [int(x) for x in doc]
        n_chunks = (n + S - 1) // S
        cmask, cpre, lead_cnt, chunk_doc0 = [], [], [], []
        doc_lead_rank = {}

        def lead(c, o):  # characters that start in [c * S, c * S + o)
            w = min(o >> 6, S // 64 - 1)
            bit = o - w * 64
            low = (1 << 64) - 1 if bit >= 64 else (1 << bit) - 1
            return o - (cpre[c][w] + bin(cmask[c][w] & low).count("1"))

        for c in range(n_chunks):
            a, e = c * S, min(c * S + S, n)
            chunk = t[a:a + S].ljust(S, b"\0")
            m = [sum(1 << i for i in range(64) if (chunk[w * 64 + i] & 0xC0) == 0x80) for w in range(S // 64)]
            pre, run = [], 0
            for w in range(S // 64):
                pre.append(run)
                run += bin(m[w]).count("1")
            cmask.append(m)
            cpre.append(pre)
            lead_cnt.append(lead(c, e - a))
            dn = next(i for i, b in enumerate(doc) if b >= a)  # first boundary at or behind the chunk start
            chunk_doc0.append(dn if doc[dn] == a else dn - 1)
            for d in range(dn, len(doc)):
                if doc[d] >= e:
                    break
                doc_lead_rank[d] = lead(c, doc[d] - a)
        lead_base = [sum(lead_cnt[:c]) for c in range(n_chunks)]
        kc = [sum(1 for b in bytes(k) if (b & 0xC0) != 0x80) for k in self.keys]
        out = []
        for d, start, end, k in self.match_batch(text, doc):
            pos = doc[d] + end  # the event's end (exclusive) in the batch; its chunk is that of its last byte
            c = (pos - 1) // S
            a = c * S
            exact = doc[d] >= a
            y = ((lead(c, pos - a) - (lead(c, doc[d] - a) if exact else 0)) << 1) | (1 if exact else 0)
            # k2d_expand
            d0 = chunk_doc0[c]
            dchunk = doc[d0] // S
            lead_adj = lead_base[c] - (lead_base[dchunk] + doc_lead_rank[d0])
            end_c = (y >> 1) + (0 if y & 1 else lead_adj)
            out.append((d, end_c - kc[k], end_c, k))
        return out

