"""Independent checker model (tests only): a dict-trie Aho-Corasick written
ONLY from the emission rules of SURVEY.md section 0.1 / 8 a4-a5, not from the
Cedar code path.  It cross-checks the C oracle (oracle/aha_oracle.c) and the
HIP path on randomized inputs.

Rules (reference src/aha/ac.cr:176-192, 265-278, 105-108):
  * state after byte i = longest suffix of text[0..i] that is a trie path;
  * if that state is not a key end: nothing is reported at i;
  * otherwise report its key, then t = fail(state); while t is a key end:
    report t's key, t = fail(t); stop at the first non-end t;
  * a NUL input byte resets the state to root (defined-behaviour contract).
"""


class ModelError(Exception):
    pass


def _b(x):
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


class ModelAC:
    def __init__(self, keys):
        keys = [_b(k) for k in keys]
        self.keys = keys
        self.children = [dict()]
        self.key_of = [-1]
        self.depth = [0]
        for idx, k in enumerate(keys):
            if len(k) == 0:
                raise ModelError("empty")
            if 0 in k:
                raise ModelError("zero")
            s = 0
            for b in k:
                nxt = self.children[s].get(b)
                if nxt is None:
                    nxt = len(self.children)
                    self.children.append(dict())
                    self.key_of.append(-1)
                    self.depth.append(self.depth[s] + 1)
                    self.children[s][b] = nxt
                s = nxt
            if self.key_of[s] >= 0:
                raise ModelError("dup")
            self.key_of[s] = idx
        n = len(self.children)
        self.fail = [0] * n
        order = []
        queue = list(self.children[0].values())
        while queue:
            nq = []
            for s in queue:
                order.append(s)
                for b, c in self.children[s].items():
                    f = self.fail[s]
                    while f and b not in self.children[f]:
                        f = self.fail[f]
                    t = self.children[f].get(b, 0)
                    self.fail[c] = t if t != c else 0
                    nq.append(c)
            queue = nq

    def _step(self, s, b):
        if b == 0:
            return 0
        while s and b not in self.children[s]:
            s = self.fail[s]
        return self.children[s].get(b, 0)

    def match(self, text, sep=None, chars=None):
        """sep: None or (size, set_bits).  Returns list of (start,end,value)."""
        if chars is None:
            chars = isinstance(text, str)
        t = _b(text)

        def blocked(ch):
            if sep is None:
                return False
            size, bits = sep
            return ch < size and ch not in set(bits)

        out = []
        s = 0
        for i, b in enumerate(t):
            s = self._step(s, b)
            if self.key_of[s] < 0:
                continue
            if i + 1 < len(t) and blocked(t[i + 1]):
                continue
            u = s
            while True:
                k = self.key_of[u]
                st = i + 1 - len(self.keys[k])
                if not (st > 0 and blocked(t[st - 1])):
                    out.append((st, i + 1, k))
                u = self.fail[u]
                if self.key_of[u] < 0:
                    break
        if chars:
            cmap = []
            ci = -1
            for b in t:
                if (b & 0xC0) != 0x80:
                    ci += 1
                cmap.append(ci)
            out = [(cmap[a], cmap[e - 1] + 1, v) for (a, e, v) in out]
        return out

    def match_longest(self, text, intersectable=False, chars=None, stale=()):
        """match_longest_ / fetch_one restated from src/aha/ac.cr:118-143, 249-263.  is_end? = "really ends a key" or
        the state's string is in `stale`: Cedar's stale END flags (cedar.cr:642-648) are a property of its slot
        history, not of the key set, so the model takes them as an input -- the set of byte strings whose node holds
        one -- and treats such a state as an end for which fetch_one yields nothing.
        A NUL byte: a Cedar node that holds a value AND has children keeps the value in a child with label 0
        (cedar.cr:441-447 finds it like any child), a node without children of its own, for which is_end? holds
        (cedar.cr:657-660) and fetch_one yields nothing: it replaces the pending end; the BFS of compile skips it
        (cedar.cr:450-463), so its fail link is unset and the byte after it is consumed at the root without a goto
        (defined for intersectable = false; the same is pinned for true, where the reference reads array[-1])."""
        if chars is None:
            chars = isinstance(text, str)
        t = _b(text)
        out = []
        nid, prev_i, prev_nid = 0, -1, -1
        stale_nodes = set()
        for path in stale:
            n = 0
            for b in path:
                n = self.children[n][b]
            stale_nodes.add(n)

        def fetch_one(i, n):
            k = self.key_of[n] if n >= 0 else -1
            if k >= 0:
                out.append((i - len(self.keys[k]) + 1, i + 1, k))

        VALUE_NODE = -2  # the label-0 child of a state that ends a key and has children
        for i, b in enumerate(t):
            while True:
                if nid == VALUE_NODE:  # no goto from it, no fail link: the pending (empty) end is yielded, the byte is gone
                    prev_i = -1
                    nid = 0
                    break
                if b == 0 and self.key_of[nid] >= 0 and self.children[nid]:
                    nid = VALUE_NODE
                    prev_i, prev_nid = i, nid
                    break
                nxt = self.children[nid].get(b) if b else None
                if nxt is not None:
                    nid = nxt
                    if self.key_of[nid] >= 0 or nid in stale_nodes:
                        prev_i, prev_nid = i, nid
                    break
                if prev_i != -1:
                    fetch_one(prev_i, prev_nid)
                    prev_i = -1
                    if not intersectable:
                        nid = 0
                if nid == 0:
                    break
                nid = self.fail[nid]
        if prev_i != -1:
            fetch_one(prev_i, prev_nid)
        if chars:
            cmap, ci = [], -1
            for b in t:
                if (b & 0xC0) != 0x80:
                    ci += 1
                cmap.append(ci)
            out = [(cmap[a], cmap[e - 1] + 1, v) for (a, e, v) in out]
        return out

    def match_chars_sep(self, chars, sep):
        """match(seq : Array(Char) | Slice(Char), sep) restated from src/aha/ac.cr:342-364: the neighbour tests look
        at the neighbouring CHAR's code point (`chr.ord < sep.size && !sep[chr.ord]`), not at a byte.
        sep = (size, set_bits).  Returns char-offset hits."""
        size, bits = sep
        bits = set(bits)

        def blocked(cp):
            return cp < size and cp not in bits

        enc = [c.encode("utf-8") for c in chars]
        t = b"".join(enc)
        cmap = []
        for ci, e in enumerate(enc):
            cmap += [ci] * len(e)
        out = []
        s = 0
        for i, b in enumerate(t):
            s = self._step(s, b)
            if self.key_of[s] < 0:
                continue
            chr_idx = cmap[i]
            if chr_idx + 1 < len(chars) and blocked(ord(chars[chr_idx + 1])):
                continue
            u = s
            while True:
                k = self.key_of[u]
                st = i + 1 - len(self.keys[k])
                if not (st > 0 and blocked(ord(chars[cmap[st] - 1]))):
                    out.append((cmap[st], cmap[i] + 1, k))
                u = self.fail[u]
                if self.key_of[u] < 0:
                    break
        return out

    def textbook(self, text):
        """All true occurrences (for showing the reference emits a subset)."""
        t = _b(text)
        out = []
        for i in range(len(t)):
            for k, key in enumerate(self.keys):
                if i + 1 >= len(key) and t[i + 1 - len(key):i + 1] == key:
                    out.append((i + 1 - len(key), i + 1, k))
        return out
