"""ctypes loader for the CPU oracle (oracle/libaha_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package ``aha_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HIT_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")])

E_EMPTY_KEY, E_ZERO_BYTE, E_DUP_KEY, E_SEP_SIZE = -2, -3, -4, -5


def build(force=False):
    if os.environ.get("AHA_ORACLE_LIB"):  # another build of the same source (the sanitizer build of the CPU suite)
        return os.environ["AHA_ORACLE_LIB"]
    so = os.path.join(_HERE, "libaha_oracle.so")
    src = os.path.join(_HERE, "aha_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libaha_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp, i32, i64, u32, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
    sig = {
        "orc_strerror": (C.c_char_p, [C.c_int]),
        "orc_cedar_new": (vp, []),
        "orc_cedar_free": (None, [vp]),
        "orc_cedar_insert": (i32, [vp, vp, i32]),
        "orc_cedar_get": (i32, [vp, vp, i32]),
        "orc_cedar_delete": (i32, [vp, vp, i32]),
        "orc_cedar_key": (i32, [vp, i32, vp, i32]),
        "orc_cedar_array_size": (i32, [vp]),
        "orc_cedar_key_num": (i32, [vp]),
        "orc_cedar_leaf_size": (i32, [vp]),
        "orc_ac_compile_keys": (vp, [vp, vp, u32, C.POINTER(C.c_int), C.POINTER(u32)]),
        "orc_ac_compile_cedar": (vp, [vp]),
        "orc_ac_free": (None, [vp]),
        "orc_ac_slots": (i32, [vp]),
        "orc_ac_keys": (i32, [vp]),
        "orc_ac_max_key_len": (u32, [vp]),
        "orc_ac_key": (i32, [vp, i32, vp, i32]),
        "orc_ac_id": (i32, [vp, vp, i32]),
        "orc_ac_match": (i64, [vp, vp, i64, C.c_int, vp, i32, vp, i64]),
        "orc_ac_match_longest": (i64, [vp, vp, i64, C.c_int, C.c_int, vp, i64]),
        "orc_ac_stale_ends": (i32, [vp]),
        "orc_ac_stale_paths": (i64, [vp, vp, i64]),
        "orc_ac_match_batch": (i64, [vp, vp, vp, u64, C.c_int, vp, i64, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


class OracleError(Exception):
    def __init__(self, code, key_index=None):
        self.code = code
        self.key_index = key_index
        super().__init__(lib().orc_strerror(code).decode())


def _b(x):
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


def pack_keys(keys):
    ks = [_b(k) for k in keys]
    offs = np.zeros(len(ks) + 1, dtype=np.uint64)
    if ks:
        offs[1:] = np.cumsum([len(k) for k in ks], dtype=np.uint64)
    blob = np.frombuffer(b"".join(ks), dtype=np.uint8) if ks else np.zeros(0, np.uint8)
    return np.ascontiguousarray(blob), offs


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _sep_arg(sep):
    """sep: None, or (size, iterable of set bit indexes) mirroring BitArray.new(size)."""
    if sep is None:
        return None, 0
    size, bits = sep
    buf = np.zeros(max(32, (size + 7) // 8), dtype=np.uint8)
    for b in bits:
        buf[b >> 3] |= 1 << (b & 7)
    return buf, size


class Cedar:
    """Aha::Cedar (src/aha/cedar.cr) restated."""

    def __init__(self):
        self._h = lib().orc_cedar_new()
        self._owned = True

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().orc_cedar_free(self._h)

    def insert(self, key):
        k = _b(key)
        r = lib().orc_cedar_insert(self._h, k, len(k))
        if r < 0:
            raise OracleError(r)
        return r

    def get(self, key):
        k = _b(key)
        r = lib().orc_cedar_get(self._h, k, len(k))
        return None if r < 0 else r

    def delete(self, key):
        k = _b(key)
        return lib().orc_cedar_delete(self._h, k, len(k))

    def key(self, sid):
        buf = C.create_string_buffer(1 << 16)
        n = lib().orc_cedar_key(self._h, sid, buf, len(buf))
        if n < 0:
            raise OracleError(n)
        return buf.raw[:n]

    @property
    def array_size(self):
        return lib().orc_cedar_array_size(self._h)

    @property
    def key_num(self):
        return lib().orc_cedar_key_num(self._h)


class AC:
    """Aha::AC (src/aha/ac.cr) restated."""

    def __init__(self, h):
        self._h = h

    def __del__(self):
        if self._h:
            try:
                lib().orc_ac_free(self._h)
            except Exception:
                pass

    @classmethod
    def compile(cls, keys):
        if isinstance(keys, Cedar):
            keys._owned = False
            return cls(lib().orc_ac_compile_cedar(keys._h))
        blob, offs = pack_keys(keys)
        return cls.compile_packed(blob, offs)

    @classmethod
    def compile_packed(cls, blob, offs):
        err = C.c_int(0)
        ek = C.c_uint32(0)
        h = lib().orc_ac_compile_keys(_ptr(blob), _ptr(offs), len(offs) - 1, C.byref(err), C.byref(ek))
        if not h:
            raise OracleError(err.value, ek.value)
        return cls(h)

    @property
    def slots(self):
        return lib().orc_ac_slots(self._h)

    @property
    def n_keys(self):
        return lib().orc_ac_keys(self._h)

    @property
    def max_key_len(self):
        return lib().orc_ac_max_key_len(self._h)

    def key(self, sid):
        buf = C.create_string_buffer(1 << 16)
        n = lib().orc_ac_key(self._h, sid, buf, len(buf))
        if n < 0:
            raise OracleError(n)
        return buf.raw[:n]

    def match(self, text, chars=None, sep=None):
        """Returns a HIT_DTYPE array.  str input defaults to char offsets (the
        String overload, matcher.cr:34-39); bytes input to byte offsets."""
        if chars is None:
            chars = isinstance(text, str)
        t = np.frombuffer(_b(text), dtype=np.uint8)
        sb, ss = _sep_arg(sep)
        cap = 1024
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n = lib().orc_ac_match(self._h, _ptr(t), t.size, int(chars), _ptr(sb), ss, _ptr(out), cap)
            if n < 0:
                raise OracleError(int(n))
            if n <= cap:
                return out[:n]
            cap = int(n)

    def stale_ends(self):
        """Nodes with a stale END flag (cedar.cr:642-648): only match_longest can observe them."""
        return int(lib().orc_ac_stale_ends(self._h))

    def stale_paths(self):
        """The stale END nodes as the byte strings that lead to them (a set)."""
        need = int(lib().orc_ac_stale_paths(self._h, None, 0))
        buf = np.zeros(max(need, 1), dtype=np.uint8)
        lib().orc_ac_stale_paths(self._h, _ptr(buf), need)
        raw, out, i = buf.tobytes(), set(), 0
        while i < need:
            n = int.from_bytes(raw[i:i + 4], "little")
            out.add(raw[i + 4:i + 4 + n])
            i += 4 + n
        return out

    def match_longest(self, text, intersectable=False, chars=None):
        if chars is None:
            chars = isinstance(text, str)
        t = np.frombuffer(_b(text), dtype=np.uint8)
        cap = max(16, t.size)
        out = np.zeros(cap, dtype=HIT_DTYPE)
        n = lib().orc_ac_match_longest(self._h, _ptr(t), t.size, int(intersectable), int(chars), _ptr(out), cap)
        return out[:n]

    def match_batch(self, corpus, doc_offsets, chars=False, cap=None):
        if isinstance(corpus, (bytes, bytearray)):
            corpus = np.frombuffer(bytes(corpus), dtype=np.uint8)
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        D = doc_offsets.size - 1
        dho = np.zeros(D + 1, dtype=np.uint64)
        if cap is None:
            cap = max(1024, corpus.size // 8)
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n = lib().orc_ac_match_batch(self._h, _ptr(corpus), _ptr(doc_offsets), D, int(chars),
                                         _ptr(out), cap, _ptr(dho))
            if n <= cap:
                return out[:n], dho
            cap = int(n)
