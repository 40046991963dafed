"""Host-side mirror of the reference's public API for the accelerated path:
``Aha::AC.compile`` / ``#match`` / ``Aha::Hit`` (src/aha/ac.cr:62-112, 280-295,
321-364; src/aha/matcher.cr:2-46), backed by libaha_hip.so through the C ABI.

Names, argument meaning and error behaviour follow the reference so that the
parity tests read like spec/ac_spec.cr:

    matcher = AC.compile(["我", "我是", "是中"])
    for hit in matcher.match("我是中国人"):
        hit.end, hit.value

``str`` input is the ``String`` overload (char offsets), ``bytes`` the
``Bytes`` overload (byte offsets), a list of 1-char strings the ``Array(Char)``
overload.  There is no CPU fallback: matching needs a HIP device.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native as N

#: Aha::Hit -- src/aha/matcher.cr:2-11
Hit = namedtuple("Hit", ["start", "end", "value"])

HIT_DTYPE = np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")])


class AhaError(RuntimeError):
    """The reference raises plain Strings; ``code`` is the C-ABI status."""

    def __init__(self, code, message=None, key_index=None):
        self.code = code
        self.key_index = key_index
        super().__init__(message or N.lib().aha_strerror(code).decode())


class BitArray:
    """Minimal stand-in for Crystal's BitArray as used by match(seq, sep)."""

    def __init__(self, size):
        self.size = int(size)
        self._bits = bytearray((max(self.size, 1) + 7) // 8)

    def __setitem__(self, i, v):
        if not 0 <= i < self.size:
            raise IndexError(i)
        if v:
            self._bits[i >> 3] |= 1 << (i & 7)
        else:
            self._bits[i >> 3] &= ~(1 << (i & 7))

    def __getitem__(self, i):
        if not 0 <= i < self.size:
            raise IndexError(i)
        return bool((self._bits[i >> 3] >> (i & 7)) & 1)


def _b(x):
    if isinstance(x, str):
        return x.encode("utf-8")
    return bytes(x)


def _pack_keys(keys):
    ks = [_b(k) for k in keys]
    offs = np.zeros(len(ks) + 1, dtype=np.uint64)
    if ks:
        offs[1:] = np.cumsum([len(k) for k in ks], dtype=np.uint64)
    blob = np.frombuffer(b"".join(ks), dtype=np.uint8) if ks else np.zeros(0, np.uint8)
    return blob, offs


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _params(chars, sep, longest=0):
    p = N.aha_match_params()
    p.struct_size = C.sizeof(N.aha_match_params)
    p.char_offsets = 1 if chars else 0
    p.longest = int(longest)
    p.sep_size = 0
    if sep is not None:
        p.sep_size = sep.size
        n = min(len(sep._bits), 32)
        for i in range(n):
            p.sep_bits[i] = sep._bits[i]
    return p


class AC:
    """Aha::AC (= ACX(Int32), src/aha/ac.cr:8-11) on the MI355X."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().aha_ac_free(h)
            except Exception:  # interpreter shutdown: the module globals may already be gone
                pass

    # -- Aha::AC.compile(keys) src/aha/ac.cr:62-69 ---------------------------
    @classmethod
    def compile(cls, keys, device=-1, host_only=False, force_wide=False):
        blob, offs = _pack_keys(keys)
        return cls.compile_packed(blob, offs, device, host_only, force_wide)

    @classmethod
    def compile_packed(cls, blob, offs, device=-1, host_only=False, force_wide=False):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        opts = N.aha_options()
        opts.struct_size = C.sizeof(N.aha_options)
        opts.device = device
        opts.flags = (N.AHA_OPT_HOST_ONLY if host_only else 0) | (N.AHA_OPT_FORCE_WIDE if force_wide else 0)
        h = C.c_void_p()
        ek = C.c_uint32(0)
        rc = N.lib().aha_ac_compile(_ptr(blob), _ptr(offs), len(offs) - 1, C.byref(opts), C.byref(h),
                                    C.byref(ek))
        if rc != N.AHA_OK:
            msg = None
            if rc == N.AHA_E_DUP_KEY:  # raise "key:#{key} appear twice." ac.cr:66
                k = bytes(blob[int(offs[ek.value]):int(offs[ek.value + 1])])
                msg = f"key:{k.decode('utf-8', 'replace')} appear twice."
            raise AhaError(rc, msg, ek.value)
        return cls(h)

    # -- AC#save(io) / AC.load(io): src/aha/ac.cr:45-60 (own container, see include/aha_hip.h) --
    def to_bytes(self):
        n = N.lib().aha_ac_save(self._h, None, 0)
        if n < 0:
            raise AhaError(int(n))
        buf = np.zeros(int(n), dtype=np.uint8)
        got = N.lib().aha_ac_save(self._h, _ptr(buf), int(n))
        assert got == n
        return buf.tobytes()

    def save(self, io):
        """io: a path or a binary file object."""
        data = self.to_bytes()
        if hasattr(io, "write"):
            io.write(data)
        else:
            with open(io, "wb") as f:
                f.write(data)

    @classmethod
    def from_bytes(cls, data, device=-1, host_only=False, force_wide=False):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        opts = N.aha_options()
        opts.struct_size = C.sizeof(N.aha_options)
        opts.device = device
        opts.flags = (N.AHA_OPT_HOST_ONLY if host_only else 0) | (N.AHA_OPT_FORCE_WIDE if force_wide else 0)
        h = C.c_void_p()
        rc = N.lib().aha_ac_load(_ptr(buf), buf.size, C.byref(opts), C.byref(h))
        if rc != N.AHA_OK:
            raise AhaError(rc, "not an aha_hip automaton file" if rc == N.AHA_E_INVALID else None)
        return cls(h)

    @classmethod
    def load(cls, io, **kw):
        if hasattr(io, "read"):
            return cls.from_bytes(io.read(), **kw)
        with open(io, "rb") as f:
            return cls.from_bytes(f.read(), **kw)

    def _check(self, rc):
        if rc != N.AHA_OK:
            msg = N.lib().aha_last_error(self._h).decode() or None
            raise AhaError(rc, msg)

    @property
    def info(self):
        i = N.aha_ac_info_t()
        i.struct_size = C.sizeof(i)  # (in: the bytes this binding's struct has; the library fills no more)
        self._check(N.lib().aha_ac_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_ if f != "struct_size"}

    # -- delegate :[] src/aha/ac.cr:41-43 ------------------------------------
    def __getitem__(self, x):
        if isinstance(x, (int, np.integer)):
            buf = C.create_string_buffer(1 << 12)
            n = N.lib().aha_ac_key(self._h, int(x), buf, len(buf))
            if n > len(buf):
                buf = C.create_string_buffer(n)
                n = N.lib().aha_ac_key(self._h, int(x), buf, len(buf))
            if n < 0:
                raise IndexError(x)
            return buf.raw[:n].decode("utf-8", "replace")
        k = _b(x)
        r = N.lib().aha_ac_id(self._h, k, len(k))
        if r < 0:
            raise IndexError(x)  # IndexError.new cedar.cr:832
        return r

    # -- #match ---------------------------------------------------------------
    def match_array(self, seq, sep=None, chars=None, longest=0):
        """All hits of one sequence as a HIT_DTYPE array (reference order)."""
        if isinstance(seq, (list, tuple)) and (not seq or isinstance(seq[0], str)):
            if sep is not None:
                return self._match_chars_sep(list(seq), sep)
            # Array(Char) overload (ac.cr:288-295): chars re-encode to UTF-8
            seq = "".join(seq)
        if chars is None:
            chars = isinstance(seq, str)
        t = np.frombuffer(_b(seq), dtype=np.uint8)
        p = _params(chars, sep, longest)
        cap = max(64, t.size // 4)
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n = C.c_uint64(0)
            rc = N.lib().aha_ac_match_bytes(self._h, _ptr(t), t.size, C.byref(p), _ptr(out), cap, C.byref(n))
            if rc == N.AHA_E_CAPACITY:
                cap = int(n.value)
                continue
            self._check(rc)
            return out[: n.value]

    def _match_chars_sep(self, chars, sep):
        """match(seq : Array(Char) | Slice(Char), sep) -- src/aha/ac.cr:342-364.  Unlike the String overload
        (matcher.cr:41-46, byte-level neighbours) this one tests the neighbouring CODE POINT: a hit is dropped when
        `chr.ord < sep.size && !sep[chr.ord]`, so a neighbour whose code point is >= sep.size never blocks.  The GPU
        yields the unfiltered byte-offset hits; the two neighbour tests and the char offsets are applied here."""
        if sep.size > 256:
            raise AhaError(N.AHA_E_SEP_SIZE, "sep BitArray size > 256 is not supported")
        enc = [c.encode("utf-8") for c in chars]
        t = b"".join(enc)
        hits = self.match_array(t, None, False)
        if not len(hits):
            return hits
        cps = np.array([ord(c) for c in chars], dtype=np.int64)
        blens = np.array([len(e) for e in enc], dtype=np.int64)
        char_of_byte = np.repeat(np.arange(len(chars), dtype=np.int64), blens)
        ok_cp = np.array([not (i < sep.size and not sep[i]) for i in range(256)], dtype=bool)  # blocked(c) for c < 256
        blocked = (cps < 256) & ~ok_cp[np.minimum(cps, 255)]
        nchars = len(chars)
        end_chr = char_of_byte[hits["end"].astype(np.int64) - 1]      # char that holds the last byte (ac.cr:346)
        right = end_chr + 1
        keep = ~((right < nchars) & blocked[np.minimum(right, nchars - 1)])
        start_chr = char_of_byte[hits["start"].astype(np.int64)]
        has_left = hits["start"] > 0                                    # ac.cr:354
        keep &= ~(has_left & blocked[np.maximum(start_chr - 1, 0)])
        out = np.zeros(int(keep.sum()), dtype=HIT_DTYPE)
        out["start"] = start_chr[keep]
        out["end"] = end_chr[keep] + 1
        out["value"] = hits["value"][keep]
        return out

    def match(self, seq, sep=None, chars=None):
        """Yields Aha::Hit like the reference's block form (ac.cr:280-286)."""
        for s, e, v in self.match_array(seq, sep, chars).tolist():
            yield Hit(s, e, v)

    def match_longest(self, seq, intersectable=False, chars=None):
        """ACX#match_longest(seq, intersectable = false) -- src/aha/ac.cr:297-319."""
        for s, e, v in self.match_array(seq, None, chars, longest=2 if intersectable else 1).tolist():
            yield Hit(s, e, v)

    def match_batch(self, corpus, doc_offsets, sep=None, chars=False, cap=None, longest=0):
        """D documents in one call: returns (hits, doc_hit_offsets).  longest: 0 = #match, 1 / 2 = #match_longest
        with intersectable false / true."""
        if isinstance(corpus, (bytes, bytearray)):
            corpus = np.frombuffer(bytes(corpus), dtype=np.uint8)
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        D = doc_offsets.size - 1
        p = _params(chars, sep, longest)
        dho = np.zeros(D + 1, dtype=np.uint64)
        if cap is None:
            cap = max(64, corpus.size // 8)
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n = C.c_uint64(0)
            rc = N.lib().aha_ac_match_batch(self._h, _ptr(corpus), _ptr(doc_offsets), D, C.byref(p), _ptr(out),
                                            cap, _ptr(dho), C.byref(n))
            if rc == N.AHA_E_CAPACITY:
                cap = int(n.value)
                continue
            self._check(rc)
            return out[: n.value], dho

    def match_batch_device(self, corpus, doc_offsets, out, doc_hit_offsets=None, sep=None, chars=False,
                           stream=None, words=None, n_words=None):
        """Device-resident batch match on torch CUDA tensors (uint8 corpus,
        int64/uint64 doc offsets, int32 [cap,3] out).  Returns the hit count;
        raises AhaError(AHA_E_CAPACITY) with .required when out is too small.
        words (int32, >= 2 cap + cap / 1024 + 2 elements) + n_words (int64[1]): the hits also as the 4-byte exchange stream
        (aha_ac_match_batch_device_stream: written by the expansion itself where the character-level engine runs)."""
        import torch

        assert corpus.is_cuda and corpus.dtype == torch.uint8 and corpus.is_contiguous()
        assert doc_offsets.is_cuda and doc_offsets.dtype in (torch.int64, torch.uint64)
        assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous()
        D = doc_offsets.numel() - 1
        cap = out.numel() // 3
        p = _params(chars, sep)
        n = C.c_uint64(0)
        s = stream if stream is not None else torch.cuda.current_stream(corpus.device).cuda_stream
        dho = doc_hit_offsets.data_ptr() if doc_hit_offsets is not None else None
        if words is not None:
            assert words.is_cuda and words.dtype == torch.int32 and n_words.is_cuda and n_words.dtype == torch.int64
            rc = N.lib().aha_ac_match_batch_device_stream(self._h, corpus.data_ptr(), doc_offsets.data_ptr(), D, corpus.numel(),
                                                          C.byref(p), out.data_ptr(), cap, dho, C.byref(n), words.data_ptr(),
                                                          words.numel(), n_words.data_ptr(), C.c_void_p(s))
        else:
            rc = N.lib().aha_ac_match_batch_device(self._h, corpus.data_ptr(), doc_offsets.data_ptr(), D,
                                                   corpus.numel(), C.byref(p), out.data_ptr(), cap, dho,
                                                   C.byref(n), C.c_void_p(s))
        if rc == N.AHA_E_CAPACITY:
            e = AhaError(rc)
            e.required = int(n.value)
            raise e
        self._check(rc)
        return int(n.value)

    def match_batch_keep(self, corpus, doc_offsets, d_hits, chars=False):
        """aha_ac_match_batch_keep: host corpus in (uploaded range by range beside the matches), the hits stay on the device in
        d_hits (a torch int32 [cap, 3] tensor on the handle's device); returns (n_hits, per-document offsets)."""
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        D = doc_offsets.size - 1
        p = _params(chars, None)
        dho = np.zeros(D + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = N.lib().aha_ac_match_batch_keep(self._h, _ptr(corpus), _ptr(doc_offsets), D, C.byref(p),
                                             C.c_void_p(d_hits.data_ptr()), d_hits.shape[0], _ptr(dho), C.byref(n))
        if rc == N.AHA_E_CAPACITY:
            e = AhaError(rc)
            e.required = int(n.value)
            raise e
        self._check(rc)
        return int(n.value), dho

    def replicate(self, device):
        """aha_ac_replicate: a second handle for the same keys on another device -- the host image is copied and uploaded,
        nothing is compiled again."""
        h = C.c_void_p()
        rc = N.lib().aha_ac_replicate(self._h, device, C.byref(h))
        if rc != N.AHA_OK:
            raise AhaError(rc)
        return AC(h)

    def match_corpus(self, corpus, cap=None, sep=None, chars=False, longest=0):
        """Matches a batch that already lives in HBM (DeviceCorpus: uploaded once through the C ABI, no GPU framework
        involved) and downloads the hits: -> (hits, doc_hit_offsets).  The upload is not repeated per call."""
        D = corpus.n_docs
        if cap is None:
            cap = max(64, corpus.n_bytes // 8)
        p = _params(chars, sep, longest)
        dev = corpus.device
        dho = DeviceBuffer(dev, (D + 1) * 8)
        while True:
            out = DeviceBuffer(dev, cap * 12)
            n = C.c_uint64(0)
            rc = N.lib().aha_ac_match_batch_device(self._h, corpus.ptr, corpus.doc_ptr, D, corpus.n_bytes, C.byref(p),
                                                   out.ptr, cap, dho.ptr, C.byref(n), None)
            if rc == N.AHA_E_CAPACITY:
                cap = int(n.value)
                continue
            self._check(rc)
            return out.download(np.zeros(int(n.value), dtype=HIT_DTYPE)), dho.download(np.zeros(D + 1, dtype=np.uint64))

    # -- exchange format of the multi-GPU all-gatherv: {end, value} pairs <-> Hit triples ------------
    def hits_pack_device(self, hits, n, pairs, stream=None):
        """hits [>=n,3] int32 -> pairs [>=n,2] int32, both on the handle's device (asynchronous)."""
        import torch

        assert hits.is_cuda and pairs.is_cuda and hits.dtype == pairs.dtype == torch.int32
        assert hits.is_contiguous() and pairs.is_contiguous() and hits.numel() >= 3 * n and pairs.numel() >= 2 * n
        s = stream if stream is not None else torch.cuda.current_stream(hits.device).cuda_stream
        self._check(N.lib().aha_ac_hits_pack_device(self._h, hits.data_ptr(), n, pairs.data_ptr(), C.c_void_p(s)))

    def hits_unpack_device(self, pairs, n, hits, chars=False, stream=None):
        """pairs [>=n,2] int32 -> hits [>=n,3] int32 (start = end - key length), asynchronous."""
        import torch

        assert hits.is_cuda and pairs.is_cuda and hits.dtype == pairs.dtype == torch.int32
        assert hits.is_contiguous() and pairs.is_contiguous() and hits.numel() >= 3 * n and pairs.numel() >= 2 * n
        s = stream if stream is not None else torch.cuda.current_stream(hits.device).cuda_stream
        self._check(N.lib().aha_ac_hits_unpack_device(self._h, pairs.data_ptr(), n, 1 if chars else 0,
                                                      hits.data_ptr(), C.c_void_p(s)))

    def stream_format(self):
        """(step_bits, len_bits) of the 4-byte exchange stream's words for this automaton (include/aha_hip.h)."""
        sb, lb = C.c_uint32(0), C.c_uint32(0)
        self._check(N.lib().aha_ac_stream_format(self._h, C.byref(sb), C.byref(lb)))
        return int(sb.value), int(lb.value)

    def hits_pack4_device(self, hits, n, words, n_words, stream=None):
        """hits [>=n,3] int32 -> the 4-byte exchange stream in `words` (int32, capacity >= 2n + ceil(n/1024)); the
        stream length lands in n_words[0] (int64 device tensor).  Asynchronous."""
        import torch

        assert hits.is_cuda and words.is_cuda and n_words.is_cuda and hits.dtype == words.dtype == torch.int32
        assert n_words.dtype == torch.int64 and hits.is_contiguous() and words.is_contiguous() and hits.numel() >= 3 * n
        s = stream if stream is not None else torch.cuda.current_stream(hits.device).cuda_stream
        self._check(N.lib().aha_ac_hits_pack4_device(self._h, hits.data_ptr(), n, words.data_ptr(), words.numel(),
                                                     n_words.data_ptr(), C.c_void_p(s)))

    def hits_unpack4_device(self, words, n, hits, chars=False, stream=None):
        """4-byte exchange stream of n hits -> hits [>=n,3] int32, asynchronous."""
        import torch

        assert hits.is_cuda and words.is_cuda and hits.dtype == words.dtype == torch.int32
        assert hits.is_contiguous() and words.is_contiguous() and hits.numel() >= 3 * n
        s = stream if stream is not None else torch.cuda.current_stream(hits.device).cuda_stream
        self._check(N.lib().aha_ac_hits_unpack4_device(self._h, words.data_ptr(), n, 1 if chars else 0,
                                                       hits.data_ptr(), C.c_void_p(s)))

    def hits_unpack4_segs_device(self, words, segs, hits, chars=False, stream=None):
        """Several 4-byte streams inside `words` -> their places in `hits`, ONE launch: segs = [(word offset of the
        stream, its hit count, row offset in hits), ...] (at most 64)."""
        import torch

        assert hits.is_cuda and words.is_cuda and hits.dtype == words.dtype == torch.int32
        assert hits.is_contiguous() and words.is_contiguous()
        arr = (N.aha_stream_seg * max(len(segs), 1))()
        for k, (wo, n, oo) in enumerate(segs):
            assert hits.numel() >= 3 * (oo + n)
            arr[k].word_offset, arr[k].n_hits, arr[k].out_offset = int(wo), int(n), int(oo)
        s = stream if stream is not None else torch.cuda.current_stream(hits.device).cuda_stream
        self._check(N.lib().aha_ac_hits_unpack4_segs_device(self._h, words.data_ptr(), arr, len(segs),
                                                            1 if chars else 0, hits.data_ptr(), C.c_void_p(s)))

    @property
    def n_keys(self):
        return self.info["n_keys"]

    def key_lengths(self, chars=False):
        """Length of every key in bytes (or in chars): Hit#end - Hit#start of its hits."""
        ln = self.export(N.AHA_IMG_KEY_LN, np.uint32).reshape(-1, 2)[:, 0].astype(np.int32)
        if chars:
            return self.export(N.AHA_IMG_KEY_KC, np.uint32).astype(np.int32) + 1
        return ln

    def export(self, which, dtype):
        """One array of the automaton image (data; host-logic tests, debugging)."""
        n = N.lib().aha_ac_export(self._h, which, None, 0)
        if n < 0:
            raise AhaError(int(n))
        buf = np.zeros(int(n) // np.dtype(dtype).itemsize, dtype=dtype)
        if n:
            N.lib().aha_ac_export(self._h, which, _ptr(buf), int(n))
        return buf

    def set_profiling(self, enabled=True):
        self._check(N.lib().aha_ac_set_profiling(self._h, 1 if enabled else 0))

    def release_scratch(self):
        """Frees the handle's grow-only device scratch."""
        self._check(N.lib().aha_ac_release_scratch(self._h))

    def scratch_bytes(self):
        """Device bytes the handle currently holds as scratch (all scratch sets)."""
        return int(N.lib().aha_ac_scratch_bytes(self._h))

    def last_timing(self):
        t = N.aha_timing()
        t.struct_size = C.sizeof(t)
        self._check(N.lib().aha_ac_last_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in t._fields_ if f != "struct_size"}


# Aha::ACBig = ACX(Int64) (src/aha/ac.cr:9): node ids of 64 bits, the same Hit with an Int32 value (ac.cr:273) -- the
# library's numbering has no such limit below 2^31 keys, so one class answers for both names.
ACBig = AC


class DeviceBuffer:
    """HBM obtained through the C ABI (aha_buffer_alloc): what a caller without a GPU framework uses."""

    def __init__(self, device, n_bytes):
        self.device, self.n_bytes = device, int(n_bytes)
        p = C.c_void_p()
        rc = N.lib().aha_buffer_alloc(device, self.n_bytes, C.byref(p))
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_last_error(None).decode() or None)
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.n_bytes
        rc = N.lib().aha_buffer_upload(self.device, self.ptr, _ptr(arr), arr.nbytes)
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_last_error(None).decode() or None)

    def download(self, arr):
        assert arr.flags["C_CONTIGUOUS"] and arr.nbytes <= self.n_bytes
        rc = N.lib().aha_buffer_download(self.device, _ptr(arr), self.ptr, arr.nbytes)
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_last_error(None).decode() or None)
        return arr

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            try:
                N.lib().aha_buffer_free(self.device, p)
            except Exception:  # interpreter shutdown
                pass


class DeviceCorpus:
    """A batch resident in HBM (aha_corpus_upload): bytes of all documents + D + 1 offsets, validated on upload."""

    def __init__(self, corpus, doc_offsets, device=0):
        if isinstance(corpus, (bytes, bytearray)):
            corpus = np.frombuffer(bytes(corpus), dtype=np.uint8)
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        h = C.c_void_p()
        rc = N.lib().aha_corpus_upload(device, _ptr(corpus), _ptr(doc_offsets), doc_offsets.size - 1, C.byref(h))
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_last_error(None).decode() or None)
        self._h = h
        self.device = device
        self.n_docs = int(N.lib().aha_corpus_n_docs(h))
        self.n_bytes = int(N.lib().aha_corpus_n_bytes(h))
        self.ptr = C.c_void_p(N.lib().aha_corpus_bytes(h))
        self.doc_ptr = C.c_void_p(N.lib().aha_corpus_doc_offsets(h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().aha_corpus_free(h)
            except Exception:  # interpreter shutdown
                pass


class ACGroup:
    """Several GPUs of one node behind one object (aha_group_*, include/aha_hip.h): contiguous byte-balanced
    document ranges, one per device entry, all-gatherv of the hit buffers (RCCL between distinct devices)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().aha_group_free(h)
            except Exception:  # interpreter shutdown
                pass

    @classmethod
    def compile(cls, keys, devices, host_only=False):
        blob, offs = _pack_keys(keys)
        return cls.compile_packed(blob, offs, devices, host_only)

    @classmethod
    def compile_packed(cls, blob, offs, devices, host_only=False):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        ek = C.c_uint32(0)
        rc = N.lib().aha_group_compile(_ptr(blob), _ptr(offs), offs.size - 1, _ptr(dev), dev.size,
                                       N.AHA_OPT_HOST_ONLY if host_only else 0, C.byref(h), C.byref(ek))
        if rc != N.AHA_OK:
            raise AhaError(rc, key_index=ek.value)
        return cls(h)

    @staticmethod
    def partition(doc_offsets, n_parts):
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        bounds = np.zeros(n_parts + 1, dtype=np.uint64)
        rc = N.lib().aha_group_partition(_ptr(doc_offsets), doc_offsets.size - 1, n_parts, _ptr(bounds))
        if rc != N.AHA_OK:
            raise AhaError(rc)
        return bounds

    def match_batch(self, corpus, doc_offsets, chars=False, cap=None):
        if isinstance(corpus, (bytes, bytearray)):
            corpus = np.frombuffer(bytes(corpus), dtype=np.uint8)
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        D = doc_offsets.size - 1
        p = _params(chars, None)
        dho = np.zeros(D + 1, dtype=np.uint64)
        if cap is None:
            cap = max(64, corpus.size // 8)
        while True:
            out = np.zeros(cap, dtype=HIT_DTYPE)
            n = C.c_uint64(0)
            rc = N.lib().aha_group_match_batch(self._h, _ptr(corpus), _ptr(doc_offsets), D, C.byref(p), _ptr(out), cap,
                                               _ptr(dho), C.byref(n))
            if rc == N.AHA_E_CAPACITY:
                cap = int(n.value)
                continue
            if rc != N.AHA_OK:
                raise AhaError(rc, N.lib().aha_group_last_error(self._h).decode() or None)
            return out[: n.value], dho

    def upload_corpus(self, corpus, doc_offsets):
        """The batch resident on the group's devices (aha_group_corpus_upload): every shard's document range on its device."""
        if isinstance(corpus, (bytes, bytearray)):
            corpus = np.frombuffer(bytes(corpus), dtype=np.uint8)
        corpus = np.ascontiguousarray(corpus, dtype=np.uint8)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        h = C.c_void_p()
        rc = N.lib().aha_group_corpus_upload(self._h, _ptr(corpus), _ptr(doc_offsets), doc_offsets.size - 1, C.byref(h))
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_group_last_error(self._h).decode() or None)
        return GroupCorpus(self, h, doc_offsets.size - 1)

    def match_corpus(self, gcorpus, chars=False):
        """aha_group_match_batch_device: every device matches its resident range, then the all-gatherv; the hits stay on the
        devices (download_shard reads one device's copy of the whole stream).  Returns (n_hits, per-document offsets)."""
        p = _params(chars, None)
        dho = np.zeros(gcorpus.n_docs + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = N.lib().aha_group_match_batch_device(self._h, gcorpus._h, C.byref(p), _ptr(dho), C.byref(n))
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_group_last_error(self._h).decode() or None)
        return int(n.value), dho

    def download_shard(self, shard):
        """The gathered hit stream as device `shard` holds it after the last match_batch (every device holds it all)."""
        n = C.c_uint64(0)
        N.lib().aha_group_download_shard(self._h, shard, None, 0, C.byref(n))
        out = np.zeros(max(int(n.value), 1), dtype=HIT_DTYPE)
        rc = N.lib().aha_group_download_shard(self._h, shard, _ptr(out), out.size, C.byref(n))
        if rc != N.AHA_OK:
            raise AhaError(rc, N.lib().aha_group_last_error(self._h).decode() or None)
        return out[: n.value]

    def last_timing(self):
        t = N.aha_group_timing()
        rc = N.lib().aha_group_last_timing(self._h, C.byref(t))
        if rc != N.AHA_OK:
            raise AhaError(rc)
        return {f: getattr(t, f) for f, _ in t._fields_ if f != "struct_size"}


class GroupCorpus:
    """A batch resident on the devices of an ACGroup (aha_group_corpus_*); keeps its group alive."""

    def __init__(self, group, handle, n_docs):
        self._g, self._h, self.n_docs = group, handle, n_docs

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                N.lib().aha_group_corpus_free(h)
            except Exception:  # interpreter shutdown
                pass
