"""GPU parity tests: the HIP path (through the C ABI of libaha_hip.so) against
the CPU oracle on the same inputs -- bit-exact hit triples in the same order.
Reads like spec/ac_spec.cr; run with `pytest -m gpu` on an MI355X."""
import json
import os
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC, AhaError, BitArray, Hit, synth
from aha_amd import _native as N
from pymodel import ModelAC
from test_oracle_vs_model import as_list, rand_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["v2", "v1", "v2p", "u", "ur", "u23", "uh", "k", "p", "f", "auto"], autouse=True)
def engine(request, monkeypatch):
    """Every parity test runs on the single-traversal engine (scan_v2.hip, byte level), on the two-pass engine
    (kernels.hip: the fallback for tiny capacities and very long keys), on the single-traversal engine with its LDS
    prefix capped at 1024 slots ("v2p": small automata then also take the partial-prefix kernel with the shadow fail
    links and the HBM probe path) and on the character-level engine ("u", scan_unit.hip: AHA_ENGINE=unit builds the
    unit image for every eligible key set, also the mostly-ASCII ones that would not get one by default; byte-offset
    calls through the event regions then run it, everything else the single-traversal engine; its post pass is the fused
    expansion wherever the output chains are short enough, "ur" -- AHA_UNIT_POST=regroup -- keeps it to the general
    regroup + count + expand passes; "u23" --
    AHA_UNIT_BASE_BITS=23 -- builds every image in the wide format of images beyond 2^22 slots: 23-bit bases, 6-bit filter,
    3-bit hit count in the event record; "uh" -- AHA_UNIT_HEADER_BESIDE=1 -- the traversal that requests a state's fail header
    beside its probe, ku_traverse<.., HB>, which the library picks for key sets like cfg 5's; "u" pins the header trip; "f" --
    AHA_ENGINE=filter -- the prefix-filter engine, scan_filter.hip, for every key set it takes: keys of 3 to 64 bytes, byte
    offsets, no separator filter -- what the library picks for such key sets when they get no character-level image; "k" --
    AHA_ENGINE=skip -- the skip-ahead traversal, scan_skip.hip: the unit image for every eligible key set like "u", its walk
    started only at the marks of a first pass wherever no key is a single character, whatever the mark filter's fill; "p" --
    AHA_ENGINE=pair -- the pair engine, scan_pair.hip: the unit image likewise, byte-offset calls answered by the stateless pair
    pass + deep walks wherever the key set admits it).  "auto" sets
    no variable: the library decides per key set, which is what a caller and bench.py get.  The variables are read when a
    handle is compiled."""
    only = os.environ.get("AHA_TEST_ENGINES")  # (development: run the suite on some variants only, e.g. AHA_TEST_ENGINES=f,auto)
    if only and request.param not in only.split(","):
        pytest.skip("variant not selected by AHA_TEST_ENGINES")
    if request.param == "auto":
        monkeypatch.delenv("AHA_ENGINE", raising=False)
    else:
        monkeypatch.setenv("AHA_ENGINE", {"v1": "v1", "u": "unit", "ur": "unit", "u23": "unit", "uh": "unit", "f": "filter", "k": "skip", "p": "pair"}.get(request.param, "v2"))
    if request.param in ("u", "uh"):
        monkeypatch.setenv("AHA_UNIT_HEADER_BESIDE", "1" if request.param == "uh" else "0")
    else:
        monkeypatch.delenv("AHA_UNIT_HEADER_BESIDE", raising=False)
    if request.param == "u23":
        monkeypatch.setenv("AHA_UNIT_BASE_BITS", "23")
    else:
        monkeypatch.delenv("AHA_UNIT_BASE_BITS", raising=False)
    if request.param == "ur":
        monkeypatch.setenv("AHA_UNIT_POST", "regroup")
    else:
        monkeypatch.delenv("AHA_UNIT_POST", raising=False)
    if request.param == "v2p":
        monkeypatch.setenv("AHA_LDS_SLOTS", "1024")
    else:
        monkeypatch.delenv("AHA_LDS_SLOTS", raising=False)
    return request.param


G = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(G, "reference_kats.json"), encoding="utf-8"))


def gpu_list(h):
    return [tuple(int(v) for v in x) for x in h.tolist()]


def stale_paths_of(ac, keys):
    """The library's own replay of Cedar's stale END flags (AHA_IMG_STALE_ENDS), as a set of byte strings."""
    a = ac.export(N.AHA_IMG_STALE_ENDS, np.uint32).reshape(-1, 2)
    return set(bytes(keys[int(k)][:int(n)]) for k, n in a)


# ---- the reference's own specs, run against the HIP path --------------------

def test_spec_ac():  # spec/ac_spec.cr:5-12
    matcher = AC.compile(["我", "我是", "是中"])
    matched = [(hit.end, hit.value) for hit in matcher.match("我是中国人")]
    assert matched == [(1, 0), (2, 1), (3, 2)]


def test_spec_ac_with_sep():  # spec/ac_spec.cr:25-34
    matcher = AC.compile(["a", "aa"])
    sep = BitArray(256)
    sep[ord(" ")] = True
    matched = [(hit.end, hit.value) for hit in matcher.match("a aaa", sep)]
    assert matched == [(1, 0)]


def test_spec_ac_char_array():  # spec/ac_spec.cr:36-43
    matcher = AC.compile(["我", "我是", "是中"])
    matched = [(hit.end, hit.value) for hit in matcher.match(list("我是中国人"))]
    assert matched == [(1, 0), (2, 1), (3, 2)]


def test_spec_ac_save_load(tmp_path):  # spec/ac_spec.cr:14-23 -- and, unlike the spec, the LOADED automaton is matched
    matcher = AC.compile(["我", "我是", "是中"])
    matcher.save(str(tmp_path / "aha.bin"))
    loaded = AC.load(str(tmp_path / "aha.bin"))
    matched = [(hit.end, hit.value) for hit in loaded.match("我是中国人")]
    assert matched == [(1, 0), (2, 1), (3, 2)]
    rng = random.Random(5)
    keys = rand_keys(rng, 300, b"abc", 1, 7)
    a = AC.compile(keys)
    b = AC.from_bytes(a.to_bytes())
    text = bytes(rng.choice(b"abc") for _ in range(5000))
    assert gpu_list(b.match_array(text)) == gpu_list(a.match_array(text)) == as_list(orc.AC.compile(keys).match(text))


@pytest.mark.parametrize("kat", KATS["ac_match"], ids=lambda k: k["cite"][:24] + k["api"])
def test_reference_kats(kat):
    ac = AC.compile(kat["keys"])
    sep = None
    if kat["sep"]:
        sep = BitArray(kat["sep"]["size"])
        for b in kat["sep"]["set"]:
            sep[b] = True
    seq = list(kat["text"]) if kat["api"] == "chars" else kat["text"]
    assert [[h.end, h.value] for h in ac.match(seq, sep)] == kat["expect_end_value"]


def test_char_array_with_sep_tests_code_points():
    """match(Array(Char), sep) (ac.cr:342-364) looks at the neighbour's CODE POINT, the String overload
    (matcher.cr:41-46) at the neighbouring byte: they differ as soon as a neighbour is not ASCII."""
    ac = AC.compile(["a"])
    sep = BitArray(256)  # nothing is a separator
    assert [tuple(h) for h in ac.match(list("中a中"), sep)] == [(1, 2, 0)]  # U+4E2D >= sep.size: never blocks
    assert [tuple(h) for h in ac.match("中a中", sep)] == []                 # bytes 0xAD / 0xE4 < 256 and not set: blocked
    rng = random.Random(9)
    cps = [chr(c) for c in list(range(0x4E00, 0x4E10)) + list(range(97, 103)) + list(range(0x430, 0x436)) + [0xE9, 0xFF, 32]]
    keys = sorted({"".join(rng.choice(cps[:-1]) for _ in range(rng.randint(1, 3))) for _ in range(120)})
    g = AC.compile(keys)
    m = ModelAC(keys)
    for size, bits in ((256, [32, 0xE9]), (100, [32, 97]), (0xF0, [0xE9])):
        sep = BitArray(size)
        for b in bits:
            sep[b] = True
        text = [rng.choice(cps) for _ in range(3000)]
        assert [tuple(h) for h in g.match(text, sep)] == m.match_chars_sep(text, (size, bits))


# ---- match_longest (src/aha/ac.cr:118-143, 297-319; spec/ac_longest_match_spec.cr) -----------------------------

@pytest.mark.parametrize("kat", KATS["ac_match_longest"], ids=lambda k: k["cite"][-5:])
def test_reference_match_longest_kats(kat):  # spec/ac_longest_match_spec.cr:5-63 (String and Array(Char) forms)
    ac = AC.compile(kat["keys"])
    for seq in (kat["text"], list(kat["text"]), kat["text"].encode()):
        got = [[h.start, h.end, kat["keys"][h.value]] for h in ac.match_longest(seq, kat["intersectable"])]
        assert got == kat["expect"]


@pytest.mark.parametrize("seed", range(6))
def test_match_longest_random(seed):
    """Random automata and ragged batches, both forms, byte and char offsets (three device paths: a thread per
    document for intersectable = false and for char offsets, a thread per chunk with a 2 * Lmax warm-up for
    intersectable = true), against the ORACLE on every seed: the library replays Cedar's slot history
    (cedar_replay.cpp), so the stale END flags of cedar.cr:642-648 -- five of the six seeds hold some -- are
    reproduced, not exempted.  The independent model (given the oracle's stale set) must agree as well.
    The text holds NUL bytes: after a state that ends a key and has children a NUL reaches the Cedar node that keeps
    the value -- it replaces the pending end by one that yields nothing and swallows the next byte (kernels.hip)."""
    rng = random.Random(600 + seed)
    alphabet = [b"ab", b"abc", "abж中".encode(), bytes(range(0x61, 0x6B))][seed % 4]
    keys = rand_keys(rng, rng.randint(1, 80), alphabet, 1, [4, 9, 30][seed % 3])
    g = AC.compile(keys)
    m = ModelAC(keys)
    o = orc.AC.compile(keys)
    stale = o.stale_paths()
    assert stale_paths_of(g, keys) == stale
    docs = [bytes(rng.choice(alphabet + b" ") for _ in range(rng.choice([0, 1, 2, 7, 100, 1023, 1024, 1025, 5000])))
            for _ in range(40)] + [bytes(rng.choice(alphabet) for _ in range(40000))] + \
           [bytes(rng.choice(alphabet + b"\x00") for _ in range(n)) for n in (3, 50, 3000, 30000)] + \
           [bytes(0 if rng.random() < 1 / 300 else rng.choice(alphabet) for _ in range(40000))]  # a NUL now and then: the
    # chunks' warm-ups reach back past it (kernels.hip, k_longest_chunks); the dense ones above end at a double NUL or give up
    if seed == 0:  # a NUL behind every byte (UTF-16 read as bytes): the chunked form gives up, document by document
        docs.append(b"".join(bytes([rng.choice(alphabet), 0]) for _ in range(6000)))
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8)
    for inter in (False, True):
        for chars in (False, True):
            if chars:
                try:
                    [d.decode("utf-8") for d in docs]
                except UnicodeDecodeError:
                    continue  # char offsets are defined for valid UTF-8 only
            want, want_off = [], [0]
            for d in docs:
                want += as_list(o.match_longest(d, inter, chars=chars))
                want_off.append(len(want))
            gh, gd = g.match_batch(corpus, offs, chars=chars, longest=2 if inter else 1)
            assert gpu_list(gh) == want, (inter, chars, len(stale))
            assert gd.tolist() == want_off
            assert want == [t for d in docs for t in m.match_longest(d, inter, chars=chars, stale=stale)]
    # the single-sequence entry and the capacity protocol
    t = docs[-1]
    assert [tuple(h) for h in g.match_longest(t, True)] == as_list(o.match_longest(t, True))
    with pytest.raises(AhaError) as e:  # no separator overload of match_longest in the reference
        sep = BitArray(256)
        g.match_array(t, sep, longest=1)
    assert e.value.code == N.AHA_E_INVALID


def test_match_longest_config2_keys_with_stale_ends():
    """BASELINE cfg 2's own key set (1 000 keys: 121 stale END nodes in the reference's Cedar) on 512 KiB of cfg 2 text
    cut into 16 documents, both forms, against the oracle."""
    blob, offs, nf = synth.keys(2)
    corpus, doc = synth.corpus(2, blob, offs, nf, n_bytes=512 << 10, doc_bytes=32 << 10)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    keys = [bytes(blob[offs[i]:offs[i + 1]]) for i in range(offs.size - 1)]
    stale = o.stale_paths()
    assert len(stale) == 121 and stale_paths_of(g, keys) == stale
    for inter in (False, True):
        want, want_off = [], [0]
        for d in range(doc.size - 1):
            want += as_list(o.match_longest(corpus[int(doc[d]):int(doc[d + 1])].tobytes(), inter, chars=False))
            want_off.append(len(want))
        gh, gd = g.match_batch(corpus, doc, longest=2 if inter else 1)
        assert gpu_list(gh) == want, inter
        assert gd.tolist() == want_off


def test_byte_level_triples():
    ac = AC.compile(["我", "我是", "是中"])
    assert list(ac.match("我是中国人".encode())) == [Hit(0, 3, 0), Hit(0, 6, 1), Hit(3, 9, 2)]
    assert list(ac.match("我是中国人")) == [Hit(0, 1, 0), Hit(0, 2, 1), Hit(1, 3, 2)]


# ---- derived semantics, both slot formats ----------------------

@pytest.mark.parametrize("wide", [False, True])
def test_subset_semantics(wide):
    assert list(AC.compile(["c", "abcd"], force_wide=wide).match(b"abc")) == []
    assert gpu_list(AC.compile(["a", "aa"], force_wide=wide).match_array(b"aa")) == [(0, 1, 0), (0, 2, 1), (1, 2, 0)]
    assert gpu_list(AC.compile(["xabc", "abc", "bcz", "c"], force_wide=wide).match_array(b"xabc")) == [(0, 4, 0), (1, 4, 1)]


def test_nul_and_empty():
    ac = AC.compile(["ab", "abc", "b"])
    assert gpu_list(ac.match_array(b"ab\x00abc\x00b")) == [(0, 2, 0), (1, 2, 2), (3, 5, 0), (4, 5, 2), (3, 6, 1), (7, 8, 2)]
    assert len(ac.match_array(b"")) == 0
    hits, dho = ac.match_batch(b"", [0, 0, 0])
    assert len(hits) == 0 and dho.tolist() == [0, 0, 0]


def test_zero_keys_and_single_byte_docs():
    ac = AC.compile([])
    assert len(ac.match_array(b"anything at all")) == 0
    ac = AC.compile(["a"])
    docs = [b"a", b"", b"b", b"a", b"a"]
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    hits, dho = ac.match_batch(np.frombuffer(b"".join(docs), dtype=np.uint8), offs)
    assert gpu_list(hits) == [(0, 1, 0)] * 3 and dho.tolist() == [0, 1, 1, 1, 2, 3]


def test_sep_size_error():
    ac = AC.compile(["a"])
    with pytest.raises(AhaError) as e:
        list(ac.match(b"a", BitArray(257)))
    assert e.value.code == N.AHA_E_SEP_SIZE
    assert str(e.value) == "sep BitArray size > 256 is not supported"


# ---- randomized parity vs the oracle ---------------------------------------

@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("seed", range(6))
def test_random_small_alphabet(seed, wide):
    rng = random.Random(seed)
    alphabet = [b"ab", b"abc", b"abcd\xe4\xb8"][seed % 3]
    keys = rand_keys(rng, rng.randint(1, 60), alphabet, 1, 8)
    # several chunks long: matches straddle chunk boundaries all the time
    text = bytes(rng.choice(alphabet + (b"\x00" if seed % 2 else b"")) for _ in range(5000))
    g = AC.compile(keys, force_wide=wide)
    o = orc.AC.compile(keys)
    assert gpu_list(g.match_array(text)) == as_list(o.match(text))
    assert gpu_list(g.match_array(text)) == ModelAC(keys).match(text)


@pytest.mark.parametrize("seed", range(4))
def test_random_batch_ragged_docs(seed):
    rng = random.Random(100 + seed)
    keys = rand_keys(rng, 200, b"abcde", 1, 9)
    g = AC.compile(keys)
    o = orc.AC.compile(keys)
    docs = [bytes(rng.choice(b"abcde") for _ in range(rng.choice([0, 0, 1, 3, 50, 255, 256, 257, 1000, 5000])))
            for _ in range(60)]
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8)
    gh, gd = g.match_batch(corpus, offs)
    oh, od = o.match_batch(corpus, offs)
    assert np.array_equal(gd, od)
    assert gh.tobytes() == oh.tobytes()


@pytest.mark.parametrize("seed", range(4))
def test_random_utf8_chars_and_sep(seed):
    rng = random.Random(200 + seed)
    cps = [chr(c) for c in list(range(0x4E00, 0x4E30)) + list(range(97, 105)) + list(range(0x430, 0x438))] + [" "]
    keys, seen = [], set()
    while len(keys) < 150:
        k = "".join(rng.choice(cps[:-1]) for _ in range(rng.randint(1, 4)))
        if k not in seen:
            seen.add(k)
            keys.append(k)
    text = "".join(rng.choice(cps) for _ in range(4000))
    g = AC.compile(keys)
    o = orc.AC.compile(keys)
    assert gpu_list(g.match_array(text)) == as_list(o.match(text))
    assert gpu_list(g.match_array(text.encode())) == as_list(o.match(text.encode()))
    sep = BitArray(256)
    sep[32] = True
    assert gpu_list(g.match_array(text, sep)) == as_list(o.match(text, sep=(256, [32])))
    sep = BitArray(100)
    sep[32] = True
    sep[97] = True
    assert gpu_list(g.match_array(text.encode(), sep)) == as_list(o.match(text.encode(), sep=(100, [32, 97])))
    # chars + batch: per-document char offsets
    docs = [text[i:i + 333] for i in range(0, len(text), 333)]
    enc = [d.encode() for d in docs]
    offs = np.cumsum([0] + [len(d) for d in enc]).astype(np.uint64)
    gh, gd = g.match_batch(np.frombuffer(b"".join(enc), dtype=np.uint8), offs, chars=True)
    oh, od = o.match_batch(np.frombuffer(b"".join(enc), dtype=np.uint8), offs, chars=True)
    assert np.array_equal(gd, od) and gh.tobytes() == oh.tobytes()


def test_long_keys_grow_the_chunk():
    rng = random.Random(7)
    keys = [bytes(rng.choice(b"ab") for _ in range(n)) for n in (700, 300, 5, 2, 1)]
    keys = list(dict.fromkeys(keys))
    text = bytes(rng.choice(b"ab") for _ in range(3000)) + keys[0] + keys[1] + bytes(rng.choice(b"ab") for _ in range(3000))
    assert gpu_list(AC.compile(keys).match_array(text)) == as_list(orc.AC.compile(keys).match(text))


def test_capacity_error_reports_required():
    import ctypes as C

    ac = AC.compile(["a"])
    t = np.frombuffer(b"a" * 1000, dtype=np.uint8)
    out = np.zeros(10, dtype=orc.HIT_DTYPE)
    n = C.c_uint64(0)
    rc = N.lib().aha_ac_match_bytes(ac._h, t.ctypes.data, t.size, None, out.ctypes.data, 10, C.byref(n))
    assert rc == N.AHA_E_CAPACITY and n.value == 1000
    assert gpu_list(out) == [(i, i + 1, 0) for i in range(10)]


def test_dense_expansion_capacity_and_alignment(engine):
    """The hit-dense expansion (k2d_expand_dense: windows of 512 hits, hand-issued counted stores) at its corners: an output
    whose capacity ends inside a window / at a window's end / one hit short (AHA_E_CAPACITY with the count, nothing written behind
    the capacity), an output address that is not 16-byte aligned (the uncounted store path), chains longer than the record's
    count field beside short ones, many short documents.  Against the oracle (src/aha/ac.cr:265-278)."""
    import torch

    if engine not in ("v2", "u", "ur", "auto", "f"):
        pytest.skip("one byte-level and the character-level variants")
    for keys, text in ((["a" * k for k in range(1, 13)], "a" * 5000 + "b" + "a" * 3000),
                       (["中" * k for k in range(1, 10)] + ["中国"], ("中" * 40 + "国") * 150),
                       (["ab", "b", "abab", "bab", "abc"], "ababcab" * 3000)):
        ac = AC.compile(keys)
        o = orc.AC.compile(keys)
        raw = text.encode()
        corpus = np.frombuffer(raw, dtype=np.uint8)
        for doc in (np.array([0, len(raw)], dtype=np.uint64),
                    np.array(sorted(set(range(0, len(raw), 999)) | {len(raw)}), dtype=np.uint64)):
            oh, od = o.match_batch(corpus, doc, cap=len(raw) * 16)
            total = len(oh)
            assert total > 4 * len(raw) // 4  # (hit-dense: the dense expansion's batch)
            dc, dd = torch.from_numpy(corpus.copy()).cuda(), torch.from_numpy(doc.astype(np.int64)).cuda()
            dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
            want = torch.from_numpy(np.ascontiguousarray(oh).view(np.int32).reshape(-1, 3))
            big = torch.full((total + 64 + 4, 3), -7, dtype=torch.int32, device="cuda")
            for shift in (0, 1):  # rows of 12 bytes: shift 1 = an output address that is 4- but not 16-byte aligned
                for cap in (total, total + 5, total - 1, total - 300, 512, 513, 1000):
                    big.fill_(-7)
                    out = big[shift:shift + cap]
                    if cap >= total:
                        assert ac.match_batch_device(dc, dd, out, dho) == total
                        assert torch.equal(out[:total].cpu(), want)
                        assert np.array_equal(dho.cpu().numpy().astype(np.uint64), od)
                    else:
                        with pytest.raises(AhaError) as e:
                            ac.match_batch_device(dc, dd, out, dho)
                        assert e.value.code == N.AHA_E_CAPACITY and e.value.required == total
                    assert bool((big[shift + cap:] == -7).all()) and bool((big[:shift] == -7).all())  # nothing behind / in front of the buffer


def test_capacity_error_when_event_temp_overflows():
    # more events than the single-traversal engine's temp (sized from cap) can
    # hold: it must hand over to the two-pass engine and still report the count
    import ctypes as C

    ac = AC.compile(["a", "aa"])
    n_bytes = 3_000_000
    t = np.full(n_bytes, ord("a"), dtype=np.uint8)
    out = np.zeros(4, dtype=orc.HIT_DTYPE)
    n = C.c_uint64(0)
    rc = N.lib().aha_ac_match_bytes(ac._h, t.ctypes.data, t.size, None, out.ctypes.data, 4, C.byref(n))
    assert rc == N.AHA_E_CAPACITY and n.value == 2 * n_bytes - 1
    assert gpu_list(out) == [(0, 1, 0), (0, 2, 1), (1, 2, 0), (1, 3, 1)]


def test_config5_at_full_size(engine):
    """BASELINE config 5 at its full size -- 1 M keys, 256 MiB of hit-dense text, ~950 M hits (11 GB of triples) -- on the
    library's own choice of engine.  The oracle needs minutes for that, so the whole batch is checked through properties
    computed on the device (every hit's length is its key's, ends ascend inside a document, the per-document offsets cut the
    hit list where the documents change, their last entry is the count) and sampled documents from the front, the middle and
    the end of the batch against the oracle (src/aha/ac.cr:176-192, 265-278)."""
    if engine != "auto":
        pytest.skip("once, on the library's own choice")
    import torch

    blob, offs, nf = synth.keys(5)
    corpus, doc = synth.corpus(5, blob, offs, nf, n_bytes=1 << 28, doc_bytes=1 << 20)
    g = AC.compile_packed(blob, offs)
    g.set_profiling(True)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    D = doc.size - 1
    dho = torch.zeros(D + 1, dtype=torch.int64, device="cuda")
    cap = 4 * corpus.size
    out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    n = g.match_batch_device(dc, dd, out, dho)
    assert g.last_timing()["engine"] == 4 and n > 3 * corpus.size
    hits = out[:n]
    klen = torch.from_numpy((offs[1:] - offs[:-1]).astype(np.int32)).cuda()
    assert bool(((hits[:, 1] - hits[:, 0]) == klen[hits[:, 2].long()]).all())          # Hit(idx - len + 1, idx + 1, value)
    assert int(dho[0]) == 0 and int(dho[D]) == n and bool((dho[1:] >= dho[:-1]).all())
    # ends ascend inside a document: the only places where end[i + 1] < end[i] are document changes
    drop = torch.nonzero(hits[1:, 1] < hits[:-1, 1]).flatten() + 1
    cuts = torch.unique(dho[1:D])
    assert bool(torch.isin(drop, cuts).all())
    # every hit lies inside its document
    doc_of_hit = torch.searchsorted(dho[1:].contiguous(), torch.arange(n, device="cuda", dtype=torch.int64), right=True)
    dlen = (dd[1:] - dd[:-1])[doc_of_hit]
    assert bool((hits[:, 0] >= 0).all()) and bool((hits[:, 1].long() <= dlen).all())
    del doc_of_hit, dlen, drop
    # sampled documents against the oracle
    o = orc.AC.compile_packed(blob, offs)
    h_dho = dho.cpu().numpy()
    for d in (0, 1, D // 2, D - 2, D - 1):
        a, e = int(doc[d]), int(doc[d + 1])
        oh, _ = o.match_batch(corpus[a:e], np.array([0, e - a], dtype=np.uint64), cap=8 * (e - a))
        got = hits[int(h_dho[d]):int(h_dho[d + 1])].cpu().numpy()
        assert got.shape[0] == len(oh) and got.tobytes() == oh.tobytes(), d


def test_a_repeated_pass_is_reported(engine):
    """A batch denser than the capacity said overflows a chunk's event region: the match runs once more with full-size
    regions, bit-exact -- and aha_timing.repeats tells the caller that the call cost two passes."""
    if engine == "v1":
        pytest.skip("the two-pass engine has no event regions")
    import torch

    keys = ["ab", "b", "中", "中国"]
    ac = AC.compile(keys)
    ac.set_profiling(True)
    text = ("ab" * 20000 + "中国" * 20000).encode() + b"x" * 3_000_000  # hits in the first chunks only
    corpus = np.frombuffer(text, dtype=np.uint8)
    doc = np.array([0, corpus.size], dtype=np.int64)
    exp, _ = orc.AC.compile(keys).match_batch(corpus, doc.astype(np.uint64))
    dc, dd = torch.from_numpy(corpus.copy()).cuda(), torch.from_numpy(doc).cuda()
    out = torch.zeros((len(exp) + 64, 3), dtype=torch.int32, device="cuda")
    assert ac.match_batch_device(dc, dd, out) == len(exp)
    assert out[:len(exp)].cpu().numpy().tobytes() == exp.tobytes()
    # (the byte-level engine takes its slab pipeline for a capacity this small: no regions, nothing to repeat)
    assert ac.last_timing()["repeats"] == (1 if ac.last_timing()["engine"] == 4 else 0)
    # with room for one hit per byte the regions are full size at once
    out = torch.zeros((corpus.size, 3), dtype=torch.int32, device="cuda")
    assert ac.match_batch_device(dc, dd, out) == len(exp)
    assert ac.last_timing()["repeats"] == 0


# ---- the BASELINE configs at oracle-sized scale ------------------------------

@pytest.mark.parametrize("cfg,K,nbytes,docb", [(2, 1000, 1 << 22, 1 << 16), (3, 100_000, 1 << 23, 1 << 18),
                                                (5, 96_000, 1 << 21, 1 << 17)])
def test_config_parity(cfg, K, nbytes, docb):
    blob, offs, nf = synth.keys(cfg, K=K)
    corpus, doc = synth.corpus(cfg, blob, offs, nf, n_bytes=nbytes, doc_bytes=docb)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    gh, gd = g.match_batch(corpus, doc)
    oh, od = o.match_batch(corpus, doc, cap=len(gh) + 16)
    assert len(gh) == len(oh) and np.array_equal(gd, od)
    assert gh.tobytes() == oh.tobytes()
    if cfg == 3:  # String overload on the headline shape
        gh, gd = g.match_batch(corpus, doc, chars=True)
        oh, od = o.match_batch(corpus, doc, chars=True, cap=len(gh) + 16)
        assert gh.tobytes() == oh.tobytes() and np.array_equal(gd, od)


def test_config5_at_full_key_count(engine):
    """BASELINE config 5 at its real automaton size: 1 M keys -> 8-byte wide slots, an image far beyond L2, three
    nested key families, 3.5 hits per byte.  A fresh handle, so the first call runs the event regions into their
    overflow and is repeated with full-size regions (capi.cpp match_v2, rc 2); then a sparse call on the same
    handle.  Hits and per-document offsets against the oracle, byte and char offsets."""
    if engine not in ("v2", "auto"):
        pytest.skip("on the byte-level engine and on the library's own choice -- the character-level traversal over a unit "
                    "image with 23-bit bases (the two-pass engine runs the same automaton in test_config_parity)")
    import torch

    blob, offs, nf = synth.keys(5)
    assert offs.size - 1 == 1_000_000 and nf > 0
    corpus, doc = synth.corpus(5, blob, offs, nf, n_bytes=1 << 24, doc_bytes=1 << 20)
    g = AC.compile_packed(blob, offs)
    info = g.info
    assert info["slot_bytes"] == 8 and info["image_bytes"] > (100 << 20)
    want_engine = 2
    if engine == "auto":  # 3.9 M unit states: beyond 2^22 slots
        assert info["unit_enabled"] == 1 and info["unit_base_bits"] == 23 and info["unit_slots"] > 1 << 22
        assert info["unit_header_beside"] == 1 and info["unit_headers"] * 5 >= 3_900_000  # a quarter of its states own a fail header
        want_engine = 4
    g.set_profiling(True)
    o = orc.AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
    for chars in (False, True):
        oh, od = o.match_batch(corpus, doc, chars=chars, cap=corpus.size * 4)
        assert len(oh) > 3 * corpus.size  # hit-dense
        out = torch.zeros((len(oh) + 16, 3), dtype=torch.int32, device="cuda")
        n = g.match_batch_device(dc, dd, out, dho, chars=chars)
        assert n == len(oh)
        assert out[:n].cpu().numpy().tobytes() == oh.tobytes()
        assert np.array_equal(dho.cpu().numpy().astype(np.uint64), od)
        assert g.last_timing()["engine"] == want_engine
        del out
    # a sparse batch on the same handle (the keys' alphabet never occurs)
    sparse = np.frombuffer(b"0123456789 " * 100_000, dtype=np.uint8)
    sdoc = np.array([0, 300_000, sparse.size], dtype=np.uint64)
    gh, gd = g.match_batch(sparse, sdoc)
    oh, od = o.match_batch(sparse, sdoc)
    assert gh.tobytes() == oh.tobytes() and np.array_equal(gd, od)
    # and the dense batch again after it: the handle's pipeline choice must not depend on the call history
    oh, od = o.match_batch(corpus[: 1 << 21], np.array([0, 1 << 21], dtype=np.uint64), cap=1 << 24)
    gh, gd = g.match_batch(corpus[: 1 << 21], np.array([0, 1 << 21], dtype=np.uint64))
    assert gh.tobytes() == oh.tobytes() and np.array_equal(gd, od)


def test_unaligned_device_corpus_keeps_the_fast_engine(engine):
    """A device corpus that is not 16-byte aligned (a slice of a larger buffer) must not fall to the two-pass engine:
    same hits, same engine, at least 70 % of the aligned rate (one device-to-device copy in front of the match: ~0.18 ms
    for these 256 MiB against ~0.7 ms for the match)."""
    if engine not in ("v2", "u"):
        pytest.skip("on the byte-level and on the character-level engine")
    import torch

    blob, offs, nf = synth.keys(3)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 28)
    g = AC.compile_packed(blob, offs)
    g.set_profiling(True)
    big = torch.zeros(corpus.size + 64, dtype=torch.uint8, device="cuda")
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    res = {}
    for shift in (0, 1):
        view = big[shift:shift + corpus.size]
        view.copy_(torch.from_numpy(corpus))
        assert view.data_ptr() % 16 == shift
        try:
            n = g.match_batch_device(view, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"))
        except AhaError as e:
            n = e.required
        out = torch.zeros((n + 16, 3), dtype=torch.int32, device="cuda")
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            assert g.match_batch_device(view, dd, out) == n
            best = min(best, time.perf_counter() - t0)
        assert g.last_timing()["engine"] == (4 if engine in ("u", "u23", "uh") else 5 if engine == "f" else 6 if engine == "k" else 7 if engine == "p" else 2)
        res[shift] = (best, out[:n].cpu().numpy().tobytes())
    assert res[0][1] == res[1][1]
    assert res[0][0] / res[1][0] >= 0.7, (res[0][0], res[1][0])
    g.release_scratch()
    assert g.match_batch_device(big[1:1 + corpus.size], dd, out) == n  # scratch grows back


@pytest.mark.parametrize("chunk", [None, "4096", "8192", "16384", "32768"])
def test_prefix_filter_engine_edges(engine, monkeypatch, chunk):
    """The prefix-filter engine's own corners (scan_filter.hip), each against the oracle (src/aha/ac.cr:176-192, 265-278) and
    with the engine that answered: keys of exactly 3 bytes (the masked window) and of 64 (four blocks of steps), starts in the
    bytes before a chunk, keys across chunk and document boundaries, documents of a few bytes (more boundaries near a chunk
    than its table holds), empty documents, NUL bytes, a batch shorter than a window, more nested keys on one walk than a lane
    keeps (handed back), every chunk size."""
    if engine not in ("f", "auto"):
        pytest.skip("the prefix-filter engine's variants")
    if chunk:
        monkeypatch.setenv("AHA_FILTER_CHUNK", chunk)
    else:
        monkeypatch.delenv("AHA_FILTER_CHUNK", raising=False)
    rng = random.Random(77)

    def run(keys, text, cuts, want_engine):
        ac = AC.compile(keys)
        ac.set_profiling(True)
        o = orc.AC.compile(keys)
        t = np.frombuffer(text, dtype=np.uint8)
        doc = np.array(sorted([0, len(text)] + [c for c in (cuts or []) if c <= len(text)]), dtype=np.uint64)  # (twice: empty)
        gh, gd = ac.match_batch(t, doc)
        oh, od = o.match_batch(t, doc)
        assert np.asarray(gh).tobytes() == oh.tobytes() and np.array_equal(np.asarray(gd, dtype=np.uint64), od)
        if len(text) and want_engine:
            assert ac.last_timing()["engine"] == want_engine, (ac.last_timing(), ac.info["filter_prefix_bytes"])
        return len(oh)

    # keys of exactly three bytes and of sixty-four; filler the keys do not start with
    k3 = [b"abc", b"bcd", b"xyz", b"zzz"]
    long = bytes(rng.choice(b"qrstuv") for _ in range(64))
    keys = k3 + [long, long[:40], long[10:50], b"abcd" * 8]
    body = []
    for i in range(4000):
        # (the last key is sixteen key starts in 32 bytes: rare, or a chunk is denser than the engine takes)
        body.append(rng.choice(keys[:-1] if rng.random() < 0.9 else keys) if rng.random() < 0.2 else b" " * rng.randint(1, 40))
    text = b"".join(body)
    assert run(keys, text, None, 5) > 500
    # ... the same text cut into documents at random places (keys across boundaries are no hits), with empty documents
    cuts = sorted(rng.randint(0, len(text)) for _ in range(300))
    run(keys, text, cuts + cuts[5:9], 5)
    # keys that start in the bytes before a chunk and end in it, at every chunk size's boundaries; NUL bytes beside them
    pad = bytearray(b"-" * (3 * 32768 + 100))
    for edge in range(4096, len(pad) - 70, 4096):
        back = (1, 2, 3, 17, 63)[(edge // 4096) % 5]
        pad[edge - back:edge - back + 64] = long
        pad[edge + 2000:edge + 2003] = b"abc"
        pad[edge + 100] = 0
        pad[edge + 101:edge + 104] = b"xyz"
    run(keys, bytes(pad), [4096, 8192 + 1, 32768 - 1, 32768, 65536 + 3], 5)
    # documents of a few bytes: hundreds of boundaries in reach of one chunk
    tiny = (b"abc" + b"-" * 29) * 1500  # (a key start per 32 bytes: below the density the engine hands back)
    run(keys, tiny, list(range(0, len(tiny), 7)), 5)
    run(keys, tiny, list(range(0, len(tiny), 3)), 5)
    # a batch shorter than the filter's window, an empty one, one of exactly a key
    for t in (b"", b"ab", b"abc", b"zabcd", long):
        run(keys, t, None, None)
    # six keys nested on one trie path: more END steps than a walk keeps -- compile looks at the deepest nesting and builds no
    # filter for such a key set (round 5 found out in kf_walk, after the filter had run, call after call)
    nest = [b"abc", b"abcd", b"abcde", b"abcdef", b"abcdefg", b"abcdefgh"]
    t = (b"-" * 50 + b"abcdefgh") * 200
    assert run(nest, t, None, 2) == 6 * 200
    assert AC.compile(nest, host_only=True).info["filter_prefix_bytes"] == 0
    # ... four still fit
    assert run(nest[:4], t, None, 5) == 4 * 200

    # ---- char offsets (matcher.cr:34-39) over text that is not ASCII: kf_walk counts its chunk's continuation bytes itself.
    # Keys of 1-, 2-, 3- and 4-byte characters between ASCII ones, documents cut at character boundaries (several inside a
    # chunk, some across chunks, empty ones), a document start exactly at a chunk start, a chunk of nothing but continuation
    # bytes' owners (3-byte characters), the last chunk shorter than a chunk.
    def run_chars(keys, text, cut_chars, want_engine=5):
        ac = AC.compile(keys)
        ac.set_profiling(True)
        o = orc.AC.compile(keys)
        raw = text.encode()
        at = np.cumsum([0] + [len(c.encode()) for c in text])  # byte offset of every character (and of the end)
        t = np.frombuffer(raw, dtype=np.uint8)
        doc = np.array(sorted([0, len(raw)] + [int(at[c]) for c in cut_chars if c <= len(text)]), dtype=np.uint64)
        for chars in (True, False):
            gh, gd = ac.match_batch(t, doc, chars=chars)
            oh, od = o.match_batch(t, doc, chars=chars)
            assert np.asarray(gh).tobytes() == oh.tobytes() and np.array_equal(np.asarray(gd, dtype=np.uint64), od)
            assert ac.last_timing()["engine"] == want_engine, "chars=%s %s" % (chars, sorted(ac.last_timing().items()))
        return len(oh)

    ukeys = ["abc", "bcd", "naïve", "日本語", "x😀y", "ключ", "éé", "end of line"]
    assert AC.compile(ukeys, host_only=True).info["unit_enabled"] == 0
    parts = []
    for i in range(6000):
        r = rng.random()
        parts.append(rng.choice(ukeys) if r < 0.15 else rng.choice(["è", "月", "😁", "я"]) * rng.randint(1, 12) if r < 0.5
                     else " " * rng.randint(1, 30))
    utext = "".join(parts)
    assert run_chars(ukeys, utext, []) > 500
    cuts = sorted(rng.randint(0, len(utext)) for _ in range(200))
    run_chars(ukeys, utext, cuts + cuts[3:6])
    run_chars(ukeys, utext, list(range(0, len(utext), 5)))  # documents of five characters: the general way to a candidate's document
    # a document that starts exactly at byte 4096 / 8192 / ..: 3-byte characters up to it
    tri = "月" * (4096 // 3) + " " * (4096 % 3)  # (no key starts with it: a run of key starts is not this engine's text)
    utext2 = (tri + "日本語abc") * 9 + "naïve"
    run_chars(ukeys, utext2, [len(tri) * k + 6 * (k - 1) for k in range(1, 9)])
    # plain ASCII with char offsets: the count is the offset
    run_chars(ukeys, ("abc" + " " * 29) * 1500 + "end of line", [17, 1000])


def test_engine_selected(engine, monkeypatch):
    ac = AC.compile(["ab", "b"])
    ac.set_profiling(True)
    assert gpu_list(ac.match_array(b"abab" * 100))[:3] == [(0, 2, 0), (1, 2, 1), (2, 4, 0)]
    assert ac.last_timing()["engine"] in ((1,) if engine == "v1" else (2, 4, 5, 6, 7) if engine in ("u", "ur", "u23", "uh", "k", "p", "auto") else (2, 5) if engine == "f" else (2,))
    ac = AC.compile(["ab", "ba"])
    ac.set_profiling(True)
    assert gpu_list(ac.match_array(b"ab ba " * 60))[:3] == [(0, 2, 0), (3, 5, 1), (6, 8, 0)]
    assert ac.last_timing()["engine"] in ((1,) if engine == "v1" else (2, 4, 5, 6, 7) if engine in ("u", "ur", "u23", "uh", "k", "p", "auto") else (2, 5) if engine == "f" else (2,))
    if engine in ("u", "ur", "u23", "uh"):
        # the library's own choice (no AHA_ENGINE): keys of multi-byte characters get the character-level traversal for
        # byte- and char-offset batches, ASCII keys keep the byte-level one
        monkeypatch.delenv("AHA_ENGINE", raising=False)
        cjk = AC.compile(["中国", "国人", "人"])
        cjk.set_profiling(True)
        text = "中国人民" * 3000
        hits, _ = cjk.match_batch(np.frombuffer(text.encode(), dtype=np.uint8),
                                  np.array([0, len(text.encode())], dtype=np.uint64), cap=40000)
        assert cjk.info["unit_enabled"] == 1 and cjk.last_timing()["engine"] == 4 and len(hits) == 9000
        # ... also when the caller expects few hits (the byte-level engine would take its slab pipeline then)
        sparse = "民" * 200000 + "中国人"
        hits, _ = cjk.match_batch(np.frombuffer(sparse.encode(), dtype=np.uint8),
                                  np.array([0, len(sparse.encode())], dtype=np.uint64), cap=64)
        assert cjk.last_timing()["engine"] == 4 and [tuple(h) for h in hits.tolist()] == [
            (600000, 600006, 0), (600003, 600009, 1), (600006, 600009, 2)]
        asc = AC.compile(["abc", "bcd"])
        asc.set_profiling(True)
        # keys of 3 bytes: the prefix-filter engine, while the text has few places where a key could start ...
        got = asc.match_array(b"abcd" + b" " * 60 + (b"xbcdx" + b"-" * 59) * 300)
        assert asc.info["unit_enabled"] == 0 and asc.last_timing()["engine"] == 5
        assert len(got) == 302 and [tuple(h) for h in got[:3].tolist()] == [(0, 3, 0), (1, 4, 1), (65, 68, 1)]
        # ... and a text that is nothing but key starts goes back to the byte-level engine (same hits, kf_walk gives up)
        got = asc.match_array(b"abcd" * 3000)
        assert asc.last_timing()["engine"] == 2 and len(got) == 6000
        assert asc.info["filter_prefix_bytes"] == 3 and asc.info["filter_words"] == 1024
        # a handle whose batch came back stays away from the filter for a few calls (2, 4, .. 64), then tries it again
        sparse_text = b"-" * 5000 + b"abcd"
        engines = [asc.match_array(sparse_text).shape[0] and asc.last_timing()["engine"] for _ in range(6)]
        assert engines[0] == 2 and engines[-1] == 5 and sorted(engines) == engines
        # char offsets: the same engine -- while the batch is plain ASCII a char offset is a byte offset (kf_filter looks at
        # every byte anyway), and when it is not kf_walk counts the characters of its chunk
        one = lambda t: (np.frombuffer(t, dtype=np.uint8), np.array([0, len(t)], dtype=np.uint64))
        hits, _ = asc.match_batch(*one(sparse_text), chars=True)
        assert asc.last_timing()["engine"] == 5 and [tuple(h) for h in hits.tolist()] == [(5000, 5003, 0), (5001, 5004, 1)]
        utf = "é".encode() * 2500 + b"abcd"
        hits, _ = asc.match_batch(*one(utf), chars=True)
        assert asc.last_timing()["engine"] == 5 and [tuple(h) for h in hits.tolist()] == [(2500, 2503, 0), (2501, 2504, 1)]
        hits, _ = asc.match_batch(*one(utf))  # (byte offsets: non-ASCII text is nothing special)
        assert asc.last_timing()["engine"] == 5 and [tuple(h) for h in hits.tolist()] == [(5000, 5003, 0), (5001, 5004, 1)]
    # char offsets run on the character-level engine too, the separator filter on the byte-level engines only
    assert [tuple(h) for h in ac.match("abab")] == [(0, 2, 0), (1, 3, 1), (2, 4, 0)]
    assert ac.last_timing()["engine"] in ((1,) if engine == "v1" else (2, 4, 5, 6, 7) if engine in ("u", "ur", "u23", "uh", "k", "p", "auto") else (2, 5) if engine == "f" else (2,))
    sep = BitArray(256)
    sep[ord(" ")] = True
    assert [tuple(h) for h in ac.match("ab ba", sep)] == [(0, 2, 0), (3, 5, 1)]
    assert ac.last_timing()["engine"] == (1 if engine == "v1" else 2)


# ---- ragged batches over random automata ------------------------------------------

def _keys_ge2(rng, n, alphabet, maxlen):
    ks = set()
    n = min(n, sum(len(alphabet) ** k for k in range(2, maxlen + 1)) // 2)
    while len(ks) < n:
        ks.add(bytes(rng.choice(alphabet) for _ in range(rng.randint(2, maxlen))))
    return sorted(ks)


@pytest.mark.parametrize("seed", range(8))
def test_ragged_random_batches(seed):
    """Random automata (keys of two bytes and more), ragged batches: empty and one-byte documents, documents that end
    inside a key, NUL bytes, matches across chunk boundaries, runs of tiny documents."""
    rng = random.Random(4000 + seed)
    alphabet = [b"ab", b"abc", b"abcd\xe4\xb8\xad", bytes(range(0x61, 0x7B)), b"ab\xd0\xb0\xb1"][seed % 5]
    keys = _keys_ge2(rng, rng.randint(1, 300), alphabet, [6, 12, 40][seed % 3])
    g = AC.compile(keys)
    o = orc.AC.compile(keys)
    fill = alphabet + (b"\x00 " if seed % 2 else b" ")
    lens = [0, 0, 1, 2, 3, 50, 255, 256, 257, 1000, 4095, 4096, 4097, 9000]
    docs = [bytes(rng.choice(fill) for _ in range(rng.choice(lens))) for _ in range(40)]
    docs += [bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 5))) for _ in range(1500)]  # tiny documents
    docs += [bytes(rng.choice(alphabet) for _ in range(20000))]
    rng.shuffle(docs)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8)
    gh, gd = g.match_batch(corpus, offs)
    oh, od = o.match_batch(corpus, offs, cap=len(gh) + 16)
    assert np.array_equal(gd, od)
    assert gh.tobytes() == oh.tobytes()


def test_long_and_nested_keys():
    """Keys up to 240 bytes, a suffix-closed family (every suffix a key: output chains) and a
    broken chain, in one document that is several chunks long."""
    rng = random.Random(77)
    w = bytes(rng.choice(b"abcdefgh") for _ in range(16))
    keys = [w[j:] for j in range(15)] + [b"x" + w, bytes(rng.choice(b"ab") for _ in range(240)), b"ab" * 100]
    v = bytes(rng.choice(b"ijklmnop") for _ in range(16))
    keys += [v, v[1:], v[2:] + b"#"] + [v[j:] for j in range(3, 15)]
    keys = list(dict.fromkeys(keys))
    g = AC.compile(keys)
    o = orc.AC.compile(keys)
    text = b"".join(rng.choice([w, v, b"x" + w, keys[16], keys[17], b"ab", b" ", v[2:] + b"#"]) for _ in range(600))
    assert gpu_list(g.match_array(text)) == as_list(o.match(text))


def test_dense_output_chains_through_the_expansion_windows(engine):
    """Fifteen nested keys (every prefix of a run of one character): every position ends fifteen hits, so the
    character-level engine's fused expansion stages far more hits per block of events than one window holds, and one
    more key takes the chains beyond the record's 4-bit count -- that key set uses the general post passes.  Byte and
    char offsets, one long document and many short ones."""
    for n_keys, ch in ((15, "a"), (15, "中"), (16, "a")):
        keys = [ch * k for k in range(1, n_keys + 1)]
        g = AC.compile(keys)
        o = orc.AC.compile(keys)
        filler = "b" if ch == "a" else "国"
        text = ((ch * 700 + filler) * 40).encode()
        for doc in (np.array([0, len(text)], dtype=np.uint64),
                    np.concatenate([np.arange(0, len(text), 997, dtype=np.uint64), [len(text)]]).astype(np.uint64)):
            corpus = np.frombuffer(text, dtype=np.uint8)
            for chars in (False, True):
                if chars and ch == "中" and doc.size > 2:
                    continue  # documents cut inside characters: char offsets are defined for valid UTF-8 only
                gh, gd = g.match_batch(corpus, doc, chars=chars)
                oh, od = o.match_batch(corpus, doc, chars=chars)
                assert len(gh) == len(oh) > 200_000
                assert np.array_equal(np.asarray(gh).view(np.int32), np.asarray(oh).view(np.int32))
                assert np.array_equal(np.asarray(gd, dtype=np.uint64), np.asarray(od, dtype=np.uint64))


def test_malformed_utf8_and_ragged_units(engine):
    """Text that is NOT valid UTF-8 around keys that are: truncated characters, stray continuation bytes, bytes >= 0xF0,
    NUL, characters outside the keys' alphabet, documents that start or end inside a character, characters that
    straddle the 32-byte pieces and the chunk boundaries.  The character-level traversal must see exactly what the
    byte-level automaton sees (unit.hpp); the other engines run the same batch."""
    rng = random.Random(31)
    chars = ["a", "b", "é", "ж", "я", "中", "国", "人", "我", "是", "々", " "]
    keys = sorted({"".join(rng.choice(chars[:-1]) for _ in range(rng.randint(1, 5))) for _ in range(300)})
    g = AC.compile(keys)
    o = orc.AC.compile(keys)
    junk = [b"\xe4", b"\xe4\xb8", b"\xb8", b"\xad\xad", b"\xf0\x9f\x98\x80", b"\x00", b"\xc3", b"\xff", b"\xe4\xe4\xb8\xad",
            "€".encode(), "\uffee".encode(), "\u0100".encode()]
    docs = []
    for _ in range(60):
        parts = []
        for _ in range(rng.choice([0, 1, 3, 40, 400, 3000])):
            r = rng.random()
            parts.append(rng.choice(keys).encode() if r < 0.3 else rng.choice(chars).encode() if r < 0.85
                         else rng.choice(junk))
        d = b"".join(parts)
        cut = rng.randint(0, 2)  # some documents lose their first / last bytes: they start or end inside a character
        docs.append(d[cut:len(d) - rng.randint(0, 2)] if len(d) > 6 else d)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    corpus = np.frombuffer(b"".join(docs), dtype=np.uint8)
    gh, gd = g.match_batch(corpus, offs)
    oh, od = o.match_batch(corpus, offs, cap=len(gh) + 16)
    assert np.array_equal(gd, od)
    assert gh.tobytes() == oh.tobytes()


def test_device_resident_entry_point():
    import torch

    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 22, doc_bytes=1 << 16)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    oh, od = o.match_batch(corpus, doc)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    out = torch.zeros((len(oh) + 8, 3), dtype=torch.int32, device="cuda")
    dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
    n = g.match_batch_device(dc, dd, out, dho)
    assert n == len(oh)
    assert out[:n].cpu().numpy().tobytes() == oh.tobytes()
    assert np.array_equal(dho.cpu().numpy().astype(np.uint64), od)
    small = torch.zeros((5, 3), dtype=torch.int32, device="cuda")
    with pytest.raises(AhaError) as e:
        g.match_batch_device(dc, dd, small)
    assert e.value.code == N.AHA_E_CAPACITY and e.value.required == len(oh)


def test_host_entry_pipeline_and_buffer_api(engine):
    """The two ways a caller without a GPU framework reaches the device path.  (1) aha_ac_match_batch on host buffers:
    a 160 MiB batch is cut into three document ranges that run through the upload / match / download pipeline on
    private streams -- same hits and offsets as ONE device call on the whole batch, also when the capacity is too
    small first (count and offsets must still be exact) and with char offsets.  (2) aha_corpus_upload +
    aha_buffer_alloc / _download: the batch uploaded once through the C ABI, matched with the device entry point on
    raw pointers (no torch), hits downloaded.  The oracle checks the first documents."""
    if engine not in ("v2", "u"):
        pytest.skip("on the byte-level and on the character-level engine")
    import torch

    from aha_amd import DeviceCorpus

    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=160 << 20, doc_bytes=1 << 20)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dev_corpus = DeviceCorpus(corpus, doc, device=0)
    assert dev_corpus.n_docs == doc.size - 1 and dev_corpus.n_bytes == corpus.size
    for chars in (False, True):
        out = torch.zeros((corpus.size // 8, 3), dtype=torch.int32, device="cuda")
        dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
        n = g.match_batch_device(dc, dd, out, dho, chars=chars)
        want = out[:n].cpu().numpy().tobytes()
        want_off = dho.cpu().numpy().astype(np.uint64)
        d8 = 8  # the oracle on the first documents
        oh, od = o.match_batch(corpus[:int(doc[d8])], doc[:d8 + 1], chars=chars)
        assert want[:oh.nbytes] == oh.tobytes() and np.array_equal(want_off[:d8 + 1], od)
        gh, gd = g.match_batch(corpus, doc, chars=chars)          # host entry, generous capacity
        assert gh.tobytes() == want and np.array_equal(gd, want_off)
        gh, gd = g.match_batch(corpus, doc, chars=chars, cap=1000)  # too small first: AHA_E_CAPACITY, then again
        assert gh.tobytes() == want and np.array_equal(gd, want_off)
        ch, cd = g.match_corpus(dev_corpus, chars=chars)           # resident corpus, raw pointers, no torch
        assert ch.tobytes() == want and np.array_equal(cd, want_off)
    with pytest.raises(AhaError):
        DeviceCorpus(corpus[:100], np.array([0, 50, 40, 100], dtype=np.uint64))  # offsets must ascend


# ---- the headline configuration at FULL size ----------------------------------
@pytest.mark.parametrize("cfg", [2, 3])
def test_full_size_properties(cfg, engine):
    """BASELINE configs 2 (1k ASCII keys, 64 MiB) and 3 (100k keys, 1 GiB) at their full sizes: size-independent
    properties instead of a full oracle run -- ordering, every hit spells its
    key, document independence (any split of the batch gives the same hits),
    engine agreement by checksum -- plus the oracle on a sample of documents (cfg 2: on ALL of them: 64 MiB are cheap)."""
    if engine not in ("v2", "u", "f", "auto"):
        pytest.skip("full-size run: the byte-level and the character-level traversals, and the library's own choice")
    import hashlib

    import torch

    blob, offs, nf = synth.keys(cfg)
    corpus, doc = synth.corpus(cfg, blob, offs, nf)
    D = doc.size - 1
    g = AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dho = torch.zeros(D + 1, dtype=torch.int64, device="cuda")
    try:
        n = g.match_batch_device(dc, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"), dho)
    except AhaError as e:
        assert e.code == N.AHA_E_CAPACITY
        n = e.required
    out = torch.zeros((n + 16, 3), dtype=torch.int32, device="cuda")
    assert g.match_batch_device(dc, dd, out, dho) == n
    hits = out[:n].cpu().numpy()
    offsets = dho.cpu().numpy()
    assert offsets[0] == 0 and offsets[-1] == n and np.all(np.diff(offsets) >= 0)
    # (1) per document: end offsets ascend; start < end; offsets inside the document
    doc_of_hit = np.repeat(np.arange(D), np.diff(offsets))
    end = hits[:, 1].astype(np.int64)
    same_doc = doc_of_hit[1:] == doc_of_hit[:-1]
    assert np.all(end[1:][same_doc] >= end[:-1][same_doc])
    doc_len = np.diff(doc.astype(np.int64))
    assert np.all(hits[:, 0] >= 0) and np.all(hits[:, 0] < hits[:, 1]) and np.all(end <= doc_len[doc_of_hit])
    # (2) every sampled hit spells its key
    key_len = np.diff(offs.astype(np.int64))
    assert np.array_equal(hits[:, 1] - hits[:, 0], key_len[hits[:, 2]])
    rng = np.random.default_rng(1)
    for i in rng.integers(0, n, size=20000):
        s0 = int(doc[doc_of_hit[i]]) + int(hits[i, 0])
        k = int(hits[i, 2])
        assert corpus[s0:s0 + int(key_len[k])].tobytes() == blob[int(offs[k]):int(offs[k + 1])].tobytes()
    # (3) document independence: two half-batches concatenate to the same stream
    h = hashlib.sha256(hits.tobytes()).hexdigest()
    mid = D // 2
    parts = []
    for lo, hi in ((0, mid), (mid, D)):
        sub_doc = (doc[lo:hi + 1] - doc[lo]).astype(np.int64)
        sub = dc[int(doc[lo]):int(doc[hi])]
        m = g.match_batch_device(sub, torch.from_numpy(sub_doc).cuda(), out, None)
        parts.append(out[:m].cpu().numpy().copy())
    assert hashlib.sha256(np.concatenate(parts).tobytes()).hexdigest() == h
    # (4) the oracle on a sample of whole documents
    o = orc.AC.compile_packed(blob, offs)
    if cfg == 2:  # the whole batch against the oracle, offsets included
        oh, od = o.match_batch(corpus, doc)
        assert hits.tobytes() == oh.tobytes() and np.array_equal(offsets, od.astype(np.int64))
        return
    for d in rng.integers(0, D, size=6):
        oh, _ = o.match_batch(corpus[int(doc[d]):int(doc[d + 1])], np.array([0, doc_len[d]], dtype=np.uint64))
        assert hits[offsets[d]:offsets[d + 1]].tobytes() == oh.tobytes()


def test_corpus_beyond_4_gib(engine):
    """Corpus-level offsets are 64-bit (SURVEY 8 b: per-document Int32, corpus uint64): a batch of 4 GiB + 1 MiB.
    The documents on both sides of the 2^32 boundary and a sample of the others against the oracle; global
    invariants over all hits."""
    if engine != "v2":
        pytest.skip("done once, on the default engine")
    import torch

    blob, offs, nf = synth.keys(3)
    n_bytes = (1 << 32) + (1 << 20)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
    D = doc.size - 1
    assert int(doc[-1]) == n_bytes > 0xFFFFFFFF
    g = AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dho = torch.zeros(D + 1, dtype=torch.int64, device="cuda")
    try:
        n = g.match_batch_device(dc, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"), dho)
    except AhaError as e:
        assert e.code == N.AHA_E_CAPACITY
        n = e.required
    out = torch.zeros((n + 16, 3), dtype=torch.int32, device="cuda")
    assert g.match_batch_device(dc, dd, out, dho) == n
    del dc
    offsets = dho.cpu().numpy()
    assert offsets[0] == 0 and offsets[-1] == n and np.all(np.diff(offsets) >= 0)
    hits = out[:n].cpu().numpy()
    key_len = np.diff(offs.astype(np.int64))
    assert np.array_equal(hits[:, 1] - hits[:, 0], key_len[hits[:, 2]])
    doc_len = np.diff(doc.astype(np.int64))
    doc_of_hit = np.repeat(np.arange(D), np.diff(offsets))
    assert np.all(hits[:, 0] >= 0) and np.all(hits[:, 1].astype(np.int64) <= doc_len[doc_of_hit])
    o = orc.AC.compile_packed(blob, offs)
    cross = int(np.searchsorted(doc, 1 << 32, side="right")) - 1  # the document that contains byte 2^32
    rng = np.random.default_rng(4)
    for d in sorted(set([0, cross - 1, cross, min(cross + 1, D - 1), D - 1] + [int(x) for x in rng.integers(0, D, 4)])):
        oh, _ = o.match_batch(corpus[int(doc[d]):int(doc[d + 1])], np.array([0, doc_len[d]], dtype=np.uint64))
        assert hits[offsets[d]:offsets[d + 1]].tobytes() == oh.tobytes(), d


def test_single_large_document_vs_oracle(engine):
    """SURVEY 8d single-document variant: one 256 MiB document, so every chunk
    but the first starts in the middle of a sequence (warm-up overlap at scale).
    Full comparison with the oracle."""
    if engine not in ("v2", "u"):
        pytest.skip("the byte-level and the character-level traversals")
    import torch

    blob, offs, nf = synth.keys(3)
    corpus, _ = synth.corpus(3, blob, offs, nf, n_bytes=1 << 28)
    doc = np.array([0, corpus.size], dtype=np.uint64)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    oh, _ = o.match_batch(corpus, doc, cap=corpus.size // 16)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    out = torch.zeros((len(oh) + 16, 3), dtype=torch.int32, device="cuda")
    dho = torch.zeros(2, dtype=torch.int64, device="cuda")
    n = g.match_batch_device(dc, dd, out, dho)
    assert n == len(oh)
    assert out[:n].cpu().numpy().tobytes() == oh.tobytes()
    assert dho.cpu().tolist() == [0, n]


@pytest.mark.parametrize("chars", [False, True])
def test_exchange_pairs_round_trip(chars):
    """{end, value} pairs <-> Hit triples (aha_ac_hits_pack_device / _unpack_device): the multi-GPU payload."""
    import torch

    rng = random.Random(11)
    alphabet = ["a", "b", "我", "是", "ж"]
    keys = sorted({"".join(rng.choice(alphabet) for _ in range(rng.randint(1, 5))) for _ in range(200)})
    text = "".join(rng.choice(alphabet) for _ in range(20000))
    ac = AC.compile(keys)
    hits = ac.match_array(text, chars=chars)
    assert len(hits) > 1000
    ref = as_list(orc.AC.compile(keys).match(text, chars=chars))
    assert gpu_list(hits) == ref
    dev = torch.device("cuda:0")
    t = torch.from_numpy(hits.view(np.int32).reshape(-1, 3).copy()).to(dev)
    n = t.shape[0]
    pairs = torch.zeros((n + 7, 2), dtype=torch.int32, device=dev)
    out = torch.full((n + 5, 3), -7, dtype=torch.int32, device=dev)
    ac.hits_pack_device(t, n, pairs)
    assert torch.equal(pairs[:n], t[:, 1:3]) and int(pairs[n:].abs().sum()) == 0
    ac.hits_unpack_device(pairs, n, out, chars=chars)
    torch.cuda.synchronize()
    assert torch.equal(out[:n], t) and bool((out[n:] == -7).all())
    assert np.array_equal(ac.key_lengths(chars)[hits["value"]], hits["end"] - hits["start"])


def test_exchange_words_round_trip(engine):
    """The 4-byte exchange stream (aha_ac_hits_pack4_device / _unpack4_device): bit-identical to its CPU restatement
    (aha_amd/distributed.py pack4_host) and a lossless round trip -- dense hits, sparse hits with gaps beyond the
    step field, many tiny documents, char offsets, zero hits, a count that is not a multiple of 1024 -- in both word
    layouts: the key's length carried in the word (what 20 000 keys of at most 24 bytes get: 15 + 5 + 12 bits) and
    looked up on arrival (keys long enough that id and length leave fewer than 6 bits for the step)."""
    if engine != "v2":
        pytest.skip("independent of the match engine")
    import torch

    from aha_amd.distributed import PK4_BLOCK, pack4_host

    dev = torch.device("cuda:0")
    blob, offs, nf = synth.keys(3, K=20_000)
    g = AC.compile_packed(blob, offs)
    cases = []
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 22, doc_bytes=1 << 16)
    cases.append((corpus, doc, False))
    cases.append((corpus, doc, True))
    cases.append((corpus[: 1 << 20], np.arange(0, (1 << 20) + 1, 64, dtype=np.uint64), False))  # 16 Ki tiny documents
    sparse = np.full(1 << 22, 0x20, dtype=np.uint8)  # mostly blanks: a key every ~9000 bytes
    rng = np.random.default_rng(5)
    for p in range(100, sparse.size - 100, 9000):
        k = int(rng.integers(0, 20_000))
        kb = blob[int(offs[k]):int(offs[k + 1])]
        sparse[p:p + kb.size] = kb
    cases.append((sparse, np.array([0, sparse.size], dtype=np.uint64), False))
    cases.append((np.full(4096, 0x20, dtype=np.uint8), np.array([0, 4096], dtype=np.uint64), False))  # zero hits
    for corpus, doc, chars in cases:
        dc = torch.from_numpy(corpus).to(dev)
        dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
        out = torch.zeros((corpus.size // 4 + 16, 3), dtype=torch.int32, device=dev)
        n = g.match_batch_device(dc, dd, out, None, chars=chars)
        nb = (n + PK4_BLOCK - 1) // PK4_BLOCK
        words = torch.full((2 * n + nb + 8,), -1, dtype=torch.int32, device=dev)
        n_words = torch.zeros(1, dtype=torch.int64, device=dev)
        g.hits_pack4_device(out, n, words, n_words)
        torch.cuda.synchronize()
        fmt = g.stream_format()
        assert fmt == (12, 5)
        want = pack4_host(out[:n].cpu(), fmt)
        assert int(n_words[0]) == want.numel()
        assert torch.equal(words[: want.numel()].cpu(), want)
        back = torch.full((n + 3, 3), -7, dtype=torch.int32, device=dev)
        g.hits_unpack4_device(words, n, back, chars=chars)
        torch.cuda.synchronize()
        assert torch.equal(back[:n], out[:n]) and bool((back[n:] == -7).all())
        # several streams rebuilt by ONE launch (what an 8-GPU step does with its seven peers' streams): three copies
        # of this stream and an empty one, at word offsets and in an order that have nothing to do with the output order
        nw = int(n_words[0])
        land = torch.full((3 * nw + 5,), -1, dtype=torch.int32, device=dev)
        place = [2 * nw + 5, 0, nw + 2]
        for q in place:
            land[q:q + nw] = words[:nw]
        multi = torch.full((3 * n + 3, 3), -7, dtype=torch.int32, device=dev)
        g.hits_unpack4_segs_device(land, [(place[0], n, 2 * n), (0, 0, 0), (place[1], n, 0), (place[2], n, n)], multi,
                                   chars=chars)
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(multi[k * n:(k + 1) * n], out[:n]), k
        assert bool((multi[3 * n:] == -7).all())
        if n > 50_000 and not chars:
            assert want.numel() < n + n // 16  # about 4 bytes per hit where hits are dense
    with pytest.raises(AhaError):
        g.hits_pack4_device(torch.zeros((5000, 3), dtype=torch.int32, device=dev), 5000,
                            torch.zeros(5000, dtype=torch.int32, device=dev), n_words)  # capacity below 2n + n/1024
    # the other layout: one key of 5000 bytes beside the 20 000 -- 15 bits of id + 13 bits of length leave too few for the step
    rng = np.random.default_rng(11)
    long_key = rng.integers(0x61, 0x7B, size=5000, dtype=np.uint8)
    blob2 = np.concatenate([blob[: int(offs[-1])], long_key])
    offs2 = np.concatenate([offs, [int(offs[-1]) + 5000]]).astype(np.uint64)
    g2 = AC.compile_packed(blob2, offs2)
    assert g2.stream_format() == (12, 0)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 20, doc_bytes=1 << 16)
    corpus = np.concatenate([corpus, long_key, corpus[:1000]])
    doc = np.concatenate([doc, [corpus.size]]).astype(np.uint64)
    dc = torch.from_numpy(corpus).to(dev)
    dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
    out = torch.zeros((corpus.size // 4 + 16, 3), dtype=torch.int32, device=dev)
    n = g2.match_batch_device(dc, dd, out, None)
    assert n > 1000 and int((out[:n, 2] == 20_000).sum()) == 1
    words = torch.full((2 * n + n // 1024 + 8,), -1, dtype=torch.int32, device=dev)
    g2.hits_pack4_device(out, n, words, n_words)
    torch.cuda.synchronize()
    want = pack4_host(out[:n].cpu(), (12, 0))
    assert int(n_words[0]) == want.numel() and torch.equal(words[: want.numel()].cpu(), want)
    back = torch.full((n, 3), -7, dtype=torch.int32, device=dev)
    g2.hits_unpack4_device(words, n, back)
    torch.cuda.synchronize()
    assert torch.equal(back, out[:n])


class _LoopbackDist:
    """Stand-in for torch.distributed with two ranks whose peer is a copy of this rank: what rank 0 sends to rank 1
    comes back as rank 1's payload.  Exercises HitGatherer's device path (pack kernels, sizes read from the device,
    rebuild kernels) on the one GPU of the test box; the real transports are covered by the gloo tests (CPU tensors)
    and by bench.py --gpus N (RCCL)."""

    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Done:
        def wait(self):
            return None

    def get_rank(self, group=None):
        return 0

    def get_world_size(self, group=None):
        return 2

    def all_gather_into_tensor(self, out, inp, group=None):
        out.view(2, -1)[:] = inp

    def batch_isend_irecv(self, ops):
        sends = [o.tensor for o in ops if o.op == "isend"]
        recvs = [o.tensor for o in ops if o.op == "irecv"]
        assert len(sends) == len(recvs) == 1
        recvs[0].copy_(sends[0])
        return [self._Done()]


@pytest.mark.parametrize("exchange", ["triples", "pairs", "words"])
def test_hit_gatherer_device_path(engine, exchange):
    if engine != "v2":
        pytest.skip("independent of the match engine")
    import torch

    from aha_amd.distributed import HitGatherer

    dev = torch.device("cuda:0")
    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 22, doc_bytes=1 << 16)
    g = AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).to(dev)
    dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
    out = torch.zeros((corpus.size // 4, 3), dtype=torch.int32, device=dev)
    n = g.match_batch_device(dc, dd, out, None)
    hg = HitGatherer(_LoopbackDist(), dev, ac=g, exchange=exchange)
    allh, counts = hg.all_gatherv(out, n)
    assert counts == [n, n]
    assert torch.equal(allh[:n], out[:n]) and torch.equal(allh[n:], out[:n])
    hg.start(out, n, 0)
    hg.start(out, n, 1)
    for slot in (0, 1):
        h2, c2 = hg.finish(slot)
        assert c2 == counts and torch.equal(h2, allh)
    per_hit = 4.0 * hg.last_payload_elems / n
    assert per_hit == {"triples": 12.0, "pairs": 8.0}.get(exchange, per_hit) and per_hit <= 12.0
    if exchange == "words":
        assert per_hit < 4.2


@pytest.mark.parametrize("transport", ["copies", "self-rccl"])
def test_group_of_shards_on_one_device(engine, transport, monkeypatch):
    """aha_group_match_batch with three shards on cuda:0 (device list [0, 0, 0]): partition, concurrent matches on
    three handles / streams, all-gatherv of the 4-byte stream, rebuild of all arrived streams by one launch.
    "copies": the peers' streams travel by device-to-device copies.  "self-rccl" (AHA_GROUP_RCCL=self): every shard
    also sends its OWN stream to itself through RCCL -- one communicator of one rank per shard, grouped
    ncclSend/ncclRecv through the same payload()/landing() code as the real exchange -- and its part of the gathered
    buffer is rebuilt from what RCCL delivered: dlopen of librccl.so, the symbol signatures, ncclInt32,
    ncclCommInitAll and the ordering behind the pack kernels run on this one GPU (only the n > 1 topology does not).
    Same hits and offsets as the single handle and the oracle, in EVERY shard's gathered buffer."""
    if engine not in ("v2", "u"):
        pytest.skip("on the byte-level and on the character-level engine")
    from aha_amd import ACGroup

    if transport == "self-rccl":
        monkeypatch.setenv("AHA_GROUP_RCCL", "self")
    else:
        monkeypatch.delenv("AHA_GROUP_RCCL", raising=False)
    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 23, doc_bytes=1 << 16)
    grp = ACGroup.compile_packed(blob, offs, [0, 0, 0])
    o = orc.AC.compile_packed(blob, offs)
    for chars in (False, True):
        oh, od = o.match_batch(corpus, doc, chars=chars)
        gh, gd = grp.match_batch(corpus, doc, chars=chars, cap=16)  # too small first: the capacity protocol
        assert np.array_equal(gd, od) and gh.tobytes() == oh.tobytes()
        for shard in range(3):  # every device holds the whole ordered stream
            assert grp.download_shard(shard).tobytes() == oh.tobytes(), (chars, shard)
    t = grp.last_timing()
    assert t["n_devices"] == 3 and t["exchange"] == (2 if transport == "self-rccl" else 0) and t["n_hits"] == len(oh)
    assert t["packed"] == 1 and t["wire_bytes"] < 4.3 * len(oh)  # the 4-byte stream travelled between the shards
    # the resident entry (aha_group_corpus_upload + aha_group_match_batch_device): the ranges stay on the devices, the hits too
    res = grp.upload_corpus(corpus, doc)
    for chars in (True, False):
        oh, od = o.match_batch(corpus, doc, chars=chars)
        for _ in range(2):  # (the second call meets buffers of the right size)
            n, gd = grp.match_corpus(res, chars=chars)
            assert n == len(oh) and np.array_equal(gd, od)
            for shard in range(3):
                assert grp.download_shard(shard).tobytes() == oh.tobytes(), (chars, shard)
    t = grp.last_timing()
    assert t["n_devices"] == 3 and t["n_hits"] == len(oh) and t["packed"] == 1
    del res
    if transport == "self-rccl":  # a group of ONE shard has nothing to exchange -- except in this mode
        g1 = ACGroup.compile_packed(blob, offs, [0])
        gh, gd = g1.match_batch(corpus, doc)
        oh, od = o.match_batch(corpus, doc)
        assert np.array_equal(gd, od) and gh.tobytes() == oh.tobytes() and g1.download_shard(0).tobytes() == oh.tobytes()
        assert g1.last_timing()["exchange"] == 2
    # ragged: fewer documents than shards, empty documents, an empty batch
    g2 = ACGroup.compile(["ab", "b"], [0, 0, 0, 0])
    o2 = orc.AC.compile(["ab", "b"])
    for docs in ([b"abab"], [b"", b"ab", b""], [], [b"", b""]):
        offs2 = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
        c2 = np.frombuffer(b"".join(docs), dtype=np.uint8)
        gh, gd = g2.match_batch(c2, offs2)
        oh, od = o2.match_batch(c2, offs2)
        assert np.array_equal(gd, od) and gh.tobytes() == oh.tobytes()


def test_group_shards_in_turn_copy_their_hits_from_inside_their_pipelines(engine):
    """Shards that share a device run one after the other and copy the hits of every ~64 MiB range to their place in the
    caller's buffer from inside their own pipeline (group.cpp): two shards of two ranges each against the single handle,
    with a buffer that fits exactly, and with one that is a hit short -- AHA_E_CAPACITY, the count, nothing beyond cap."""
    if engine != "auto":
        pytest.skip("once, on the library's own choice")
    import ctypes as C
    from aha_amd import ACGroup

    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=200 << 20, doc_bytes=1 << 20)
    D = doc.size - 1
    doc_u64 = doc.astype(np.uint64)
    one = AC.compile_packed(blob, offs)
    want, want_dho = one.match_batch(corpus, doc, cap=corpus.size // 8)
    n = len(want)
    grp = ACGroup.compile_packed(blob, offs, [0, 0])
    hit_t = np.dtype([("start", "<i4"), ("end", "<i4"), ("value", "<i4")])
    out = np.full(n + 8, -7, dtype=np.int32).repeat(3).view(hit_t)  # guard hits behind the capacity
    dho = np.zeros(D + 1, dtype=np.uint64)
    got = C.c_uint64(0)
    rc = N.lib().aha_group_match_batch(grp._h, corpus.ctypes.data, doc_u64.ctypes.data, D, None, out.ctypes.data, n,
                                       dho.ctypes.data, C.byref(got))
    assert rc == 0 and got.value == n
    assert out[:n].tobytes() == np.asarray(want).tobytes() and np.array_equal(dho, np.asarray(want_dho, dtype=np.uint64))
    assert (out[n:]["start"] == -7).all()
    assert grp.last_timing()["n_devices"] == 2
    out2 = np.full(n + 8, -7, dtype=np.int32).repeat(3).view(hit_t)
    rc = N.lib().aha_group_match_batch(grp._h, corpus.ctypes.data, doc_u64.ctypes.data, D, None, out2.ctypes.data, n - 1,
                                       dho.ctypes.data, C.byref(got))
    assert rc == N.AHA_E_CAPACITY and got.value == n
    assert (out2[n - 1:]["start"] == -7).all()  # nothing was written at or beyond cap


def test_concurrent_calls_on_one_handle():
    """The handle is immutable after compile; concurrent #match calls must be safe (SURVEY.md 8 b, threading)."""
    import threading

    rng = random.Random(21)
    keys = rand_keys(rng, 400, b"abcd", 1, 6)
    ac = AC.compile(keys)
    o = orc.AC.compile(keys)
    # twelve threads: more than the eight scratch sets of a handle, so some calls wait for a set
    texts = [bytes(rng.choice(b"abcd") for _ in range(rng.randint(1000, 60000))) for _ in range(12)]
    want = [as_list(o.match(t)) for t in texts]
    got = [None] * len(texts)
    errs = []

    def work(i):
        try:
            for _ in range(5):
                got[i] = gpu_list(ac.match_array(texts[i]))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(texts))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert got == want


def test_concurrent_device_calls_lease_their_own_scratch(engine):
    """Two host threads, one stream each, one handle: the calls run on separate scratch sets (no handle-wide lock) and
    both give the oracle's answer."""
    if engine != "v2":
        pytest.skip("one engine is enough")
    import threading

    import torch

    blob, offs, nf = synth.keys(3, K=20_000)
    g = AC.compile_packed(blob, offs)
    o = orc.AC.compile_packed(blob, offs)
    jobs = []
    for seed in (0, 1, 2):
        corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 24, doc_bytes=1 << 18, seed=seed + 5)
        oh, _ = o.match_batch(corpus[: int(doc[4])], doc[:5])  # the oracle on the first four documents
        jobs.append((corpus, doc, oh))
    res = [None] * len(jobs)
    errs = []

    def work(i):
        try:
            corpus, doc, _ = jobs[i]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                dc = torch.from_numpy(corpus).cuda()
                dd = torch.from_numpy(doc.astype(np.int64)).cuda()
                out = torch.zeros((corpus.size // 8, 3), dtype=torch.int32, device="cuda")
                dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
                st.synchronize()
                for _ in range(8):
                    n = g.match_batch_device(dc, dd, out, dho, stream=st.cuda_stream)
                res[i] = (out[:n].cpu().numpy().copy(), dho.cpu().numpy().copy())
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for (corpus, doc, oh), (hits, dho) in zip(jobs, res):
        assert dho[4] == len(oh) and hits[: len(oh)].tobytes() == oh.tobytes()
    assert g.scratch_bytes() > 0


def test_scratch_is_bounded_by_the_capacity_the_caller_gives(engine):
    """the temp of a call follows the hits the caller allows for -- event regions: 16 B per hit of
    capacity + 1/8 B per input byte; slab pipeline (fewer than 16 hits per chunk): 41 B per hit of capacity + 80 MB --
    not 8 B per input byte; release_scratch gives it back to the device."""
    if engine != "v2":
        pytest.skip("the bound is the single-traversal engine's")
    import torch

    blob, offs, nf = synth.keys(3, K=20_000)
    n_bytes = 1 << 28
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes, doc_bytes=1 << 20)
    g = AC.compile_packed(blob, offs)
    dc = torch.from_numpy(corpus).cuda()
    dd = torch.from_numpy(doc.astype(np.int64)).cuda()
    dho = torch.zeros(doc.size, dtype=torch.int64, device="cuda")
    try:
        n = g.match_batch_device(dc, dd, torch.zeros((1, 3), dtype=torch.int32, device="cuda"), dho)
    except AhaError as e:
        n = e.required
    g.release_scratch()
    assert g.scratch_bytes() == 0
    g.set_profiling(True)
    cap = n + n // 8
    out = torch.zeros((cap, 3), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    assert g.match_batch_device(dc, dd, out, dho) == n
    assert g.last_timing()["engine"] == 2
    free1, _ = torch.cuda.mem_get_info()
    bound = 41 * cap + n_bytes // 8 + (128 << 20)
    assert g.scratch_bytes() <= bound, (g.scratch_bytes(), bound)
    assert free0 - free1 <= bound + (64 << 20), (free0 - free1, bound)  # what the device really lost (allocator slack)
    assert g.scratch_bytes() < 8 * n_bytes // 4                          # far below one event per input byte
    g.release_scratch()
    assert g.scratch_bytes() == 0
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 >= free0 - (64 << 20)


def test_device_entry_validates_its_offsets():
    import torch

    ac = AC.compile(["ab"])
    t = torch.from_numpy(np.frombuffer(b"abab" * 64, dtype=np.uint8).copy()).cuda()
    out = torch.zeros((256, 3), dtype=torch.int32, device="cuda")
    assert ac.match_batch_device(t, torch.tensor([0, 100, 256], dtype=torch.int64).cuda(), out) == 128
    for bad in ([1, 100, 256], [0, 100, 255], [0, 200, 100, 256], [0, 100, 300]):
        with pytest.raises(AhaError) as e:
            ac.match_batch_device(t, torch.tensor(bad, dtype=torch.int64).cuda(), out)
        assert e.value.code == N.AHA_E_INVALID
        # the verdict of a refused call does not outlive it, and nothing was written through the bad offsets
        assert ac.match_batch_device(t, torch.tensor([0, 100, 256], dtype=torch.int64).cuda(), out) == 128
    # a larger batch (the single-traversal pipelines validate on the device, in front of the traversal, without a read-back):
    # offsets that point far outside the text must not be followed
    big = torch.from_numpy(np.frombuffer(b"abab" * (1 << 18), dtype=np.uint8).copy()).cuda()
    out2 = torch.zeros((1 << 19, 3), dtype=torch.int32, device="cuda")
    good = torch.tensor([0, 1 << 19, 1 << 20], dtype=torch.int64).cuda()
    assert ac.match_batch_device(big, good, out2) == 1 << 19
    for bad in ([0, 1 << 40, 1 << 20], [0, (1 << 20) + 64, 1 << 20], [1 << 62, 1 << 19, 1 << 20]):
        with pytest.raises(AhaError) as e:
            ac.match_batch_device(big, torch.tensor(bad, dtype=torch.int64).cuda(), out2)
        assert e.value.code == N.AHA_E_INVALID
    assert ac.match_batch_device(big, good, out2) == 1 << 19


def test_bad_offsets_reach_the_host_through_the_prefix_filter_engines_hand_back(engine):
    """A char-offset call over text that is not plain ASCII is handed back by kf_walk with a plain store to the same word that
    holds k_check_docs' verdict: the verdict must win -- the call is refused with AHA_E_INVALID instead of being repeated on the
    byte-level engine with offsets nobody validated (round-5 advice; src/aha/matcher.cr:34-39 is the overload)."""
    if engine not in ("f", "auto"):
        pytest.skip("the prefix-filter engine's hand-back")
    import torch

    ac = AC.compile(["abcd", "wxyz"])
    assert ac.info["filter_prefix_bytes"] == 4
    text = ("é" * 50 + "abcd wxyz ").encode() * 40
    t = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    n = t.numel()
    out = torch.zeros((4096, 3), dtype=torch.int32, device="cuda")
    good = torch.tensor([0, n // 2, n], dtype=torch.int64).cuda()
    want = ac.match_batch_device(t, good, out, chars=True)
    assert want == 80
    for bad in ([1, n // 2, n], [0, n // 2, n - 1], [0, n, n // 2, n], [0, n // 2, n + 44], [0, 1 << 40, n]):
        for chars in (True, False):
            with pytest.raises(AhaError) as e:
                ac.match_batch_device(t, torch.tensor(bad, dtype=torch.int64).cuda(), out, chars=chars)
            assert e.value.code == N.AHA_E_INVALID
        assert ac.match_batch_device(t, good, out, chars=True) == want


def test_sequence_longer_than_int32_is_rejected():
    import ctypes as C

    ac = AC.compile(["a"])
    offs = np.array([0, 1 << 31], dtype=np.uint64)  # Int32 offsets (matcher.cr:3-5): one sequence < 2^31 bytes
    t = np.zeros(16, dtype=np.uint8)
    n = C.c_uint64(0)
    rc = N.lib().aha_ac_match_batch(ac._h, t.ctypes.data, offs.ctypes.data, 1, None, None, 0, None, C.byref(n))
    assert rc == N.AHA_E_TOO_LONG


def test_hits_kept_on_the_device_and_a_replicated_handle(engine):
    """aha_ac_match_batch_keep (host corpus in, hits left in the caller's device buffer) gives the hits of aha_ac_match_batch,
    also over several ~64 MiB ranges; aha_ac_replicate's copy (same keys, nothing compiled again) matches like the oracle."""
    if engine not in ("v2", "auto"):
        pytest.skip("on the byte-level engine and on the library's own choice")
    import torch

    blob, offs, nf = synth.keys(3, K=20_000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=160 << 20, doc_bytes=1 << 20)
    ac = AC.compile_packed(blob, offs)
    for chars in (False, True):
        want, wd = ac.match_batch(corpus, doc, chars=chars, cap=corpus.size // 16)
        d_hits = torch.zeros((len(want) + 8, 3), dtype=torch.int32, device="cuda")
        n, gd = ac.match_batch_keep(corpus, doc, d_hits, chars=chars)
        assert n == len(want) and np.array_equal(gd, wd)
        assert d_hits[:n].cpu().numpy().tobytes() == want.tobytes()
        with pytest.raises(AhaError) as e:  # a buffer one hit short: the count, no overrun
            ac.match_batch_keep(corpus, doc, d_hits[: n - 1], chars=chars)
        assert e.value.code == N.AHA_E_CAPACITY and e.value.required == n
    twin = ac.replicate(0)
    assert twin.info["n_keys"] == ac.info["n_keys"] and twin.info["unit_enabled"] == ac.info["unit_enabled"]
    o = orc.AC.compile_packed(blob, offs)
    small, sdoc = corpus[:int(doc[8])], doc[:9]
    oh, od = o.match_batch(small, sdoc)
    th, td = twin.match_batch(small, sdoc, cap=len(oh) + 4)
    assert th.tobytes() == oh.tobytes() and np.array_equal(td, od)
    del ac  # the copy owns its image
    th, td = twin.match_batch(small, sdoc, chars=True, cap=len(oh) + 4)
    oh, od = o.match_batch(small, sdoc, chars=True)
    assert th.tobytes() == oh.tobytes() and np.array_equal(td, od)


def test_match_leaves_the_exchange_stream(engine):
    """aha_ac_match_batch_device_stream: the hits AND their 4-byte exchange stream from one call (the pack kernels behind the
    match): bit-identical to the pack's CPU restatement, rebuilt exactly by the unpack -- dense and sparse text, tiny documents,
    char offsets, zero hits."""
    if engine not in ("v2", "u", "ur", "auto"):
        pytest.skip("the byte-level engine (packs behind the match), the fused expansion, the general post passes, the library's choice")
    import torch

    from aha_amd.distributed import PK4_BLOCK, pack4_host

    dev = torch.device("cuda:0")
    blob, offs, nf = synth.keys(3, K=20_000)
    g = AC.compile_packed(blob, offs)
    g.set_profiling(True)
    cases = []
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 23, doc_bytes=1 << 16)
    cases.append((corpus, doc, False))
    cases.append((corpus, doc, True))
    cases.append((corpus[: 1 << 20], np.arange(0, (1 << 20) + 1, 64, dtype=np.uint64), False))  # 16 Ki tiny documents
    sparse = np.full(1 << 22, 0x20, dtype=np.uint8)
    rng = np.random.default_rng(5)
    for p in range(100, sparse.size - 100, 9000):
        k = int(rng.integers(0, 20_000))
        kb = blob[int(offs[k]):int(offs[k + 1])]
        sparse[p:p + kb.size] = kb
    cases.append((sparse, np.array([0, sparse.size], dtype=np.uint64), False))
    cases.append((np.full(4096, 0x20, dtype=np.uint8), np.array([0, 4096], dtype=np.uint64), False))  # zero hits
    for corpus, doc, chars in cases:
        dc = torch.from_numpy(corpus).to(dev)
        dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
        cap = corpus.size // 8 + 16
        ref = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
        n = g.match_batch_device(dc, dd, ref, None, chars=chars)
        out = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
        words = torch.full((2 * cap + cap // 1024 + 2,), -1, dtype=torch.int32, device=dev)
        n_words = torch.zeros(1, dtype=torch.int64, device=dev)
        dho = torch.zeros(doc.size, dtype=torch.int64, device=dev)
        assert g.match_batch_device(dc, dd, out, dho, chars=chars, words=words, n_words=n_words) == n
        assert torch.equal(out[:n], ref[:n])
        nw = int(n_words[0])
        nb = (n + PK4_BLOCK - 1) // PK4_BLOCK
        assert n + 2 * nb <= nw + nb <= 2 * n + nb + 1 or n == 0
        want = pack4_host(ref[:n].cpu(), g.stream_format())
        assert nw == want.numel() and torch.equal(words[:nw].cpu(), want)
        back = torch.full((n + 3, 3), -7, dtype=torch.int32, device=dev)
        g.hits_unpack4_device(words, n, back, chars=chars)
        torch.cuda.synchronize()
        assert torch.equal(back[:n], ref[:n]) and bool((back[n:] == -7).all())
    # too little room for the stream is said before anything runs
    with pytest.raises(AhaError) as e:
        g.match_batch_device(dc, dd, out, None, words=words[:64], n_words=n_words)
    assert e.value.code == N.AHA_E_CAPACITY


def test_skip_engine_selection_and_edges(engine, monkeypatch):
    """The skip-ahead traversal (scan_skip.hip, aha_timing.engine = 6), each case against the oracle (src/aha/ac.cr:176-192,
    265-278): which calls it answers -- byte offsets of a key set without a one-character key --, marks hundreds of bytes apart
    (pseudo jumps at the end of a lane's two words of marks), keys across the 124-byte limit of those words, across chunk ends
    and document boundaries, documents of a few bytes, empty documents, malformed UTF-8, NUL bytes, batches shorter than a
    piece of the marking pass and not a multiple of it."""
    if engine not in ("k", "p"):
        pytest.skip("the skip-ahead traversal's and the pair engine's own cases")
    E = 6 if engine == "k" else 7
    rng = random.Random(11)
    keys = ["中国人", "ab", "abc", "bcab", "我是", "是中国", "国人民", "яж", "жя中"]
    g = AC.compile(keys)
    g.set_profiling(True)
    o = orc.AC.compile(keys)
    assert g.info["unit_enabled"] == 1 and g.info["skip_filter_words"] >= 1024 and g.info["skip_pairs"] == 8
    assert g.info["pair_engine"] == (1 if engine == "p" else 0)

    gave_up = [False]

    def check(text, doc, want_engine=None, **kw):
        want_engine = E if want_engine is None else want_engine
        t = np.frombuffer(text, dtype=np.uint8)
        d = np.asarray(doc, dtype=np.uint64)
        gh, gd = g.match_batch(t, d, **kw)
        oh, od = o.match_batch(t, d, cap=len(gh) + 16, **kw)
        assert len(gh) == len(oh) and np.array_equal(gd, od) and gh.tobytes() == oh.tobytes(), (text[:200], doc[:8])
        if len(text):
            # (the pair engine gives a batch with documents of a few bytes -- two boundaries in one 32-byte piece -- to engine 4,
            # and stops trying on a handle that made it give up three times)
            ok = (want_engine, 4) if (engine == "p" and (len(doc) > 2 or gave_up[0])) else (want_engine,)
            assert g.last_timing()["engine"] in ok
            gave_up[0] = gave_up[0] or (engine == "p" and want_engine == 7 and g.last_timing()["engine"] == 4)

    # (a hit per ~8 bytes: the pair engine keeps up to 320 events per tile of 2 KiB, a denser batch is engine 4's)
    text = "我是中国人民 -- ab 国々 abc ･･･ bcab 民民民 яжя中 ".encode() * 50
    check(text, [0, len(text)])
    check(text, [0, len(text)], want_engine=4, chars=True)  # char offsets: the walk that counts characters
    # sparse: marks hundreds of bytes apart, keys at every distance from the window and chunk ends
    parts = []
    for i in range(400):
        parts.append((" " * rng.randint(0, 300) + rng.choice("xyz々民") * rng.randint(0, 40)).encode())
        parts.append(rng.choice(keys).encode() * rng.randint(1, 3))
    text = b"".join(parts)
    check(text, [0, len(text)])
    cuts = sorted({0, len(text)} | {rng.randrange(0, len(text)) for _ in range(200)})
    check(text, cuts + [len(text)] * 3)  # documents cut anywhere (also inside characters), empty documents at the end
    check(text, [0, 0, 0, 5, 5, 7, len(text)])
    # malformed UTF-8, NUL
    bad = [b"\xe4", b"\xe4\xb8", b"\xb8", b"\xad\xad", b"\xf0\x9f\x98\x80", b"\x00", b"\xc3", b"\xff", b"\xe4\xe4\xb8\xad"]
    parts = []
    for i in range(3000):
        x = rng.random()
        parts.append(rng.choice(keys).encode() if x < 0.3 else rng.choice(bad) if x < 0.5 else rng.choice("ab中国 яж").encode())
    text = b"".join(parts)
    check(text, [0, len(text)])
    cuts = sorted({0, len(text)} | {rng.randrange(0, len(text)) for _ in range(50)})
    check(text, cuts)
    # short batches: below a piece, one byte over a piece, empty
    for n in (0, 1, 2, 5, 63, 64, 65, 127, 129, 4095, 4097):
        t = ("ab中国人abc" * 500).encode()[:n]
        check(t, [0, len(t)], want_engine=E if n >= 64 else 4)
    # a key set with a one-character key keeps the plain character-level traversal
    uk = AC.compile(["中", "中国", "国人"])
    uk.set_profiling(True)
    assert uk.info["skip_filter_words"] == 0
    t = "中国人".encode() * 100
    assert len(uk.match_array(t)) == 300 and uk.last_timing()["engine"] == 4
