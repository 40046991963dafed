"""Pins the CPU oracle (oracle/aha_oracle.c) to the reference's own
known-answer specs (tests/golden/reference_kats.json, transcribed from
spec/*.cr by scripts/make_reference_kats.py) and to hand-derived cases that
follow from the cited reference lines ("derived, not reference-pinned")."""
import json
import os

import pytest

import pyoracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(G, "reference_kats.json"), encoding="utf-8"))


def _sep(s):
    return None if s is None else (s["size"], s["set"])


@pytest.mark.parametrize("kat", KATS["ac_match"], ids=lambda k: k["cite"][:24] + k["api"])
def test_reference_ac_match_kats(kat):
    ac = orc.AC.compile(kat["keys"])
    # Array(Char) overload (ac.cr:288-295): chars re-encode to the same UTF-8
    # bytes and the same char_map, so both APIs share the String path here.
    hits = ac.match(kat["text"], chars=True, sep=_sep(kat["sep"]))
    assert [[int(h["end"]), int(h["value"])] for h in hits] == kat["expect_end_value"]


def test_reference_first_kat_byte_level():
    # byte-level triples implied by spec/ac_spec.cr:5-12
    ac = orc.AC.compile(["我", "我是", "是中"])
    hits = ac.match("我是中国人".encode())
    assert [tuple(h) for h in hits.tolist()] == [(0, 3, 0), (0, 6, 1), (3, 9, 2)]
    hits = ac.match("我是中国人")
    assert [tuple(h) for h in hits.tolist()] == [(0, 1, 0), (0, 2, 1), (1, 3, 2)]


@pytest.mark.parametrize("kat", KATS["ac_match_longest"], ids=lambda k: k["cite"][-5:])
def test_reference_match_longest_kats(kat):
    for via_trie in (True, False):
        if via_trie:  # spec builds a Cedar and calls AC.compile(trie)
            t = orc.Cedar()
            for k in kat["keys"]:
                t.insert(k)
            ac = orc.AC.compile(t)
        else:
            ac = orc.AC.compile(kat["keys"])
        for chars in (False, True):
            hits = ac.match_longest(kat["text"], kat["intersectable"], chars=chars)
            got = [[int(h["start"]), int(h["end"]), ac.key(int(h["value"])).decode()] for h in hits]
            assert got == kat["expect"]


def test_reference_cedar_insert_delete():
    t = orc.Cedar()
    for op, key, want in KATS["cedar_insert_delete"]["ops"]:
        got = t.insert(key) if op == "insert" else t.delete(key)
        assert got == want, (op, key)


def test_reference_cedar_words_roundtrip():
    words = open(os.path.join(G, "cedar_words.txt"), encoding="utf-8").read().split("\n")[:-1]
    assert len(words) == KATS["cedar_words"]["count"]
    t = orc.Cedar()
    for w in words:
        t.insert(w)
    # the list holds duplicates?  the spec asserts trie.size == lines.size
    assert t.key_num == len(words)
    for i, w in enumerate(words):
        assert t.key(i).decode() == w
        assert t.get(w) == i


# ---- derived from the cited lines, not reference-pinned ----

def test_derived_subset_semantics():
    # state "abc" is a path node without END: nothing reported although "c" ends there
    assert len(orc.AC.compile(["c", "abcd"]).match(b"abc")) == 0
    # all-end chain
    h = orc.AC.compile(["a", "aa"]).match(b"aa")
    assert [tuple(x) for x in h.tolist()] == [(0, 1, 0), (0, 2, 1), (1, 2, 0)]
    # broken chain: fail("abc") = "bc" is a non-end path node -> "c" never reported
    h = orc.AC.compile(["xabc", "abc", "bcz", "c"]).match(b"xabc")
    assert [tuple(x) for x in h.tolist()] == [(0, 4, 0), (1, 4, 1)]


def test_derived_sep_unfiltered_trace():
    # unfiltered output of the "ac with sep" keys
    h = orc.AC.compile(["a", "aa"]).match(b"a aaa")
    assert [tuple(x) for x in h.tolist()] == [(0, 1, 0), (2, 3, 0), (2, 4, 1), (3, 4, 0), (3, 5, 1), (4, 5, 0)]


def test_derived_error_paths():
    with pytest.raises(orc.OracleError) as e:
        orc.AC.compile(["ab", "cd", "ab"])
    assert e.value.code == orc.E_DUP_KEY and e.value.key_index == 2
    with pytest.raises(orc.OracleError) as e:
        orc.AC.compile(["ab", ""])
    assert e.value.code == orc.E_EMPTY_KEY and e.value.key_index == 1
    with pytest.raises(orc.OracleError) as e:
        orc.AC.compile([b"a\x00b"])
    assert e.value.code == orc.E_ZERO_BYTE
    with pytest.raises(orc.OracleError) as e:
        orc.AC.compile(["a"]).match(b"a", sep=(257, []))
    assert e.value.code == orc.E_SEP_SIZE


def test_derived_nul_contract():
    # NUL resets to root and reports nothing (wherever the reference is defined)
    ac = orc.AC.compile(["ab", "abc", "b"])
    h = ac.match(b"ab\x00abc\x00b")
    assert [tuple(x) for x in h.tolist()] == [(0, 2, 0), (1, 2, 2), (3, 5, 0), (4, 5, 2), (3, 6, 1), (7, 8, 2)]


def test_empty_inputs():
    ac = orc.AC.compile(["a"])
    assert len(ac.match(b"")) == 0
    hits, dho = ac.match_batch(b"aa", [0, 0, 1, 1, 2])
    assert dho.tolist() == [0, 0, 1, 1, 2]
    assert [tuple(x) for x in hits.tolist()] == [(0, 1, 0), (0, 1, 0)]
