"""The hash image (aha_amd/csrc/hash.hpp) against the oracle, on the CPU: the tables are built by the library (host only) and
walked by tests/hashsim.py -- the sequential walk of scan_hash.hip and the start-parallel formulation (pair filter, goto walks,
blocking by the earlier starts' reach)."""
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC
from aha_amd import _native as N
from hashsim import HashSim
from test_oracle_vs_model import as_list
from test_unit_twin import CHARS, rand_word


def compile_hash(keys, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "hash")
    return AC.compile(keys, host_only=True)


@pytest.mark.parametrize("seed", range(12))
def test_hash_image_matches_the_oracle(seed, monkeypatch):
    rng = random.Random(100 + seed)
    keys = sorted({rand_word(rng, 2, rng.choice([2, 4, 7])) for _ in range(rng.choice([3, 40, 400]))})
    ac = compile_hash(keys, monkeypatch)
    sim = HashSim(ac)
    o = orc.AC.compile(keys)
    for _ in range(6):
        parts = []
        for _ in range(rng.randint(0, 120)):
            r = rng.random()
            if r < 0.35:
                parts.append(rng.choice(keys).encode())
            elif r < 0.8:
                parts.append(rng.choice(CHARS).encode())
            elif r < 0.9:  # malformed: truncated characters, stray continuation bytes, bytes >= 0xF0, NUL
                parts.append(rng.choice([b"\xe4", b"\xe4\xb8", b"\xb8", b"\xad\xad", b"\xf0\x9f\x98\x80", b"\x00", b"\xc3",
                                         b"\xff", b"\xe4\xe4\xb8\xad"]))
            else:
                parts.append(bytes([rng.randrange(1, 256)]))
        text = b"".join(parts)
        want = as_list(o.match(text))
        assert sim.match(text) == want, (keys, text)
        assert sim.match_parallel(text) == want, (keys, text)


def test_nested_and_broken_chains(monkeypatch):
    """cfg 5's families in small: runs c, cc, ccc..., suffix-closed sets, and a chain whose middle is a path but no key -- the
    reference stops there (ac.cr:106-108), and so must the start-parallel form (the longer path blocks the shorter keys)."""
    keys = ["aa", "aaa", "aaaa", "aaaaa", "bcdef", "cdef", "def", "ef", "wxyz", "xyz", "yz#", "zz", "xy"]
    ac = compile_hash(keys, monkeypatch)
    sim = HashSim(ac)
    o = orc.AC.compile(keys)
    for text in (b"aaaaaaaaaa", b"abcdefef", b"wxyzz", b"wxyz#", b"xyzzxyz#", b"aaaaabcdefwxyz", b"yz#yz"):
        want = as_list(o.match(text))
        assert sim.match(text) == want and sim.match_parallel(text) == want, text


def test_key_sets_without_a_hash_image(monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "hash")
    one = AC.compile(["我", "我是", "是中"], host_only=True)  # a key of one character: the unit image's walk
    assert one.info["unit_enabled"] == 1 and one.export(N.AHA_IMG_HASH_PARAMS, np.uint32).size == 0
    monkeypatch.setenv("AHA_ENGINE", "unit")
    assert AC.compile(["我是", "是中"], host_only=True).export(N.AHA_IMG_HASH_PARAMS, np.uint32).size == 0  # not asked for


def test_hash_image_of_the_headline_keys(monkeypatch):
    """cfg 3's 100 000 keys: the tables place (one multiplier, displacements of one byte), the filter is a quarter full, and
    both walks agree with the oracle on 32 KiB of cfg 3's text."""
    from aha_amd import synth
    monkeypatch.setenv("AHA_ENGINE", "hash")
    blob, offs, nf = synth.keys(3)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 15, doc_bytes=1 << 15)
    ac = AC.compile_packed(blob, offs, host_only=True)
    par = ac.export(N.AHA_IMG_HASH_PARAMS, np.uint32)
    assert par.size == 8 and par[4] > 80000 and par[6] < 400
    text = corpus[:int(doc[1])].tobytes()
    o = orc.AC.compile_packed(blob, offs)
    oh, _ = o.match_batch(corpus[:int(doc[1])], np.array([0, int(doc[1])], dtype=np.uint64))
    want = [tuple(int(x) for x in h) for h in oh]
    sim = HashSim(ac)
    assert sim.match(text) == want
    assert sim.match_parallel(text) == want
