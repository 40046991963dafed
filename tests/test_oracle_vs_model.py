"""Randomized agreement between the C oracle (Cedar restatement) and the
independent Python dict-trie model (tests/pymodel.py)."""
import random

import pytest

import pyoracle as orc
from pymodel import ModelAC


def rand_keys(rng, n, alphabet, lo, hi):
    n = min(n, max(1, sum(len(set(alphabet)) ** k for k in range(lo, hi + 1)) // 2))  # never ask for more than exist
    seen = set()
    keys = []
    while len(keys) < n:
        k = bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi)))
        if k not in seen:
            seen.add(k)
            keys.append(k)
    return keys


def as_list(h):
    return [tuple(int(v) for v in x) for x in h.tolist()]


@pytest.mark.parametrize("seed", range(40))
def test_small_alphabet_nested(seed):
    # tiny alphabets force nested suffix/prefix keys: broken and complete chains
    rng = random.Random(seed)
    alphabet = b"ab" if seed % 3 == 0 else (b"abc" if seed % 3 == 1 else b"abcd\xe4\xb8")
    keys = rand_keys(rng, rng.randint(1, 40), alphabet, 1, 7)
    text = bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 400)))
    o = orc.AC.compile(keys)
    m = ModelAC(keys)
    assert as_list(o.match(text)) == m.match(text)
    tb = set(m.textbook(text))
    assert set(m.match(text)) <= tb  # the reference reports a subset of the true matches


@pytest.mark.parametrize("seed", range(12))
def test_many_keys_relocation(seed):
    # enough keys to exercise Cedar's resolve/relocation and block lists
    rng = random.Random(1000 + seed)
    alphabet = bytes(range(1, 256)) if seed % 2 else bytes(range(97, 123))
    keys = rand_keys(rng, 3000, alphabet, 1, 12)
    o = orc.AC.compile(keys)
    m = ModelAC(keys)
    for i in (0, 1, 17, 2999):
        assert o.key(i) == keys[i]
    pieces = []
    for _ in range(300):
        if rng.random() < 0.5:
            pieces.append(rng.choice(keys))
        else:
            pieces.append(bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 9))))
    text = b"".join(pieces)
    assert as_list(o.match(text)) == m.match(text)


@pytest.mark.parametrize("seed", range(10))
def test_utf8_string_api_and_sep(seed):
    rng = random.Random(2000 + seed)
    cps = [chr(c) for c in list(range(0x4E00, 0x4E20)) + list(range(97, 105)) + list(range(0x430, 0x438))] + [" "]
    keys = []
    seen = set()
    while len(keys) < 60:
        k = "".join(rng.choice(cps[:-1]) for _ in range(rng.randint(1, 4)))
        if k not in seen:
            seen.add(k)
            keys.append(k)
    text = "".join(rng.choice(cps) for _ in range(300))
    o = orc.AC.compile(keys)
    m = ModelAC(keys)
    assert as_list(o.match(text)) == m.match(text)
    assert as_list(o.match(text.encode())) == m.match(text.encode())
    sep = (256, [32])
    assert as_list(o.match(text, sep=sep)) == m.match(text, sep=sep)
    sep = (100, [32, 97])
    assert as_list(o.match(text.encode(), sep=sep)) == m.match(text.encode(), sep=sep)


def test_nul_bytes_random():
    rng = random.Random(7)
    keys = rand_keys(rng, 30, b"ab", 1, 5)
    o = orc.AC.compile(keys)
    m = ModelAC(keys)
    for _ in range(50):
        text = bytes(rng.choice(b"ab\x00") for _ in range(60))
        assert as_list(o.match(text)) == m.match(text)


def test_batch_matches_per_doc():
    rng = random.Random(9)
    keys = rand_keys(rng, 50, b"abc", 1, 6)
    o = orc.AC.compile(keys)
    m = ModelAC(keys)
    docs = [bytes(rng.choice(b"abc") for _ in range(rng.choice([0, 1, 5, 40, 200]))) for _ in range(30)]
    offs = [0]
    for d in docs:
        offs.append(offs[-1] + len(d))
    hits, dho = o.match_batch(b"".join(docs), offs)
    exp = []
    eoff = [0]
    for d in docs:
        exp += m.match(d)
        eoff.append(len(exp))
    assert as_list(hits) == exp
    assert dho.tolist() == eoff


def test_match_longest_model_agrees_with_oracle():
    """The independent restatement of match_longest (tests/pymodel.py) against the oracle's Cedar-based one.  Cedar's
    stale END flags (cedar.cr:642-648) are observable here; the model takes the set of stale nodes from the oracle
    (they follow from Cedar's slot history, which a dict-trie does not have) and must then agree everywhere."""
    rng = random.Random(99)
    with_stale = 0
    for _ in range(400):
        keys = rand_keys(rng, rng.randint(1, 14), b"abc", 1, 5)
        o = orc.AC.compile(keys)
        stale = o.stale_paths()
        assert len(stale) == o.stale_ends()
        with_stale += bool(stale)
        m = ModelAC(keys)
        for _ in range(6):
            text = bytes(rng.choice(b"abc ") for _ in range(rng.randint(0, 60)))
            for inter in (False, True):
                assert as_list(o.match_longest(text, inter)) == m.match_longest(text, inter, stale=stale), \
                    (keys, text, inter)
    assert with_stale >= 100  # most random small-alphabet automata hold at least one
