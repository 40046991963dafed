"""The pair engine (aha_amd/csrc/scan_pair.hip; unit.hpp, PAIR TABLE) against the oracle, on the CPU: filter, pair table and
unit image are built by the library (host only), tests/pairsim.py runs the stateless pair pass, the deep walks, the voiding and
the filling the way the kernels do."""
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC
from pairsim import PairSim
from test_skip_twin import BAD, CHARS, oracle_hits, rand_docs, rand_text, rand_word


def compile_pair(keys, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "pair")
    return AC.compile(keys, host_only=True)


@pytest.mark.parametrize("seed", range(16))
def test_pair_engine_matches_the_oracle(seed, monkeypatch):
    rng = random.Random(2000 + seed)
    keys = sorted({rand_word(rng, 2, rng.choice([2, 4, 7])) for _ in range(rng.choice([3, 40, 400]))})
    ac = compile_pair(keys, monkeypatch)
    if not ac.info["pair_engine"]:
        pytest.skip("more than three deep END states on one trie path")
    sim = PairSim(ac)
    o = orc.AC.compile(keys)
    for _ in range(5):
        text = rand_text(rng, keys, rng.randint(0, 400), rng.choice([0.05, 0.3, 0.7]))
        doc = rand_docs(rng, len(text))
        assert sim.match_batch(text, doc) == oracle_hits(o, text, doc), (keys, text, doc)


def test_pair_engine_overlapping_deep_walks(monkeypatch):
    """deep walks that start inside one another: the later walk's END states count only behind the earlier walk's reach; the
    events a walk covers are void, those behind its reach are not"""
    keys = ["abcde", "bcdxy", "cdx", "dxyz", "ab", "bc", "cd", "dx", "xy", "yz", "zz"]
    ac = compile_pair(keys, monkeypatch)
    assert ac.info["pair_engine"] == 1
    sim = PairSim(ac)
    o = orc.AC.compile(keys)
    for text in (b"abcdxyzz", b"abcdexy", b"xabcdxyzzab", b"abcabcdxabcdxyz" * 9, b"bcdxbcdxyzzabcde"):
        assert sim.match_batch(text, [0, len(text)]) == oracle_hits(o, text, [0, len(text)]), text
    rng = random.Random(4)
    for _ in range(20):
        text = "".join(rng.choice("abcdxyz ") for _ in range(rng.randint(1, 300))).encode()
        doc = rand_docs(rng, len(text))
        assert sim.match_batch(text, doc) == oracle_hits(o, text, doc), (text, doc)


def test_pair_engine_on_the_headline_shape(monkeypatch):
    from aha_amd import synth
    monkeypatch.delenv("AHA_ENGINE", raising=False)
    monkeypatch.setenv("AHA_PAIR", "1")
    blob, offs, nf = synth.keys(3, K=3000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 15, doc_bytes=1 << 13)
    ac = AC.compile_packed(blob, offs, host_only=True)
    assert ac.info["pair_engine"] == 1
    sim = PairSim(ac)
    st = {}
    text = corpus.tobytes()
    got = sim.match_batch(text, doc, stats=st)
    assert got == oracle_hits(orc.AC.compile_packed(blob, offs), text, doc)
    assert st["cands"] < 0.05 * len(text), st  # the deep candidates are a few per cent of the positions
