"""CPU tests of the position-parallel engine's algorithm: the library's exported filter tables interpreted by
tests/ppsim.py, with the exact pass re-stated on the independent model, against the model's sequential automaton
(and through it the oracle, tests/test_oracle_vs_model.py).  No GPU."""
import random

import numpy as np
import pytest

from aha_amd import AC, synth
from ppsim import PpSim
from pymodel import ModelAC
from test_oracle_vs_model import rand_keys


def _keys_ge2(rng, n, alphabet, maxlen):
    ks = set()
    n = min(n, sum(len(alphabet) ** k for k in range(2, maxlen + 1)) // 2)
    while len(ks) < n:
        L = rng.randint(2, maxlen)
        ks.add(bytes(rng.choice(alphabet) for _ in range(L)))
    return sorted(ks)


@pytest.mark.parametrize("seed", range(6))
def test_pp_twin_random_small_alphabet(seed):
    rng = random.Random(1000 + seed)
    alphabet = b"abc" if seed % 2 == 0 else bytes([0x61, 0x62, 0xE4, 0xB8, 0xAD])
    keys = _keys_ge2(rng, rng.randint(1, 40), alphabet, 9)
    ac = AC.compile(keys, host_only=True)
    if not ac.info["pp_enabled"]:
        pytest.skip("preconditions")
    m = ModelAC(keys)
    sim = PpSim(ac, m)
    for _ in range(40):
        n = rng.randint(0, 120)
        t = bytes(rng.choice(alphabet + b" \x00") for _ in range(n))
        assert sim.match(t) == m.match(t, chars=False), (keys, t)


def test_pp_twin_subset_semantics():
    # SURVEY.md section 0.1 counter-examples: the engine must reproduce the truncated output chain
    for keys, text in ((["xabc", "abc", "bcz", "cc"], "xabc"), (["aa", "aaa", "aaaa"], "aaaaaa"),
                       (["ab", "abcd", "bc", "cd"], "abcd abc bcd")):
        ac = AC.compile(keys, host_only=True)
        assert ac.info["pp_enabled"]
        m = ModelAC(keys)
        assert PpSim(ac, m).match(text.encode()) == m.match(text.encode(), chars=False)


def test_pp_disabled_by_preconditions():
    assert not AC.compile(["a", "bc"], host_only=True).info["pp_enabled"]          # 1-byte key
    assert not AC.compile(["ab", "x" * 241], host_only=True).info["pp_enabled"]    # longest key beyond the halo
    assert AC.compile(["ab", "x" * 240], host_only=True).info["pp_enabled"]


def test_pp_twin_cfg3_shape():
    blob, offs, nf = synth.keys(3, K=3000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=6000, doc_bytes=6000)
    keys = [bytes(blob[int(offs[i]):int(offs[i + 1])]) for i in range(offs.size - 1)]
    ac = AC.compile_packed(blob, offs, host_only=True)
    assert ac.info["pp_enabled"]
    m = ModelAC(keys)
    t = corpus.tobytes()
    assert PpSim(ac, m).match(t) == m.match(t, chars=False)
