"""The character-level image (aha_amd/csrc/unit.hpp) against the oracle, on the CPU: the image is built by the
library (host only), walked by tests/unitsim.py the way scan_unit.hip walks it."""
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC
from test_oracle_vs_model import as_list
from unitsim import UnitSim

CHARS = ["a", "b", "c", "é", "ж", "я", "中", "国", "人", "我", "是", "々", " "]


def rand_word(rng, lo, hi, chars=CHARS[:-1]):
    return "".join(rng.choice(chars) for _ in range(rng.randint(lo, hi)))


def compile_unit(keys, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "unit")  # also below 30 % multi-byte units
    return AC.compile(keys, host_only=True)


@pytest.mark.parametrize("seed", range(12))
def test_unit_image_matches_the_oracle(seed, monkeypatch):
    rng = random.Random(seed)
    keys = sorted({rand_word(rng, 1, rng.choice([2, 4, 7])) for _ in range(rng.choice([3, 40, 400]))})
    ac = compile_unit(keys, monkeypatch)
    assert ac.info["unit_enabled"] == 1
    sim = UnitSim(ac)
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
        assert sim.match(text) == as_list(o.match(text)), (keys, text)


def test_unit_image_reference_kat(monkeypatch):
    ac = compile_unit(["我", "我是", "是中"], monkeypatch)  # spec/ac_spec.cr:5-12
    assert [(e, v) for _, e, v in UnitSim(ac).match("我是中国人".encode())] == [(3, 0), (6, 1), (9, 2)]


def test_truncated_character_at_the_end_of_a_document(monkeypatch):
    keys = ["中", "中国", "a"]
    ac = compile_unit(keys, monkeypatch)
    o = orc.AC.compile(keys)
    sim = UnitSim(ac)
    for text in (b"\xe4\xb8", "中".encode()[:2] + b"a", "a中国".encode()[:-1], "中国".encode() + b"\xe5\x9b", b"\xe4"):
        assert sim.match(text) == as_list(o.match(text))


@pytest.mark.parametrize("keys,why", [
    ([b"\xe4\xb8"], "ends inside"), ([b"a\xb8"], "stray"), ([b"\xf0\x9f\x98\x80"], "0xF0"), ([b"\xe4a"], "without"),
])
def test_ineligible_key_sets_keep_the_byte_level_engines(keys, why, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "unit")
    ac = AC.compile(keys, host_only=True)
    assert ac.info["unit_enabled"] == 0


def test_when_the_image_is_built(monkeypatch):
    monkeypatch.delenv("AHA_ENGINE", raising=False)
    assert AC.compile(["中", "中国"], host_only=True).info["unit_enabled"] == 1       # multi-byte characters: built
    assert AC.compile(["ab", "abcd", "bcdef", "é"], host_only=True).info["unit_enabled"] == 0    # mostly one-byte units: not
    monkeypatch.setenv("AHA_ENGINE", "v2")
    assert AC.compile(["中", "中国"], host_only=True).info["unit_enabled"] == 0       # the byte-level engine by request
    monkeypatch.setenv("AHA_ENGINE", "unit")
    # characters from both ends of the Basic Multilingual Plane: more symbols than the root table holds in LDS
    assert AC.compile(["\u0800a", "\uffeeb"], host_only=True).info["unit_enabled"] == 0


def test_fewer_steps_than_bytes(monkeypatch):
    rng = random.Random(5)
    keys = sorted({rand_word(rng, 2, 5, CHARS[6:12]) for _ in range(300)})
    ac = compile_unit(keys, monkeypatch)
    sim = UnitSim(ac)
    text = "".join(rng.choice(CHARS[6:12]) for _ in range(3000)).encode()
    assert sim.match(text) == as_list(orc.AC.compile(keys).match(text))
    assert sim.trips < len(text) * 0.9  # 3000 characters, a tiny alphabet: many retries from fail states, a header trip for most


def test_image_with_23_bit_bases_matches_the_oracle(monkeypatch):
    """BASELINE cfg 5's million keys: 3.9 M unit states do not fit 22-bit bases, the image takes 23 (six filter bits, unit.hpp
    BASE WIDTH).  The CPU twin walks it over 64 KiB of cfg 5's own hit-dense text against the oracle."""
    from aha_amd import synth
    monkeypatch.delenv("AHA_ENGINE", raising=False)  # the library's own choice
    blob, offs, nf = synth.keys(5)
    corpus, doc = synth.corpus(5, blob, offs, nf, n_bytes=1 << 16, doc_bytes=1 << 16)
    ac = AC.compile_packed(blob, offs, host_only=True)
    info = ac.info
    assert info["unit_enabled"] == 1 and info["unit_base_bits"] == 23 and info["unit_slots"] > 1 << 22
    text = corpus[:int(doc[1])].tobytes()
    o = orc.AC.compile_packed(blob, offs)
    oh, _ = o.match_batch(corpus[:int(doc[1])], np.array([0, int(doc[1])], dtype=np.uint64))
    assert UnitSim(ac).match(text) == [tuple(int(x) for x in h) for h in oh]


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("bits", [22, 23])
def test_big_states_in_both_base_widths(seed, bits, monkeypatch):
    """Hub characters that most keys start with: states with hundreds of transitions over a wide alphabet (the image's big
    states: direct row, group records, child runs), in the 22-bit format and -- AHA_UNIT_BASE_BITS=23 -- in the wide one."""
    rng = random.Random(7000 + seed)
    cps = rng.sample(range(0x4E00, 0x9FA5), rng.choice([60, 300, 1500])) + list(range(0x61, 0x7B)) + \
        rng.sample(range(0x430, 0x450), 8)
    units = [chr(c).encode() for c in cps]
    hubs = rng.sample(units, rng.randint(1, 4))
    keys = sorted({(rng.choice(hubs) if rng.random() < 0.7 else b"") + b"".join(rng.choice(units) for _ in range(rng.randint(1, 4)))
                   for _ in range(rng.choice([2000, 8000]))})
    if bits == 23:
        monkeypatch.setenv("AHA_UNIT_BASE_BITS", "23")
    else:
        monkeypatch.delenv("AHA_UNIT_BASE_BITS", raising=False)
    ac = compile_unit(keys, monkeypatch)
    info = ac.info
    assert info["unit_enabled"] == 1 and info["unit_base_bits"] == bits and info["unit_n_big"] >= 1
    sim = UnitSim(ac)
    o = orc.AC.compile(keys)
    filler = b"".join(units[:40]) + b"\xe4\xb8\xf0 "
    for _ in range(3):
        parts = []
        while sum(map(len, parts)) < 3000:
            r = rng.random()
            if r < 0.4:
                parts.append(rng.choice(keys))
            elif r < 0.7:
                parts.append(rng.choice(hubs) + rng.choice(units))
            else:
                parts.append(bytes(rng.choice(filler) for _ in range(rng.randint(1, 9))))
        text = b"".join(parts)
        assert sim.match(text) == as_list(o.match(text))
