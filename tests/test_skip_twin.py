"""The skip-ahead traversal (aha_amd/csrc/scan_skip.hip; unit.hpp, MARKS) against the oracle, on the CPU: the mark filter
and the unit image are built by the library (host only), tests/skipsim.py marks and walks the text the way ks_mark and
ks_traverse do -- jumps, stays, early fails, pseudo jumps at the end of a lane's marks, documents, chunks with warm-up."""
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC
from skipsim import SkipSim

CHARS = ["a", "b", "c", "é", "ж", "я", "中", "国", "人", "我", "是", "々", " "]
BAD = [b"\xe4", b"\xe4\xb8", b"\xb8", b"\xad\xad", b"\xf0\x9f\x98\x80", b"\x00", b"\xc3", b"\xff", b"\xe4\xe4\xb8\xad"]


def rand_word(rng, lo, hi, chars=CHARS[:-1]):
    return "".join(rng.choice(chars) for _ in range(rng.randint(lo, hi)))


def compile_skip(keys, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "skip")  # the unit image also below 30 % multi-byte units, the marks whatever their fill
    return AC.compile(keys, host_only=True)


def oracle_hits(o, text, doc):
    oh, od = o.match_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(doc, dtype=np.uint64), cap=max(1024, 4 * len(text)))
    out = []
    for d in range(len(doc) - 1):
        for h in oh[int(od[d]):int(od[d + 1])]:
            out.append((d, int(h[0]), int(h[1]), int(h[2])))
    return out


def rand_text(rng, keys, n_parts, p_key=0.3):
    parts = []
    for _ in range(n_parts):
        x = rng.random()
        if x < p_key:
            parts.append(rng.choice(keys).encode())
        elif x < 0.85:
            parts.append(rng.choice(CHARS).encode())
        elif x < 0.93:
            parts.append(rng.choice(BAD))
        else:
            parts.append(bytes([rng.randrange(1, 256)]))
    return b"".join(parts)


def rand_docs(rng, n):
    cuts = sorted({0, n} | {rng.randrange(0, n + 1) for _ in range(rng.choice([0, 1, 3, 12]))}) if n else [0, 0]
    if rng.random() < 0.3 and n:  # empty documents
        cuts = sorted(cuts + [rng.choice(cuts)])
    return cuts


@pytest.mark.parametrize("seed", range(16))
def test_skip_walk_matches_the_oracle(seed, monkeypatch):
    rng = random.Random(1000 + seed)
    keys = sorted({rand_word(rng, 2, rng.choice([2, 4, 7])) for _ in range(rng.choice([3, 40, 400]))})
    ac = compile_skip(keys, monkeypatch)
    assert ac.info["unit_enabled"] == 1 and ac.info["skip_filter_words"] >= 1024
    sim = SkipSim(ac)
    o = orc.AC.compile(keys)
    for _ in range(5):
        text = rand_text(rng, keys, rng.randint(0, 400), rng.choice([0.05, 0.3, 0.7]))
        doc = rand_docs(rng, len(text))
        for S in (64, 256, 4096):
            assert sim.match_batch(text, doc, S=S) == oracle_hits(o, text, doc), (keys, text, doc, S)


def test_skip_walk_reference_kat(monkeypatch):
    keys = ["我是", "是中", "中国人"]  # (the reference's own KAT holds a one-character key: that key set keeps engine 4)
    ac = compile_skip(keys, monkeypatch)
    text = "我是中国人".encode()
    assert SkipSim(ac).match_batch(text, [0, len(text)]) == oracle_hits(orc.AC.compile(keys), text, [0, len(text)])
    assert AC.compile(["我", "我是", "是中"], host_only=True).info["skip_filter_words"] == 0  # spec/ac_spec.cr:5-12: no marks
    monkeypatch.delenv("AHA_ENGINE", raising=False)
    assert AC.compile(keys, host_only=True).info["skip_filter_words"] == 0  # opt-in: nothing gets it by default


def test_skip_walk_long_sparse_text_and_window_ends(monkeypatch):
    """marks hundreds of bytes apart: the walk moves by pseudo jumps (no mark in a lane's two words), also across chunks and
    documents; keys that straddle the 124-byte limit of a window and a chunk's end"""
    rng = random.Random(7)
    keys = ["中国人", "ab", "abc", "bcab", "我是", "是中国"]
    ac = compile_skip(keys, monkeypatch)
    sim = SkipSim(ac)
    o = orc.AC.compile(keys)
    for _ in range(12):
        parts = []
        for _ in range(40):
            parts.append((" " * rng.randint(0, 300) + rng.choice("xyz々") * rng.randint(0, 40)).encode())
            parts.append(rng.choice(keys).encode() * rng.randint(1, 3))
        text = b"".join(parts)
        doc = rand_docs(rng, len(text))
        for S in (64, 128, 4096):
            assert sim.match_batch(text, doc, S=S) == oracle_hits(o, text, doc)


def test_skip_trip_count_on_the_headline_shape(monkeypatch):
    """cfg 3's shape at a small scale: the walk takes a fraction of the trips of one trip per character"""
    from aha_amd import synth
    monkeypatch.delenv("AHA_ENGINE", raising=False)
    monkeypatch.setenv("AHA_SKIP", "1")  # (the engine is opt-in: AHA_ENGINE=skip, or AHA_SKIP=1 beside the library's own choice)
    blob, offs, nf = synth.keys(3, K=3000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 15, doc_bytes=1 << 13)
    ac = AC.compile_packed(blob, offs, host_only=True)
    assert ac.info["skip_filter_words"] > 0
    sim = SkipSim(ac)
    trips = [0]
    text = corpus.tobytes()
    got = sim.match_batch(text, doc, trips=trips)
    o = orc.AC.compile_packed(blob, offs)
    assert got == oracle_hits(o, text, doc)
    chars = sum(1 for b in text if (b & 0xC0) != 0x80)
    assert trips[0] < 0.5 * chars, (trips[0], chars)
