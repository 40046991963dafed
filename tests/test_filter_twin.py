"""The prefix-filter engine's formulation (aha_amd/csrc/scan_filter.hip) against the oracle, on the CPU -- tests/filtersim.py
walks the library's own byte-level image from every position its filter lets through and keeps an END step iff no earlier
start's walk is alive at it (src/aha/ac.cr:176-192 says: the state is the longest suffix that is a trie path).  Runs under the
sanitizer build too (tests/test_sanitizers.py)."""
import random

import numpy as np
import pytest

import pyoracle as orc
from aha_amd import AC
from filtersim import FilterSim


def oracle_hits(o, text, doc):
    oh, od = o.match_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(doc, dtype=np.uint64), cap=max(1024, 8 * len(text)))
    return [(d, int(h[0]), int(h[1]), int(h[2])) for d in range(len(doc) - 1) for h in oh[int(od[d]):int(od[d + 1])]]


def keyword_list(rng, n, alphabet, lo=3, hi=9, nested=0.3):
    keys = set()
    while len(keys) < n:
        if keys and rng.random() < nested:  # extensions and suffixes of keys: nested ENDs, fail chains that end in ENDs
            k = rng.choice(sorted(keys))
            k = k + bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 3))) if rng.random() < 0.5 else k[rng.randint(0, 2):]
        else:
            k = bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi)))
        if lo <= len(k) <= 64:
            keys.add(k)
    return sorted(keys)


@pytest.mark.parametrize("seed", range(12))
def test_start_parallel_rule_matches_the_oracle(seed, monkeypatch):
    monkeypatch.setenv("AHA_ENGINE", "filter")
    rng = random.Random(500 + seed)
    alphabet = list(b"abcde") if seed % 3 else list(b"ab") + [0xC3, 0xA9, 0xE4, 0xB8, 0xAD]
    keys = keyword_list(rng, rng.choice([4, 30, 200]), alphabet)
    ac = AC.compile(keys, host_only=True)
    if not ac.info["filter_prefix_bytes"]:
        pytest.skip("no prefix filter for this key set (nesting beyond what kf_walk keeps, or a filter too full)")
    sim = FilterSim(ac, keys)
    o = orc.AC.compile(keys)
    for _ in range(6):
        parts = []
        for _ in range(rng.randint(0, 200)):
            x = rng.random()
            parts.append(rng.choice(keys) if x < 0.3 else bytes([rng.choice(alphabet)]) if x < 0.8 else rng.choice([b" ", b"\0", b"--", b"\xff"]))
        text = b"".join(parts)
        n = len(text)
        doc = sorted({0, n} | {rng.randrange(0, n + 1) for _ in range(rng.choice([0, 2, 9]))}) if n else [0, 0]
        assert sim.match_batch(text, doc) == oracle_hits(o, text, doc), (keys, text, doc)


def test_start_parallel_rule_on_the_truncated_chain():
    """xabc / abc / bcz / c on "xabc" -- the chain stops at the non-END path node "bc"; and a text where an earlier
    start's walk outlives a later start's END (the END is the reference's only if that walk is dead by then)"""
    keys = [b"xabc", b"abc", b"bcz", b"czz"]
    ac = AC.compile(keys, host_only=True)
    sim = FilterSim(ac, keys)
    o = orc.AC.compile(keys)
    for text in (b"xabc", b"xabcz", b"abcbczz", b"xabxabcczz", b"bczabc" * 5):
        assert sim.match_batch(text, [0, len(text)]) == oracle_hits(o, text, [0, len(text)])


def oracle_hits_chars(o, text, doc):
    oh, od = o.match_batch(np.frombuffer(text, dtype=np.uint8), np.asarray(doc, dtype=np.uint64), cap=max(1024, 8 * len(text)), chars=True)
    return [(d, int(h[0]), int(h[1]), int(h[2])) for d in range(len(doc) - 1) for h in oh[int(od[d]):int(od[d + 1])]]


@pytest.mark.parametrize("seed", range(6))
def test_char_offsets_from_the_chunks_continuation_counts(seed, monkeypatch):
    """matcher.cr:34-39 on the prefix-filter engine: what kf_walk<.., CHARS> leaves per chunk (continuation-byte masks and running
    counts, events counted from the document's or the chunk's start, lead_cnt / chunk_doc0 / doc_lead_rank) and what
    k2d_expand<.., true> makes of it, against the oracle's char offsets -- documents cut at character boundaries, several per
    chunk and across chunks, chunks of 256 bytes so that a few KiB of text cross many."""
    monkeypatch.setenv("AHA_ENGINE", "filter")
    rng = random.Random(900 + seed)
    keys = [k.encode() for k in ("abc", "bcd", "naïve", "日本語", "x😀y", "ключ", "éé", "end of line", "abcd")]
    ac = AC.compile(keys, host_only=True)
    sim = FilterSim(ac, keys)
    o = orc.AC.compile(keys)
    for _ in range(3):
        chars = []
        for _ in range(rng.randint(1, 400)):
            x = rng.random()
            chars += list(rng.choice(keys).decode()) if x < 0.3 else [rng.choice("è月😁я -")] * rng.randint(1, 9)
        at = np.cumsum([0] + [len(c.encode()) for c in chars])
        text = "".join(chars).encode()
        cuts = sorted({0, len(chars)} | {rng.randrange(0, len(chars) + 1) for _ in range(rng.choice([0, 3, 40]))})
        doc = [int(at[c]) for c in cuts]
        if rng.random() < 0.5:
            doc = sorted(doc + doc[1:3])  # empty documents
        assert sim.match_batch_chars(text, doc, S=256) == oracle_hits_chars(o, text, doc), (text, doc)

