"""Differential fuzzing of the HIP path against the CPU oracle (test tooling, run on the GPU box):
random automata (small alphabets -> deep fail links, UTF-8-like bytes, nested keys), random batches
(ragged documents, NUL bytes), random image variants (compact/wide, capped LDS prefix, shadow fail
links on/off, two-pass, character-level and prefix-filter engines, the character-level one with the fused and with the
general post passes, the prefix-filter one with every chunk size on keyword-list shaped cases: keys of 3+ bytes, sparse text),
match_longest against the oracle (stale END flags included).  python tools/fuzz_gpu.py [seconds] [seed]"""
import faulthandler, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import pyoracle as orc
from aha_amd import AC, ACGroup
from pymodel import ModelAC

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
ALPHABETS = [b"ab", b"abc", b"abcd", b"ab\xe4\xb8\x80", b"abcdefgh", b"xyz\xd0\xb0\xd1\x8f\xe4\xb8\xad\xe5\x9b\xbd",
             bytes(range(0x61, 0x7b)), bytes(range(1, 256))]
n_cases = n_hits = n_long = n_group = n_filter = n_skip = n_pair = n_filter_chars = 0
seed = seed0
while time.time() < t_end:
    rng = random.Random(seed)
    faulthandler.dump_traceback_later(180, exit=True)  # a case that takes three minutes is a hang: say where
    alpha = rng.choice(ALPHABETS)
    nk = rng.choice([1, 3, 20, 200, 2000, 20000])
    lo, hi = rng.choice([(1, 3), (1, 8), (2, 12), (1, 24)])
    keys, seen = [], set()
    tries = 0
    big = rng.random() < 0.2
    kwl = not big and rng.random() < 0.3  # a keyword list: keys of 3+ bytes, text with few places where a key could start
    if kwl:
        alpha = rng.choice([bytes(range(0x61, 0x7b)), bytes(range(1, 256)), b"abcdefgh", b"ab\xe4\xb8\x80xyz"])
        nk = rng.choice([3, 50, 1000, 1000, 20000])
        lo, hi = rng.choice([(3, 6), (3, 12), (4, 16), (3, 40)])
        while len(keys) < nk and tries < nk * 20:
            tries += 1
            if keys and rng.random() < 0.25:  # extensions and suffixes: several keys end on one walk
                k = rng.choice(keys)
                k = k[rng.randint(0, len(k) - 3):] if rng.random() < 0.4 else k + bytes(rng.choice(alpha) for _ in range(rng.randint(1, 3)))
            else:
                k = bytes(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))
            if len(k) >= 3 and k not in seen and len(k) <= 64:
                seen.add(k)
                keys.append(k)
    if big:  # a wide alphabet of whole characters and a few hub characters that most keys start with: states with hundreds
        # of transitions (the unit image's big states: direct row, group records, child runs) beside ordinary ones
        cps = rng.sample(range(0x4E00, 0x9FA5), rng.choice([60, 300, 1500])) + list(range(0x61, 0x7B)) + \
            rng.sample(range(0x430, 0x450), 8)
        units = [chr(c).encode() for c in cps]
        hubs = rng.sample(units, rng.randint(1, 4))
        nk = rng.choice([200, 2000, 20000])
        while len(keys) < nk and tries < nk * 20:
            tries += 1
            k = (rng.choice(hubs) if rng.random() < 0.7 else b"") + b"".join(rng.choice(units) for _ in range(rng.randint(1, 4)))
            if k not in seen:
                seen.add(k)
                keys.append(k)
        alpha = b"".join(rng.sample(units, min(len(units), 40)) + hubs * 6) + b"\xe4\xb8\xf0 "
    elif not kwl and rng.random() < 0.3:  # whole UTF-8 characters as the alphabet (eligible for the character-level engine); the
        # text below is still cut anywhere and mixed with malformed sequences
        units = [c.encode() for c in rng.sample("abcéжя中国人我是々 ", rng.randint(2, 8))]
        while len(keys) < nk and tries < nk * 20:
            tries += 1
            k = b"".join(rng.choice(units) for _ in range(rng.randint(1, max(1, hi // 2))))
            if k not in seen and len(k) <= 64:
                seen.add(k)
                keys.append(k)
        alpha = b"".join(units) + b"\xe4\xb8\xf0"
    while len(keys) < nk and tries < nk * 20:
        tries += 1
        if keys and rng.random() < 0.3:  # nested / overlapping keys: suffixes and extensions of existing ones
            k = rng.choice(keys)
            k = k[rng.randint(0, len(k) - 1):] if rng.random() < 0.5 else k + bytes([rng.choice(alpha)])
        else:
            k = bytes(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))
        if k and k not in seen and len(k) <= 64:
            seen.add(k)
            keys.append(k)
    env = {"AHA_LDS_SLOTS": rng.choice([None, None, "512", "1024", "4096"]),
           "AHA_SHADOW_FAIL": rng.choice([None, None, None, "0"]),
           "AHA_ENGINE": rng.choice([None, None, "unit", "unit", "skip", "skip", "pair", "pair", "v2", "v1"]),  # (skip / pair: scan_skip.hip / scan_pair.hip where no key is a single character)
           "AHA_UNIT_POST": rng.choice([None, None, "regroup"]),
           "AHA_UNIT_HEADER_BESIDE": rng.choice([None, "0", "1"]),
           "AHA_UNIT_BASE_BITS": rng.choice([None, None, "23"]),
           "AHA_DIRECT": rng.choice([None, None, "0"]),
           "AHA_FILTER_CHUNK": rng.choice([None, "8192", "16384", "32768"])}
    if kwl:
        env["AHA_ENGINE"] = rng.choice([None, "filter"])
        env["AHA_DIRECT"] = None
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    wide = rng.random() < 0.25
    if os.environ.get("AHA_FUZZ_VERBOSE"):
        print("seed", seed, "keys", len(keys), "kwl", kwl, "big", big, "env", env, flush=True)
    ac = AC.compile(keys, force_wide=wide)
    ac.set_profiling(True)
    o = orc.AC.compile(keys)
    grp, n_shards = None, 0
    if not wide and rng.random() < 0.3:  # (the group compiles with the library's own slot format)
        n_shards = rng.choice([2, 3, 4])
        os.environ["AHA_GROUP_RCCL"] = rng.choice(["", "self"])
        if not os.environ["AHA_GROUP_RCCL"]:
            del os.environ["AHA_GROUP_RCCL"]
        grp = ACGroup.compile(keys, [0] * n_shards)
    for _ in range(rng.randint(1, 3)):
        n = rng.choice([0, 1, 17, 1000, 20000, 300000] + ([2000000] if kwl else []))
        parts = []
        p_key = rng.choice([0.02, 0.1]) if kwl else 0.4
        have = 0
        while have < n:
            r = rng.random()
            if kwl and r >= p_key:  # filler: mostly bytes no key starts with, now and then the alphabet's own
                parts.append(bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) if r < p_key + 0.05
                             else rng.choice([b" ", b"-- ", b"\n", b"0123456789 ", b"\x00", "\u00e8".encode(), "\u6708".encode(),
                                              "\U0001f601".encode(), b"\x80"]) * rng.randint(1, 12))  # (characters of 2, 3, 4 bytes and stray continuation bytes: char offsets on the prefix-filter engine)
            elif r < p_key and keys:
                parts.append(rng.choice(keys))
            elif big and r < 0.7:  # characters (not bytes) of the wide alphabet: hub + any
                parts.append(rng.choice(hubs) + rng.choice(units))
            elif r < 0.45:
                parts.append(b"\x00")
            else:
                parts.append(bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))))
            have += len(parts[-1])
        text = np.frombuffer(b"".join(parts)[:n] if n else b"", dtype=np.uint8)
        cuts = sorted(set([0, text.size] + [rng.randint(0, text.size) for _ in range(rng.choice([0, 1, 5, 60]))]))
        if rng.random() < 0.3:
            cuts = sorted(cuts + cuts[1:3])  # empty documents
        doc = np.array(cuts, dtype=np.uint64)
        chars = rng.random() < 0.3  # char offsets: both sides count the bytes outside 0x80..0xBF, also in malformed text
        gh, gd = ac.match_batch(text, doc, chars=chars)
        oh, od = o.match_batch(text, doc, chars=chars)
        ok = len(gh) == len(oh) and np.array_equal(np.asarray(gh).view(np.int32), np.asarray(oh).view(np.int32)) and \
            np.array_equal(np.asarray(gd, dtype=np.uint64), np.asarray(od, dtype=np.uint64))
        if not ok:
            print("MISMATCH seed", seed, "alphabet", alpha[:8], "keys", len(keys), "env", env, "wide", wide,
                  "n", text.size, "docs", doc.size - 1, "chars", chars, "hits gpu/oracle", len(gh), len(oh), flush=True)
            sys.exit(1)
        n_cases += 1
        n_hits += len(gh)
        n_filter += 1 if text.size and ac.last_timing()["engine"] == 5 else 0
        n_skip += 1 if text.size and ac.last_timing()["engine"] == 6 else 0
        n_pair += 1 if text.size and ac.last_timing()["engine"] == 7 else 0
        n_filter_chars += 1 if text.size and chars and ac.last_timing()["engine"] == 5 and int(text.max()) >= 0x80 else 0
        if grp is not None and text.size <= 300000:
            # the group API over shards on this one device: partition, shards in turn through the pipelined host entry, the
            # 4-byte exchange stream (or triples), the rebuild -- the caller's copy and every shard's gathered copy
            gg, gdo = grp.match_batch(text, doc, chars=chars)
            ok = len(gg) == len(oh) and np.asarray(gg).tobytes() == np.asarray(gh).tobytes() and \
                np.array_equal(np.asarray(gdo, dtype=np.uint64), np.asarray(od, dtype=np.uint64))
            for shard in range(n_shards):
                ok = ok and grp.download_shard(shard).tobytes() == np.asarray(gh).tobytes()
            if ok and rng.random() < 0.5:  # the resident entry: the ranges stay on the device, the hits too
                res = grp.upload_corpus(text, doc)
                rn, rdo = grp.match_corpus(res, chars=chars)
                ok = rn == len(oh) and np.array_equal(np.asarray(rdo, dtype=np.uint64), np.asarray(od, dtype=np.uint64))
                for shard in range(n_shards):
                    ok = ok and grp.download_shard(shard).tobytes() == np.asarray(gh).tobytes()
                del res
            if not ok:
                print("GROUP MISMATCH seed", seed, "keys", len(keys), "env", env, "shards", n_shards, "n", text.size,
                      "docs", doc.size - 1, "chars", chars, flush=True)
                sys.exit(1)
            n_group += 1
        if text.size <= 20000 and len(keys) <= 2000 and rng.random() < 0.5:
            # match_longest against the oracle (Cedar's stale END flags included) and against the independent restatement
            # (tests/pymodel.py) given the oracle's stale paths; first document of the batch
            m = ModelAC(keys)
            stale = o.stale_paths()
            d0 = bytes(text[int(doc[0]):int(doc[1])]) if doc.size > 1 else b""
            for inter in (False, True):
                got = [(h.start, h.end, h.value) for h in ac.match_longest(d0, inter)]
                want = [(int(h["start"]), int(h["end"]), int(h["value"])) for h in o.match_longest(d0, inter)]
                if got != want or got != m.match_longest(d0, inter, stale=stale):
                    print("LONGEST MISMATCH seed", seed, "keys", len(keys), "env", env, "wide", wide, "n", len(d0),
                          "intersectable", inter, flush=True)
                    sys.exit(1)
            n_long += 1
    seed += 1
    if seed % 5 == 0:
        print(f"[fuzz] {seed - seed0} automata, {n_cases} batches, {n_hits} hits ok", flush=True)
print(f"fuzz ok: {n_filter} batches on the prefix-filter engine, {n_filter_chars} of them with char offsets over non-ASCII text, {n_skip} on the skip-ahead traversal, {n_pair} on the pair engine, {n_group} group batches, {n_long} match_longest documents, {n_cases} batches over {seed - seed0} automata, {n_hits} hits compared, seeds {seed0}..{seed - 1}")
