"""Replays one seed of tools/fuzz_gpu.py and says which of the three match_longest answers differ (GPU library, oracle,
independent model with the oracle's stale paths).  python tools/fuzz_repro.py <seed>"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
src = open(os.path.join(ROOT, "tools", "fuzz_gpu.py")).read()
seed = int(sys.argv[1])
# run the fuzzer's own loop body for exactly this seed, with the mismatch handler replaced by a verbose one
src = src.replace('budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0', 'budget = 1e9')
src = src.replace('seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1', f'seed0 = {seed}')
src = src.replace('while time.time() < t_end:', 'while seed == seed0:')
src = src.replace('''                if got != want or got != m.match_longest(d0, inter, stale=stale):''',
                  '''                mm = m.match_longest(d0, inter, stale=stale)
                print("inter", inter, "gpu==oracle", got == want, "model==oracle", mm == want, "stale", len(stale), "hits", len(want), flush=True)
                if got != want:
                    for i, (a, b) in enumerate(zip(got + [None] * 3, want + [None] * 3)):
                        if a != b:
                            print("first difference at", i, "gpu", got[max(0, i - 2):i + 3], "oracle", want[max(0, i - 2):i + 3])
                            print("keys", [keys[x[2]] for x in want[max(0, i - 2):i + 3] if x], "text", d0[max(0, (b or a)[0] - 8):(b or a)[1] + 8])
                            break
                    print("stale paths", sorted(stale)[:20])
                if False:''')
exec(compile(src, "fuzz_gpu_repro", "exec"))
