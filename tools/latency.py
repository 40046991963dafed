"""Latency of small host-buffer matches (the reference's usage: one #match per string)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aha_amd import AC, synth
ac = AC.compile(["我", "我是", "是中"])
for _ in range(20): ac.match_array("我是中国人")
t = time.perf_counter(); n = 500
for _ in range(n): ac.match_array("我是中国人")
print("3 keys, 15 B text: %.1f us per match" % ((time.perf_counter() - t) / n * 1e6))
blob, offs, nf = synth.keys(3)
big = AC.compile_packed(blob, offs)
corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 16, doc_bytes=1 << 16)
txt = corpus.tobytes()
for _ in range(20): big.match_array(txt)
t = time.perf_counter(); n = 200
for _ in range(n): big.match_array(txt)
print("100k keys, 64 KiB text: %.1f us per match" % ((time.perf_counter() - t) / n * 1e6))
