"""Experiment: how much of k2_traverse's time is the deep (cold) walk of key tokens?
Same cfg-3 keys, corpora with and without key tokens, and a smaller key set."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aha_amd import AC, synth

dev = torch.device("cuda:0")
NB = 1 << 28


def run(tag, kb, ko, cb, co, cnf):
    corpus, doc = synth.corpus(3, cb, co, cnf, n_bytes=NB)
    ac = AC.compile_packed(kb, ko, device=0)
    ac.set_profiling(True)
    dc = torch.from_numpy(corpus).to(dev)
    dd = torch.from_numpy(doc.astype(np.int64)).to(dev)
    out = torch.empty((40_000_000, 3), dtype=torch.int32, device=dev)
    for _ in range(3):
        n = ac.match_batch_device(dc, dd, out)
    torch.cuda.synchronize()
    t = ac.last_timing()
    info = ac.info
    print(tag, "slots", info["n_slots"], "lds", info["lds_slots"], "hits", n, "trav ms %.3f -> %.1f GB/s" %
          (t["ms_count"], NB / t["ms_count"] / 1e6), flush=True)


kb, ko, nf = synth.keys(3)
ob, oo, onf = synth.keys(3, seed=0x5EED1234)
run("100k keys, own key tokens   ", kb, ko, kb, ko, nf)
run("100k keys, foreign tokens   ", kb, ko, ob, oo, onf)
for K in (50000, 30000, 10000):
    sb, so, snf = synth.keys(3, K=K)
    run("%6d keys, own key tokens " % K, sb, so, sb, so, snf)
    run("%6d keys, foreign tokens " % K, sb, so, ob, oo, onf)
