"""world_size-2 gloo tests (CPU) of the multi-GPU sharding logic: contiguous
document partition + all-gatherv of hit buffers reproduces the single-process
result.  The per-rank matcher here is the CPU oracle (checker only) because
there is no GPU in this environment; on the GPU box bench.py --gpus N runs the
same HitGatherer over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, keys_blob, keys_offs, corpus, doc, out_q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pyoracle as orc
    from aha_amd.distributed import HitGatherer, local_shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = orc.AC.compile_packed(keys_blob, keys_offs)
        sub, sub_doc, lo = local_shard(corpus, doc, rank, world)
        hits, dho = o.match_batch(sub, sub_doc)
        t = torch.from_numpy(hits.view(np.int32).reshape(-1, 3).copy()) if len(hits) else torch.zeros((0, 3), dtype=torch.int32)
        g = HitGatherer(dist, torch.device("cpu"))
        allh, counts = g.all_gatherv(t, len(hits))
        # overlapped form: two exchanges in flight on the two slots
        g.start(t, len(hits), 0)
        g.start(t, len(hits), 1)
        for slot in (0, 1):
            oh, oc = g.finish(slot)
            assert oc == counts and torch.equal(oh, allh)
        assert g.finish(0) is None
        # packed exchange ({end, value} pairs on the wire, triples rebuilt on arrival): same result
        from aha_amd import AC

        pac = AC.compile_packed(keys_blob, keys_offs, host_only=True)
        gp = HitGatherer(dist, torch.device("cpu"), ac=pac, packed=True)
        ph, pc = gp.all_gatherv(t, len(hits))
        assert pc == counts and torch.equal(ph, allh)
        gp.start(t, len(hits), 0)
        gp.start(t, len(hits), 1)
        for slot in (0, 1):
            oh, oc = gp.finish(slot)
            assert oc == counts and torch.equal(oh, allh)
        # 4-byte exchange stream (value << 12 | step of `end`, exceptions aside): same result
        gw = HitGatherer(dist, torch.device("cpu"), ac=pac, exchange="words")
        wh, wc = gw.all_gatherv(t, len(hits))
        assert wc == counts and torch.equal(wh, allh)
        assert gw.last_payload_elems <= len(hits) + len(hits) // 8 + 64  # about 4 bytes per hit on the wire
        gw.start(t, len(hits), 0)
        gw.start(t, len(hits), 1)
        for slot in (0, 1):
            oh, oc = gw.finish(slot)
            assert oc == counts and torch.equal(oh, allh)
        alld = g.gather_doc_hit_offsets(torch.from_numpy(dho.astype(np.int64)), counts)
        # the strong-scaling leg of bench.py, step for step (the matcher is the oracle here, the GPU there)
        from aha_amd.distributed import stream_digest, strong_scaling_pass

        def match_fn(c, d):
            return t, len(hits), torch.from_numpy(dho.astype(np.int64))

        sh, sd, secs = strong_scaling_pass(g, match_fn, sub, sub_doc)
        assert torch.equal(sh, allh) and torch.equal(sd, alld) and len(secs) == 2
        full_h, full_d = o.match_batch(corpus, doc)
        assert stream_digest(sh.numpy(), sd.numpy()) == stream_digest(full_h.view(np.int32).reshape(-1, 3), full_d)
        out_q.put((rank, allh.numpy().copy(), alld.numpy().copy(), counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgatherv_matches_single_process(world):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    from aha_amd import synth

    blob, offs, nf = synth.keys(3, K=3000)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=1 << 18, doc_bytes=1 << 13)
    # a few empty documents, including at the ends
    doc = np.concatenate([[0], doc[:5], [doc[4]], doc[5:], [doc[-1]]]).astype(np.uint64)
    o = orc.AC.compile_packed(blob, offs)
    ref_hits, ref_dho = o.match_batch(corpus, doc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, blob, offs, corpus, doc, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = ref_hits.view(np.int32).reshape(-1, 3)
    for rank, allh, alld, counts in res:
        assert sum(counts) == len(ref_hits)
        assert np.array_equal(allh, ref), f"rank {rank}: gathered hits differ"
        assert np.array_equal(alld.astype(np.uint64), ref_dho), f"rank {rank}: doc offsets differ"


def test_pack4_host_round_trip_and_format():
    """The CPU restatement of the 4-byte exchange stream (the GPU kernels are compared with it in the GPU suite)."""
    from aha_amd.distributed import PK4_BLOCK, pack4_host, unpack4_host

    rng = np.random.default_rng(3)
    klen = torch.from_numpy(rng.integers(1, 25, size=1 << 20).astype(np.int32))
    for n, max_step in ((0, 10), (1, 10), (1023, 50), (1024, 50), (1025, 50), (5000, 9000), (3000, 4100)):
        docs = rng.integers(0, 2, size=n).cumsum()  # a document change resets `end`
        step = rng.integers(0, max_step, size=n)
        end = np.zeros(n, dtype=np.int64)
        run = 0
        for i in range(n):
            run = step[i] + 1 if (i == 0 or docs[i] != docs[i - 1]) else run + step[i]
            end[i] = run + 30
        value = rng.integers(0, 1 << 20, size=n)
        hits = torch.zeros((n, 3), dtype=torch.int32)
        hits[:, 1] = torch.from_numpy(end.astype(np.int32))
        hits[:, 2] = torch.from_numpy(value.astype(np.int32))
        hits[:, 0] = hits[:, 1] - klen[hits[:, 2].long()]
        for fmt in ((12, 0), (7, 5)):  # key length looked up on arrival / carried in the word (20 + 5 + 7 bits)
            w = pack4_host(hits, fmt)
            nb = (n + PK4_BLOCK - 1) // PK4_BLOCK
            assert w.dtype == torch.int32 and n + nb <= w.numel() <= 2 * n + nb
            if n:
                first = w[n:n + nb].tolist()
                assert first[0] == 0 and first == sorted(first)
            assert torch.equal(unpack4_host(w, n, klen if fmt[1] == 0 else None, fmt), hits)


def test_partition_is_contiguous_and_balanced():
    from aha_amd.distributed import partition_docs

    rng = np.random.default_rng(0)
    sizes = rng.integers(0, 5000, size=1000)
    doc = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    for world in (1, 2, 4, 8):
        parts = partition_docs(doc, world)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        for (a, b), (c, d) in zip(parts, parts[1:]):
            assert b == c and a <= b
        loads = [int(doc[b] - doc[a]) for a, b in parts]
        assert max(loads) - min(loads) <= 2 * 5000
    # degenerate: fewer documents than ranks
    parts = partition_docs(np.array([0, 10], dtype=np.uint64), 4)
    assert parts[0][0] == 0 and parts[-1][1] == 1 and sum(b - a for a, b in parts) == 1
