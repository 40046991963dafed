"""Multi-GPU sharding of Aha::AC#match (SURVEY.md section 8 e).

The path shards naturally: the match state resets per sequence
(src/aha/ac.cr:177) and the automaton is read-only, so documents are
independent units.  One process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests):

  * partition_docs   -- contiguous document ranges balanced by cumulative
                        bytes; contiguity makes global hit order = rank order,
                        so no merge is needed;
  * HitGatherer      -- all-gatherv of the variable-length hit buffers:
                        all_gather of the per-rank counts, then ONE grouped
                        all-pairs isend/irecv exchange (ncclGroupStart ..
                        Send/Recv .. GroupEnd under RCCL).  On MI355X's fully
                        connected xGMI mesh every rank then drives its 7 links
                        at once (time ~ 12*max_r H_r / 153 GB/s), where a ring
                        all-gather would be per-link bound.  exchange="pairs"
                        puts {end, value} pairs on the links (8 B per hit; the
                        start is end - key length), exchange="words" a 4-byte
                        stream (value | step of end); the triples are rebuilt
                        on arrival.

The automaton is replicated: every rank compiles the same keys (~0.2 s for
100k keys), which is cheaper than broadcasting and needs no collective.
"""
import numpy as np
import torch


def partition_docs(doc_offsets, world):
    """Splits D documents into `world` contiguous ranges balanced by bytes.
    Returns a list of (d_lo, d_hi) with d_lo <= d_hi, covering [0, D]."""
    doc_offsets = np.asarray(doc_offsets, dtype=np.uint64)
    D = doc_offsets.size - 1
    total = int(doc_offsets[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        d = int(np.searchsorted(doc_offsets, target, side="left"))
        # nearest document boundary to the byte target
        if d > 0 and d <= D and abs(int(doc_offsets[d - 1]) - target) < abs(int(doc_offsets[min(d, D)]) - target):
            d -= 1
        d = min(max(d, cuts[-1]), D)
        cuts.append(d)
    cuts.append(D)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def local_shard(corpus, doc_offsets, rank, world):
    """This rank's documents: (corpus view, rebased doc offsets, first doc index)."""
    lo, hi = partition_docs(doc_offsets, world)[rank]
    doc_offsets = np.asarray(doc_offsets, dtype=np.uint64)
    b0, b1 = int(doc_offsets[lo]), int(doc_offsets[hi])
    return corpus[b0:b1], (doc_offsets[lo:hi + 1] - doc_offsets[lo]).astype(np.uint64), lo


PK4_BLOCK = 1024


def pack4_host(hits, fmt=(12, 0)):
    """CPU restatement of the 4-byte exchange stream (include/aha_hip.h, aha_ac_hits_pack4_device) for the gloo
    rehearsal: hits [n,3] int32 CPU tensor -> int32 tensor words[n] . first_exception[nb] . exception_end[...].
    fmt = (step_bits, len_bits) of the automaton (AC.stream_format)."""
    sb, lb = fmt
    PK4_EXC = (1 << sb) - 1
    n = int(hits.shape[0])
    nb = (n + PK4_BLOCK - 1) // PK4_BLOCK
    if n == 0:
        return torch.zeros(0, dtype=torch.int32)
    end = hits[:, 1].to(torch.int64)
    step = torch.zeros(n, dtype=torch.int64)
    step[1:] = end[1:] - end[:-1]
    exc = (step < 0) | (step >= PK4_EXC)
    exc[::PK4_BLOCK] = True
    step = torch.where(exc, torch.full_like(step, PK4_EXC), step)
    ln = (end - hits[:, 0].to(torch.int64)) if lb else torch.zeros(n, dtype=torch.int64)
    words = (hits[:, 2].to(torch.int64) << (sb + lb)) | (ln << sb) | step
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    rank = torch.cumsum(exc.to(torch.int64), 0) - exc.to(torch.int64)
    blk = rank[::PK4_BLOCK].to(torch.int32)
    assert blk.numel() == nb
    return torch.cat([words, blk, hits[:, 1][exc].to(torch.int32)])


def unpack4_host(stream, n, key_len, fmt=(12, 0)):
    """Inverse of pack4_host: -> hits [n,3] int32 (start = end - key length: from the word, or from key_len[value])."""
    sb, lb = fmt
    PK4_EXC = (1 << sb) - 1
    out = torch.zeros((n, 3), dtype=torch.int32)
    if n == 0:
        return out
    nb = (n + PK4_BLOCK - 1) // PK4_BLOCK
    words = stream[:n].to(torch.int64) & 0xFFFFFFFF
    exc_end = stream[n + nb:].to(torch.int64)
    step = words & PK4_EXC
    value = words >> (sb + lb)
    exc = step == PK4_EXC
    x = step.clone()
    x[exc] = exc_end[: int(exc.sum())]
    # segmented running sum: an exception restarts it
    tot = torch.cumsum(x, 0)
    seg = torch.cumsum(exc.to(torch.int64), 0) - 1
    head_tot = (tot - x)[exc]  # running total in front of every exception
    end = tot - head_tot[seg]
    out[:, 1] = end.to(torch.int32)
    out[:, 2] = value.to(torch.int32)
    ln = ((words >> sb) & ((1 << lb) - 1)) if lb else key_len[value].to(torch.int64)
    out[:, 0] = (end - ln).to(torch.int32)
    return out


class HitGatherer:
    """all-gatherv of hit triples ([n,3] int32 tensors) across ranks.

    What travels on the links is chosen with `exchange` (needs `ac`, the rank's automaton -- replicated, so every
    rank holds the same key lengths -- for anything but "triples"):
      "triples"  the 12-byte Hit triples as they are;
      "pairs"    {end, value}, 8 bytes per hit: Hit#start = Hit#end - len(key[value]) (src/aha/ac.cr:270-272);
      "words"    the 4-byte stream of include/aha_hip.h (value | key length | step of `end` in one word, field widths
                 from the automaton: AC.stream_format).
    The triples are rebuilt on arrival (aha_ac_hits_*_device on device tensors; plain torch on CPU tensors, i.e. in
    the gloo rehearsal).  xGMI links are the scarce resource of the exchange (one link per peer: ~60 GB/s each way
    against 8 TB/s of HBM), so bytes are traded for two small kernels.  `packed=True` is the old name of "pairs"."""

    def __init__(self, dist, device, group=None, ac=None, packed=False, chars=False, exchange=None):
        self.dist = dist
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.exchange = exchange or ("pairs" if packed else "triples")
        if self.exchange not in ("triples", "pairs", "words"):
            raise ValueError("exchange: triples, pairs or words")
        self._counts = torch.zeros(self.world * 2, dtype=torch.int64, device=device)
        self._mine = torch.zeros(2, dtype=torch.int64, device=device)
        self.packed = self.exchange != "triples"
        self.chars = bool(chars)
        self.ac = ac
        self._cuda = torch.device(device).type == "cuda"
        self._pending = {}
        self._bufs = {}
        self._send = {}
        self._land = {}
        self.last_payload_elems = 0  # int32 elements this rank sent to each peer in the last exchange
        if self.packed:
            if ac is None:
                raise ValueError("a packed exchange needs the automaton (key lengths)")
            self._fmt = ac.stream_format() if self.exchange == "words" else (12, 0)
            if self.exchange == "words" and self._fmt[1] == 0 and ac.n_keys > (1 << 20):
                raise ValueError("the 4-byte exchange stream holds key ids below 2^20")
            self._klen = None if self._cuda else torch.from_numpy(ac.key_lengths(chars))
            # the stream carries no format tag: every rank must pack and unpack with the same field widths, i.e. run the same
            # library build over the same keys -- agreed once, before the first payload travels
            from . import _native as _N
            mine = torch.tensor([int(_N.lib().aha_abi_version()), int(self._fmt[0]), int(self._fmt[1]), int(ac.n_keys)],
                                dtype=torch.int64, device=device)
            every = torch.zeros(self.world * 4, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(every, mine, group=group)
            if not bool((every.view(self.world, 4) == mine).all()):
                raise RuntimeError(f"ranks disagree on ABI / exchange stream format / key count: {every.view(self.world, 4).tolist()}")

    # -- payload: what this rank sends to every peer -----------------------------------
    def _payload(self, hits, n, slot):
        """Returns (flat int32 payload tensor, its length in int32 elements).  For "words" the length is only known
        after the pack kernels ran: it is read from the device together with the counts exchange."""
        if self.exchange == "triples":
            return hits.view(-1), 3 * n
        need = 2 * n + (n + PK4_BLOCK - 1) // PK4_BLOCK + 16  # pairs: 2n; words: 2n + ceil(n/1024) in the worst case
        send = self._send.get(slot)
        if send is None or send.numel() < need:
            send = torch.empty(need + need // 8, dtype=torch.int32, device=hits.device)
            self._send[slot] = send
        if self.exchange == "pairs":
            if n:
                if hits.is_cuda:
                    self.ac.hits_pack_device(hits, n, send)
                else:
                    send[:2 * n].view(n, 2).copy_(hits[:n, 1:3])
            return send, 2 * n
        if hits.is_cuda:
            self.ac.hits_pack4_device(hits, n, send, self._mine[1:2])
            return send, None  # length in self._mine[1] (device)
        w = pack4_host(hits[:n], self._fmt)
        send[:w.numel()].copy_(w)
        return send, int(w.numel())

    def _rebuild(self, land, elems_lo, n, out_rows):
        """The payload of one peer (land[elems_lo:...], n hits) -> triples in out_rows ([n,3] view)."""
        if n <= 0:
            return
        if self.exchange == "pairs":
            pairs = land[elems_lo:elems_lo + 2 * n]
            if pairs.is_cuda:
                self.ac.hits_unpack_device(pairs, n, out_rows, chars=self.chars)
            else:
                p = pairs.view(n, 2)
                out_rows[:, 1:3] = p
                out_rows[:, 0] = p[:, 0] - self._klen[p[:, 1].long()]
        else:
            if land.is_cuda:
                self.ac.hits_unpack4_device(land[elems_lo:], n, out_rows, chars=self.chars)
            else:
                out_rows.copy_(unpack4_host(land[elems_lo:], n, self._klen, self._fmt))

    def all_gatherv(self, hits, n):
        """hits: [cap,3] int32 on self.device, first n rows valid.  Returns
        (gathered [sum_n,3] view, counts list); rank r's hits start at
        sum(counts[:r]) -- global order = rank order (contiguous doc ranges)."""
        self.start(hits, n, slot=2)
        return self.finish(2)

    # -- overlapped form: the exchange of step i runs beside the match of step i+1 ------
    def start(self, hits, n, slot=0):
        """Issues the all-gatherv of hits[:n] without waiting for the payload
        (the 16-byte counts exchange is synchronous).  Uses output buffer
        `slot` so that two exchanges can be in flight; the caller must not
        overwrite hits[:n] before finish(slot).  Returns the counts."""
        dist = self.dist
        self.finish(slot)
        payload, elems = self._payload(hits, n, slot)
        self._mine[0] = n
        if elems is not None:
            self._mine[1] = elems
        dist.all_gather_into_tensor(self._counts, self._mine, group=self.group)
        both = [int(c) for c in self._counts.tolist()]
        counts, sizes = both[0::2], both[1::2]
        total = sum(counts)
        buf = self._bufs.get(slot)
        if buf is None or buf.shape[0] < total:
            buf = torch.empty((total + total // 8 + 16, 3), dtype=torch.int32, device=self.device)
            self._bufs[slot] = buf
        base, ebase = [0], [0]
        for c, z in zip(counts, sizes):
            base.append(base[-1] + c)
            ebase.append(ebase[-1] + z)
        if self.packed:
            land = self._land.get(slot)
            if land is None or land.numel() < ebase[-1]:
                land = torch.empty(ebase[-1] + ebase[-1] // 8 + 16, dtype=torch.int32, device=self.device)
                self._land[slot] = land
        else:
            land, ebase = buf.view(-1), [3 * b for b in base]
        mine = sizes[self.rank]
        self.last_payload_elems = mine
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if mine:
                ops.append(dist.P2POp(dist.isend, payload[:mine], peer, group=self.group))
            if sizes[peer]:
                ops.append(dist.P2POp(dist.irecv, land[ebase[peer]:ebase[peer + 1]], peer, group=self.group))
        if n:
            buf[base[self.rank]:base[self.rank + 1]].copy_(hits[:n])
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self._pending[slot] = (reqs, buf, total, counts, base, land, ebase)
        return counts

    def finish(self, slot=0):
        """Waits for the exchange issued with start(slot); returns (gathered, counts) or None."""
        if slot not in self._pending:
            return None
        reqs, buf, total, counts, base, land, ebase = self._pending.pop(slot)
        for req in reqs:
            req.wait()
        if self.packed:  # the received payload of every peer -> triples at their final place
            if self.exchange == "pairs":  # uniform layout: two launches cover all peers
                lo, hi = base[self.rank], base[self.rank + 1]
                self._rebuild(land, 0, lo, buf[:lo])
                self._rebuild(land, ebase[self.rank + 1], total - hi, buf[hi:total])
            elif land.is_cuda:  # the 4-byte streams of all peers rebuilt by ONE launch over a segment table
                segs = [(ebase[peer], counts[peer], base[peer]) for peer in range(self.world)
                        if peer != self.rank and counts[peer]]
                if segs:
                    self.ac.hits_unpack4_segs_device(land, segs, buf, chars=self.chars)
            else:
                for peer in range(self.world):
                    if peer != self.rank:
                        self._rebuild(land, ebase[peer], counts[peer], buf[base[peer]:base[peer + 1]])
        return buf[:total], counts

    def gather_doc_hit_offsets(self, dho, counts):
        """Per-document hit offsets of all ranks, rebased to the global hit
        index (dho: [D_r+1] int64 on device).  Returns a [D+1] tensor."""
        dist = self.dist
        n_local = torch.tensor([dho.numel() - 1], dtype=torch.int64, device=self.device)
        all_n = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(all_n, n_local, group=self.group)
        ns = [int(x) for x in all_n.tolist()]
        base_hits = [0]
        for c in counts:
            base_hits.append(base_hits[-1] + c)
        out = torch.empty(sum(ns) + 1, dtype=torch.int64, device=self.device)
        doc_base = [0]
        for x in ns:
            doc_base.append(doc_base[-1] + x)
        mine = (dho[:-1] + base_hits[self.rank]).contiguous()
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if ns[self.rank]:
                ops.append(dist.P2POp(dist.isend, mine, peer, group=self.group))
            if ns[peer]:
                ops.append(dist.P2POp(dist.irecv, out[doc_base[peer]:doc_base[peer + 1]], peer, group=self.group))
        if ns[self.rank]:
            out[doc_base[self.rank]:doc_base[self.rank + 1]].copy_(mine)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        out[-1] = base_hits[-1]
        return out


def strong_scaling_pass(gatherer, match_fn, sub_corpus, sub_doc, timer=None):
    """One strong-scaling step of a rank (bench.py --gpus N, and its gloo rehearsal in the CPU suite): match this
    rank's contiguous document range, all-gatherv the hit buffers, gather the per-document offsets rebased to the
    global hit index.  match_fn(sub_corpus, sub_doc) -> (hits [cap,3] int32 on gatherer.device, n, dho [D_r+1]
    int64 on gatherer.device).  Returns (gathered hits [H,3], global offsets [D+1], seconds of (match, exchange))."""
    import time

    sync = timer or (lambda: None)
    sync()
    t0 = time.perf_counter()
    hits, n, dho = match_fn(sub_corpus, sub_doc)
    sync()
    t1 = time.perf_counter()
    allh, counts = gatherer.all_gatherv(hits, n)
    alld = gatherer.gather_doc_hit_offsets(dho, counts)
    sync()
    t2 = time.perf_counter()
    return allh, alld, (t1 - t0, t2 - t1)


def stream_digest(hits, dho):
    """sha256 over the ordered hit triples and the per-document offsets (host arrays / CPU tensors)."""
    import hashlib

    h = hashlib.sha256()
    h.update(np.ascontiguousarray(np.asarray(hits, dtype=np.int32)).tobytes())
    h.update(np.ascontiguousarray(np.asarray(dho, dtype=np.int64)).tobytes())
    return h.hexdigest()
