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
                        all-gather would be per-link bound.  packed=True puts
                        {end, value} pairs on the links (8 B per hit; the start
                        is end - key length) and rebuilds the triples on arrival.

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


class HitGatherer:
    """all-gatherv of hit triples ([n,3] int32 tensors) across ranks.

    packed=True (needs `ac`, the rank's automaton -- replicated, so every rank
    holds the same key lengths): the payload on the links is {end, value} pairs,
    8 instead of 12 bytes per hit, because Hit#start = Hit#end - len(key[value])
    (src/aha/ac.cr:270-272); the triples are rebuilt on arrival
    (aha_ac_hits_pack_device / _unpack_device on device tensors; plain torch
    indexing on CPU tensors, i.e. in the gloo rehearsal).  xGMI links are the
    scarce resource of the exchange (one link per peer), HBM bandwidth is not."""

    def __init__(self, dist, device, group=None, ac=None, packed=False, chars=False):
        self.dist = dist
        self.device = device
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._counts = torch.zeros(self.world, dtype=torch.int64, device=device)
        self._mine = torch.zeros(1, dtype=torch.int64, device=device)
        self._buf = None
        self.packed = bool(packed)
        self.chars = bool(chars)
        self.ac = ac
        if self.packed:
            if ac is None:
                raise ValueError("packed exchange needs the automaton (key lengths)")
            self._klen = None if torch.device(device).type == "cuda" else torch.from_numpy(ac.key_lengths(chars))
            self._pk = {}  # slot -> (send pairs, recv pairs)

    # -- packed payload helpers --------------------------------------------------------
    def _pack(self, hits, n, slot):
        """hits[:n] -> contiguous [n,2] pairs in a per-slot send buffer."""
        send, recv = self._pk.get(slot, (None, None))
        if send is None or send.shape[0] < n:
            send = torch.empty((n + n // 8 + 16, 2), dtype=torch.int32, device=hits.device)
        self._pk[slot] = (send, recv)
        if n:
            if hits.is_cuda:
                self.ac.hits_pack_device(hits, n, send)
            else:
                send[:n].copy_(hits[:n, 1:3])
        return send

    def _recv_pairs(self, total, slot):
        send, recv = self._pk.get(slot, (None, None))
        if recv is None or recv.shape[0] < total:
            recv = torch.empty((total + total // 8 + 16, 2), dtype=torch.int32, device=self.device)
        self._pk[slot] = (send, recv)
        return recv

    def _unpack(self, pairs, lo, hi, out):
        """pairs[lo:hi] -> out[lo:hi] triples."""
        n = hi - lo
        if n <= 0:
            return
        if pairs.is_cuda:
            self.ac.hits_unpack_device(pairs[lo:hi], n, out[lo:hi], chars=self.chars)
        else:
            p = pairs[lo:hi]
            out[lo:hi, 1:3] = p
            out[lo:hi, 0] = p[:, 0] - self._klen[p[:, 1].long()]

    def all_gatherv(self, hits, n):
        """hits: [cap,3] int32 on self.device, first n rows valid.  Returns
        (gathered [sum_n,3] view, counts list); rank r's hits start at
        sum(counts[:r]) -- global order = rank order (contiguous doc ranges)."""
        if self.packed:
            self.start(hits, n, slot=2)
            return self.finish(2)
        dist = self.dist
        self._mine[0] = n
        dist.all_gather_into_tensor(self._counts, self._mine, group=self.group)
        counts = [int(c) for c in self._counts.tolist()]
        total = sum(counts)
        if self._buf is None or self._buf.shape[0] < total:
            self._buf = torch.empty((total + total // 8 + 16, 3), dtype=torch.int32, device=self.device)
        out = self._buf
        base = [0]
        for c in counts:
            base.append(base[-1] + c)
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if n:
                ops.append(dist.P2POp(dist.isend, hits[:n], peer, group=self.group))
            if counts[peer]:
                ops.append(dist.P2POp(dist.irecv, out[base[peer]:base[peer + 1]], peer, group=self.group))
        if n:
            out[base[self.rank]:base[self.rank + 1]].copy_(hits[:n])
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out[:total], counts

    # -- overlapped form: the exchange of step i runs beside the match of step i+1 ------
    def start(self, hits, n, slot=0):
        """Issues the all-gatherv of hits[:n] without waiting for the payload
        (the 8-byte counts exchange is synchronous).  Uses output buffer
        `slot` (0/1) so that two exchanges can be in flight; the caller must not
        overwrite hits[:n] before finish(slot).  Returns the counts."""
        dist = self.dist
        if not hasattr(self, "_pending"):
            self._pending = {}
            self._bufs = {}
        self.finish(slot)
        self._mine[0] = n
        dist.all_gather_into_tensor(self._counts, self._mine, group=self.group)
        counts = [int(c) for c in self._counts.tolist()]
        total = sum(counts)
        buf = self._bufs.get(slot)
        if buf is None or buf.shape[0] < total:
            buf = torch.empty((total + total // 8 + 16, 3), dtype=torch.int32, device=self.device)
            self._bufs[slot] = buf
        base = [0]
        for c in counts:
            base.append(base[-1] + c)
        payload, landing = hits, buf
        if self.packed:  # 8 B per hit on the links; triples rebuilt in finish()
            payload = self._pack(hits, n, slot)
            landing = self._recv_pairs(total, slot)
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if n:
                ops.append(dist.P2POp(dist.isend, payload[:n], peer, group=self.group))
            if counts[peer]:
                ops.append(dist.P2POp(dist.irecv, landing[base[peer]:base[peer + 1]], peer, group=self.group))
        if n:
            buf[base[self.rank]:base[self.rank + 1]].copy_(hits[:n])
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self._pending[slot] = (reqs, buf, total, counts, base)
        return counts

    def finish(self, slot=0):
        """Waits for the exchange issued with start(slot); returns (gathered, counts) or None."""
        if not hasattr(self, "_pending") or slot not in self._pending:
            return None
        reqs, buf, total, counts, base = self._pending.pop(slot)
        for req in reqs:
            req.wait()
        if self.packed:  # the received pairs of every peer -> triples at their final place
            pairs = self._pk[slot][1]
            self._unpack(pairs, 0, base[self.rank], buf)
            self._unpack(pairs, base[self.rank + 1], total, buf)
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
