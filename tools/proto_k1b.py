"""Driver + numpy model for the brute-force two-unit filter prototype (tools/proto_k1b.hip).
Prototype only: times the pass on the cfg 3 shape and checks its item lists; not the product path."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from aha_amd import synth  # noqa: E402

WAVES = int(os.environ.get("PK_WAVES", "16"))
ABL = int(os.environ.get("PK_ABL", "0"))
K1, K2, K3, K4 = 0x9E3779, 0x85EBCB, 0xC2B2AF, 0x27D4EB
M32 = np.uint64(0xFFFFFFFF)
CHUNK, OUTCAP = 4096, 512
LUT = np.array([8, 8, 8, 8, 8, 8, 16, 24], dtype=np.uint64)  # 8 * unit length by byte >> 5
LUT_LO, LUT_HI = 0x08080808, 0x18100808


def u64(x):
    return np.asarray(x, dtype=np.uint64)


def probe(W, n8, b_words):
    """(word index, mask) of windows W (8 bytes little endian) of n8 bits."""
    Wl = (W << (u64(64) - n8))  # uint64 wraps
    a1 = Wl >> u64(32)
    a0 = Wl & M32
    m = ((a1 & u64(0xFFFFFF)) * u64(K1)) & M32
    m = ((a1 >> u64(8)) * u64(K2) + m) & M32
    m = ((a0 >> u64(8)) * u64(K3) + m) & M32
    idx = ((m >> u64(17)) & u64(b_words - 1)).astype(np.int64)
    g = ((m & u64(0xFFFFFF)) * u64(K4) + a1) & M32
    sel = g >> u64(4)
    mask = u64(0)
    for j in range(4):
        mask = mask | (u64(1) << (((sel >> u64(8 * j)) & u64(7)) + u64(8 * j)))
    return idx, mask


def windows(t):
    n = t.size
    p = np.concatenate([t, np.zeros(16, np.uint8)]).astype(np.uint64)
    W = u64(0)
    for i in range(8):
        W = W | (p[i:n + i] << u64(8 * i))
    ul = LUT[(p >> u64(5)).astype(np.int64)]  # 8 * unit length of every byte
    s1 = ul[:n]
    j2 = np.arange(n) + (s1 >> u64(3)).astype(np.int64)
    n8 = s1 + ul[j2]
    return W, n8


def build(blob, offs, b_words):
    K = offs.size - 1
    keys = [bytes(blob[int(offs[i]):int(offs[i + 1])]) for i in range(K)]
    ent = set()
    ragged = 0
    for k in keys:
        u1 = int(LUT[k[0] >> 5]) // 8
        if len(k) <= u1:
            ragged += 1
            continue
        n = u1 + int(LUT[k[u1] >> 5]) // 8
        if len(k) < n:
            ragged += 1
            continue
        ent.add(k[:n])
    assert ragged == 0, f"{ragged} keys shorter than their own two-unit window"
    ents = sorted(ent)
    W = u64([int.from_bytes(e, "little") for e in ents])
    n8 = u64([8 * len(e) for e in ents])
    idx, mask = probe(W, n8, b_words)
    bl = np.zeros(b_words, np.uint32)
    np.bitwise_or.at(bl, idx, mask.astype(np.uint32))
    print(f"keys {K}: {len(ents)} two-unit prefixes; Bloom {b_words * 4 / 1024:.0f} KiB fill {np.unpackbits(bl.view(np.uint8)).mean():.3f}")
    return bl, ent


def model(t, bl, ent):
    W, n8 = windows(t)
    idx, mask = probe(W, n8, bl.size)
    pos = (bl[idx].astype(np.uint64) & mask) == mask
    nb = (n8 >> u64(3)).astype(np.int64)
    tb = t.tobytes() + b"\0" * 16
    true = np.fromiter((tb[j:j + nb[j]] in ent for j in range(t.size)), bool, count=t.size)
    assert not np.any(true & ~pos), "false negative"
    return pos, true


def main():
    import torch
    so = os.path.join(ROOT, "tools", f"libproto_k1b_w{WAVES}_a{ABL}.so")
    src = os.path.join(ROOT, "tools", "proto_k1b.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", f"-DPK_WAVES={WAVES}", f"-DPK_ABL={ABL}",
                               "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    L.proto_k1b_run.restype = C.c_int
    L.proto_k1b_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    blob, offs, nf = synth.keys(3)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
    b_words = int(os.environ.get("PK_B_WORDS", 32768))
    bl, ent = build(blob, offs, b_words)
    dev = torch.device("cuda:0")
    d_bl = torch.from_numpy(bl.view(np.int32)).to(dev)
    d_text = torch.from_numpy(corpus).to(dev)
    n_chunks = (corpus.size + CHUNK - 1) // CHUNK
    d_items = torch.zeros(n_chunks * OUTCAP, dtype=torch.int16, device=dev)
    d_icnt = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    grid = torch.cuda.get_device_properties(0).multi_processor_count

    def run(n):
        d_cnt.zero_()
        rc = L.proto_k1b_run(d_text.data_ptr(), n, d_bl.data_ptr(), b_words, LUT_LO, LUT_HI, d_items.data_ptr(),
                             d_icnt.data_ptr(), d_cnt.data_ptr(), grid, None)
        assert rc == 0, rc

    m = min(corpus.size, 1 << 21)
    run(m)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    pos, true = model(corpus[:m], bl, ent)
    print(f"model: positive {pos.sum()} ({pos.mean():.5f}/B) true {true.mean():.5f}/B; kernel positives {cnt[0]} overflow {cnt[2]}")
    if not ABL:
        assert cnt[0] == pos.sum() and cnt[2] == 0
        icnt = d_icnt.cpu().numpy()[: (m + CHUNK - 1) // CHUNK]
        items = d_items.cpu().numpy().view(np.uint16).reshape(-1, OUTCAP)
        for c in range(0, icnt.size, max(1, icnt.size // 64)):
            got = np.sort(items[c, : icnt[c]].astype(np.int64))
            want = np.nonzero(pos[c * CHUNK:min((c + 1) * CHUNK, m)])[0]
            assert np.array_equal(got, want), c
        print("item lists agree with the model on sampled chunks")
    run(corpus.size)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    reps = 5
    for _ in range(reps):
        run(corpus.size)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    cnt = d_cnt.cpu().numpy()
    print(f"k1b abl {ABL}: {ms:.3f} ms per pass over {corpus.size} bytes = {corpus.size / ms / 1e6:.1f} GB/s; positives "
          f"{cnt[0] / corpus.size:.4f}/B, overflowed chunks {cnt[2]}")


if __name__ == "__main__":
    main()
