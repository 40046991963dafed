"""Driver + numpy model for the position-parallel first-pass prototype (tools/proto_k1.hip).
Prototype only: times the pass on the cfg 3 shape and checks its item lists; not the product path."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aha_amd import synth  # noqa: E402

WAVES = int(os.environ.get("PK_WAVES", "16"))
ABL = int(os.environ.get("PK_ABL", "0"))
so = os.path.join(ROOT, "tools", f"libproto_k1_w{WAVES}_a{ABL}.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "tools", "proto_k1.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", f"-DPK_WAVES={WAVES}", "-DPK_POW2=" + os.environ.get("PK_POW2", "0"), f"-DPK_ABL={ABL}", "-shared", "-fPIC",
                           "-o", so, os.path.join(ROOT, "tools", "proto_k1.hip")])
L = C.CDLL(so)
L.proto_k1_run.restype = C.c_int
L.proto_k1_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L.proto_k1_lds.restype = C.c_int
L.proto_k1_lds.argtypes = [C.c_uint32]

K1, K2, K3 = 0x9E3779, 0x85EBCB, 0xC2B2AF
M32 = np.uint64(0xFFFFFFFF)
CHUNK, OUTCAP = 4096, 512
POW2 = int(os.environ.get("PK_POW2", "0"))


def u64(x):
    return np.asarray(x, dtype=np.uint64)


def hashes(lo, b4):
    """(h3, h4, h5, mask) of windows: lo = bytes 0..3 little endian, b4 = byte 4."""
    lo = u64(lo)
    b4 = u64(b4)
    m1 = ((lo & u64(0xFFFFFF)) * u64(K1)) & M32
    h3 = ((((lo >> u64(8)) & u64(0xFFFF)) * u64(K2)) + m1) & M32
    h4 = (((lo >> u64(8)) * u64(K2)) + m1) & M32
    h5 = ((b4 * u64(K3)) + h4) & M32
    g = m1 ^ (m1 >> u64(11))
    mask = u64(0)
    for j in range(4):
        mask = mask | (u64(1) << (((g >> u64(8 * j)) & u64(7)) + u64(8 * j)))
    return h3, h4, h5, mask


def widx(h, words):
    if POW2:
        return ((h >> u64(10)) & u64(words - 1)).astype(np.int64)
    return (((h >> u64(8)) * u64((words << 8) & 0xFFFFFF)) >> u64(32)).astype(np.int64)


def windows(t, n=None):
    """lo (bytes 0..3 little endian) and byte 4 for every position of uint8 array t (zero padded)."""
    n = t.size if n is None else n
    p = np.concatenate([t, np.zeros(8, np.uint8)]).astype(np.uint64)
    lo = p[0:n] | (p[1:n + 1] << u64(8)) | (p[2:n + 2] << u64(16)) | (p[3:n + 3] << u64(24))
    return lo, p[4:n + 4]


def build_tables(blob, offs, b_words):
    K = offs.size - 1
    lens = (offs[1:] - offs[:-1]).astype(np.int64)
    assert lens.min() >= 2, "prototype: no 1-byte keys"
    pad = np.concatenate([blob, np.zeros(8, np.uint8)]).astype(np.uint64)
    o = offs[:-1].astype(np.int64)
    b = [pad[o + i] for i in range(5)]
    pair = (b[0] | (b[1] << u64(8))).astype(np.int64)
    t2 = np.zeros(65536, np.uint8)
    np.bitwise_or.at(t2, pair[lens > 2], 1)
    np.bitwise_or.at(t2, pair[lens == 2], 2)
    t2w = np.zeros(4096, np.uint32)  # dword = idx & 0xFFF, 2-bit field idx >> 12 (idx = b0 | b1 << 8)
    for k in range(16):
        t2w |= (t2[k * 4096:(k + 1) * 4096].astype(np.uint32) & 3) << np.uint32(2 * k)
    lo = b[0] | (b[1] << u64(8)) | (b[2] << u64(16)) | (b[3] << u64(24))
    bl = np.zeros(b_words, np.uint32)
    m3 = lens == 3
    h3, _, _, mk = hashes(lo[m3] & u64(0xFFFFFF), 0)
    np.bitwise_or.at(bl, widx(h3, b_words), mk.astype(np.uint32))
    m4 = lens == 4
    _, h4, _, mk = hashes(lo[m4], 0)
    np.bitwise_or.at(bl, widx(h4, b_words), mk.astype(np.uint32))
    m5 = lens >= 5
    _, _, h5, mk = hashes(lo[m5], b[4][m5])
    np.bitwise_or.at(bl, widx(h5, b_words), mk.astype(np.uint32))
    sets = dict(e3=set((lo[m3] & u64(0xFFFFFF)).tolist()), e4=set(lo[m4].tolist()),
                p5=set((lo[m5] | (b[4][m5] << u64(32))).tolist()))
    print(f"keys {K}: len3 {m3.sum()} len4 {m4.sum()} p5 entries {len(sets['p5'])}; T2 live pairs {np.count_nonzero(t2)}; "
          f"Bloom fill {np.unpackbits(bl.view(np.uint8)).mean():.3f}")
    return t2, t2w, bl, sets


def model(t, t2, bl, sets):
    lo, b4 = windows(t)
    code = t2[(lo & u64(0xFFFF)).astype(np.int64)]
    h3, h4, h5, mk = hashes(lo, b4)
    blu = bl.astype(np.uint64)
    deep = (code & 1) != 0
    end2 = (code & 2) != 0
    p3 = deep & ((blu[widx(h3, bl.size)] & mk) == mk)
    p4 = deep & ((blu[widx(h4, bl.size)] & mk) == mk)
    p5 = deep & ((blu[widx(h5, bl.size)] & mk) == mk)
    # exact membership (no false negatives allowed)
    nd = int(deep.sum())
    t3 = np.fromiter(((int(x) & 0xFFFFFF) in sets["e3"] for x in lo[deep]), bool, count=nd)
    t4 = np.fromiter((int(x) in sets["e4"] for x in lo[deep]), bool, count=nd)
    k5 = lo[deep] | (b4[deep] << u64(32))
    t5 = np.fromiter((int(x) in sets["p5"] for x in k5), bool, count=nd)
    assert not np.any(t3 & ~p3[deep]) and not np.any(t4 & ~p4[deep]) and not np.any(t5 & ~p5[deep]), "false negative"
    true_pos = np.zeros(t.size, bool)
    true_pos[deep] = t3 | t4 | t5
    return code, p3, p4, p5, end2, true_pos


def main():
    n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    blob, offs, nf = synth.keys(3)
    corpus, doc = synth.corpus(3, blob, offs, nf, n_bytes=n_bytes)
    fixed = L.proto_k1_lds(0)
    b_words = (160 * 1024 - fixed) // 4
    if POW2:
        b_words = 1 << (b_words.bit_length() - 1)
    b_words = int(os.environ.get("PK_B_WORDS", b_words))
    print(f"waves {WAVES}, pow2 {POW2}, LDS fixed {fixed} B, b_words {b_words} ({b_words * 4 / 1024:.1f} KiB)")
    t2, t2w, bl, sets = build_tables(blob, offs, b_words)
    dev = torch.device("cuda:0")
    d_t2 = torch.from_numpy(t2w.view(np.int32)).to(dev)
    d_bl = torch.from_numpy(bl.view(np.int32)).to(dev)
    d_text = torch.from_numpy(corpus).to(dev)
    n_chunks = (corpus.size + CHUNK - 1) // CHUNK
    d_items = torch.zeros(n_chunks * OUTCAP, dtype=torch.int16, device=dev)
    d_icnt = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)
    grid = torch.cuda.get_device_properties(0).multi_processor_count

    def run(n):
        d_cnt.zero_()
        rc = L.proto_k1_run(d_text.data_ptr(), n, d_t2.data_ptr(), d_bl.data_ptr(), b_words,
                            d_items.data_ptr(), d_icnt.data_ptr(), d_cnt.data_ptr(), grid, None)
        assert rc == 0, rc

    # ---- check against the model on a prefix
    m = min(corpus.size, 1 << 21)
    run(m)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    code, p3, p4, p5, end2, true_pos = model(corpus[:m], t2, bl, sets)
    live = code != 0
    positive = p3 | p4 | p5
    keep = positive | end2
    print(f"model: live {live.sum()} ({live.mean():.4f}/B) end2 {end2.mean():.4f}/B positive {positive.mean():.5f}/B "
          f"(p3 {p3.mean():.5f} p4 {p4.mean():.5f} p5 {p5.mean():.5f}) true positive {true_pos.mean():.5f}/B items {keep.sum()} ({keep.mean():.4f}/B)")
    print(f"kernel: ring items {cnt[0]} out items {cnt[1]} overflowed chunks {cnt[2]}")
    if ABL:
        print("ablation build: checks skipped")
    else:
        assert cnt[0] == live.sum() and cnt[1] == keep.sum() and cnt[2] == 0
    icnt = d_icnt.cpu().numpy()[: (m + CHUNK - 1) // CHUNK]
    items = d_items.cpu().numpy().view(np.uint16).reshape(-1, OUTCAP)
    for c in range(0, icnt.size if not ABL else 0, max(1, icnt.size // 64)):
        got = np.sort(items[c, : icnt[c]].astype(np.int64))
        sl = slice(c * CHUNK, min((c + 1) * CHUNK, m))
        idx = np.nonzero(keep[sl])[0]
        want = np.sort(idx | (p3[sl][idx].astype(np.int64) << 12) | (p4[sl][idx].astype(np.int64) << 13) |
                       (p5[sl][idx].astype(np.int64) << 14) | (end2[sl][idx].astype(np.int64) << 15))
        assert np.array_equal(got, want), c
    print("item lists agree with the model on sampled chunks")

    # ---- timing
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
    print(f"k1: {ms:.3f} ms per pass over {corpus.size} bytes = {corpus.size / ms / 1e6:.1f} GB/s; ring items "
          f"{cnt[0] / corpus.size:.4f}/B, out items {cnt[1] / corpus.size:.4f}/B, overflowed chunks {cnt[2]}")


if __name__ == "__main__":
    main()
