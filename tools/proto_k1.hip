// proto_k1.hip -- PROTOTYPE of the position-parallel first pass (not the product path).
//
// Every start position j is classified independently ("is anything observable or long
// starting here?"):
//   T2  (LDS, 64 Ki x 2 bit, direct index by the two bytes at j):  bit0 = the pair is a trie
//       path with children (deep), bit1 = the pair is a key (END2)
//   E3/E4 (LDS Bloom): the 3 / 4 bytes at j are a key
//   P5  (LDS Bloom): the 5 bytes at j are a trie path
// Positions with deep|END2 are compacted through a wave-private LDS ring; full 64-item
// batches run the Bloom tests; items with END2 or a positive test are appended to the
// chunk's item list.  This file measures the pass and checks its item lists against a
// numpy model (tools/proto_k1.py).
//
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/libproto_k1.so tools/proto_k1.hip
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
#ifndef PK_ABL
#define PK_ABL 0
#endif
constexpr int kWaves = PK_WAVES;          // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 1024;               // bytes per wave step (64 lanes x 16 B)
constexpr uint32_t kChunk = 4096;         // bytes per item region
constexpr uint32_t kRing = 128;           // ring entries per wave (8 B each)
constexpr uint32_t kOutCap = kChunk / 8;  // items per chunk
constexpr uint32_t K1 = 0x9E3779u, K2 = 0x85EBCBu, K3 = 0xC2B2AFu;

struct Args {
  const uint8_t *text;
  uint64_t n;
  const uint32_t *t2;       // [4096] 2-bit entries
  const uint32_t *bloom;    // [b_words]
  uint32_t b_words;
  uint16_t *items;          // [n_chunks * kOutCap]
  uint32_t *item_cnt;       // [n_chunks]
  unsigned long long *counts;  // [4] ring items, out items, overflow
};

// Bloom word index of a window (bytes 0..n-1), n = 3, 4, 5: m1 = mul24(lo, K1) covers bytes 0..2
__device__ __forceinline__ uint32_t widx(uint32_t h, uint32_t words, bool pow2) {
  if (pow2) return (h >> 10) & (words - 1);
  return (uint32_t)(((uint64_t)(h >> 8) * (uint64_t)((words << 8) & 0xFFFFFFu)) >> 32);
}
// one bit in each byte of the word, chosen by 12 bits of a hash of the first three bytes
__device__ __forceinline__ uint32_t bmask(uint32_t m1) {
  const uint32_t g = m1 ^ (m1 >> 11);
  return __builtin_amdgcn_perm(0x80402010u, 0x08040201u, g & 0x07070707u);
}

__global__ __launch_bounds__(kThreads) void k1(Args A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *t2 = reinterpret_cast<uint32_t *>(smem);                 // 16 KiB
  uint32_t *bl = t2 + 4096;
  uint2 *rings = reinterpret_cast<uint2 *>(bl + A.b_words);
  uint16_t *outs = reinterpret_cast<uint16_t *>(rings + kWaves * kRing);
  for (uint32_t i = threadIdx.x; i < 4096; i += kThreads) t2[i] = A.t2[i];
  for (uint32_t i = threadIdx.x; i < A.b_words; i += kThreads) bl[i] = A.bloom[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 *ring = rings + wave * kRing;
  uint16_t *obuf = outs + wave * kOutCap;
  const uint64_t n_chunks = (A.n + kChunk - 1) / kChunk;
  const uint64_t wave_id = (uint64_t)blockIdx.x * kWaves + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
  unsigned long long n_ring = 0, n_out = 0;

  for (uint64_t chunk = wave_id; chunk < n_chunks; chunk += n_waves) {
    const uint64_t c0 = chunk * kChunk;
    uint32_t head = 0, tail = 0;  // ring cursors (wave uniform)
    uint32_t ocnt = 0;            // items in obuf (wave uniform)
    auto load16 = [&](uint64_t g) -> uint4 {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g + 16 <= A.n) v = *reinterpret_cast<const uint4 *>(A.text + g);
      else if (g < A.n) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16 && g + j < A.n; j++) w[j >> 2] |= (uint32_t)A.text[g + j] << ((j & 3) * 8);
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      return v;
    };
    auto batch = [&](uint32_t navail) {
      // one batch of up to 64 ring items: Bloom tests, results appended to obuf
      const bool valid = (uint32_t)lane < navail;
      const uint2 it = ring[(head + lane) & (kRing - 1)];
      const uint32_t lo = it.x, hi = it.y;
      const uint32_t b4 = hi & 0xFFu;
      const uint32_t m1 = (lo & 0xFFFFFFu) * K1;
      const uint32_t bm = bmask(m1);
      const uint32_t h3 = ((lo >> 8) & 0xFFFFu) * K2 + m1;
      const uint32_t h4 = (lo >> 8) * K2 + m1;
      const uint32_t h5 = b4 * K3 + h4;
      const uint32_t w3 = bl[widx(h3, A.b_words, PK_POW2)];
      const uint32_t w4 = bl[widx(h4, A.b_words, PK_POW2)];
      const uint32_t w5 = bl[widx(h5, A.b_words, PK_POW2)];
      const bool deep = (hi >> 30) & 1u;
      const bool pos3 = (w3 & bm) == bm, pos4 = (w4 & bm) == bm, pos5 = (w5 & bm) == bm;
      const bool positive = deep && (pos3 || pos4 || pos5);
      const bool end2 = (hi >> 31) & 1u;
      const bool keep = valid && (positive || end2);
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        const uint32_t my = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, ocnt));
        if (my < kOutCap) obuf[my] = (uint16_t)(((hi >> 8) & 0xFFFu) | (deep && pos3 ? 0x1000u : 0u) | (deep && pos4 ? 0x2000u : 0u) | (deep && pos5 ? 0x4000u : 0u) | (end2 ? 0x8000u : 0u));
      }
      ocnt += (uint32_t)__popcll(mask);
      head += navail;
    };

    uint4 cur = load16(c0 + (uint64_t)lane * 16);
    for (uint32_t t = 0; t < kChunk / kTile; t++) {
      const uint64_t t0 = c0 + (uint64_t)t * kTile;
      if (t0 >= A.n) break;
      const uint4 nxt = load16(t0 + kTile + (uint64_t)lane * 16);  // next tile (also the halo of lane 63)
      uint32_t n0 = __shfl_down(cur.x, 1, 64), n1 = __shfl_down(cur.y, 1, 64);
      const uint32_t x0 = __builtin_amdgcn_readfirstlane(nxt.x), x1 = __builtin_amdgcn_readfirstlane(nxt.y);
      if (lane == 63) { n0 = x0; n1 = x1; }
      const uint32_t d[6] = {cur.x, cur.y, cur.z, cur.w, n0, n1};
      const uint32_t posbase = (t * kTile + lane * 16) << 8;
      // phase A: the 16 T2 lookups of the lane (independent LDS reads), codes packed 2 bits per position
      uint32_t codes = 0;
      if (PK_ABL == 3) { tail += __popc(d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5]); cur = nxt; continue; }
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int q = k >> 2, r = k & 3;
        const uint32_t lo = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], r) : d[q];
        // entry (b0, b1): dword = b0 | b1[3:0] << 8 (banks follow the low bits of b0), 2-bit field b1[7:4]
        const uint32_t word = t2[lo & 0xFFFu];
        codes |= ((word >> ((lo >> 11) & 30u)) & 3u) << (2 * k);
      }
      // phase B: live positions go to the ring; a full batch of 64 runs the Bloom tests
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int q = k >> 2, r = k & 3;
        const uint32_t code = (codes >> (2 * k)) & 3u;
        const bool live = code != 0;
        const unsigned long long m = __ballot(live);
        if (PK_ABL >= 2) { tail += (uint32_t)__popcll(m); continue; }
        if (live) {
          const uint32_t lo = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], r) : d[q];
          const uint32_t hi = r ? __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], r) : d[q + 1];
          const uint32_t idx = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail)) & (kRing - 1);
          ring[idx] = make_uint2(lo, (hi & 0xFFu) | (posbase + ((uint32_t)k << 8)) | (code << 30));
        }
        tail += (uint32_t)__popcll(m);
        if (PK_ABL == 0) { if (tail - head >= 64) batch(64); } else if (tail - head >= 64) head += 64;
      }
      cur = nxt;
    }
    if (tail != head) batch(tail - head);
    n_ring += tail;
    n_out += ocnt;
    // flush the chunk's items
    const uint32_t nw = min(ocnt, kOutCap);
    uint16_t *dst = A.items + chunk * kOutCap;
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = obuf[i];
    if (lane == 0) {
      A.item_cnt[chunk] = ocnt;
      if (ocnt > kOutCap) atomicAdd(A.counts + 2, 1ull);
    }
  }
  if (lane == 0) {
    atomicAdd(A.counts + 0, n_ring);
    atomicAdd(A.counts + 1, n_out);
  }
}
}  // namespace

extern "C" int proto_k1_lds(uint32_t b_words) {
  return (int)(16384 + (size_t)b_words * 4 + (size_t)kWaves * kRing * 8 + (size_t)kWaves * kOutCap * 2);
}

extern "C" int proto_k1_run(const uint8_t *text, uint64_t n, const uint32_t *t2, const uint32_t *bl, uint32_t b_words,
                            uint16_t *items, uint32_t *item_cnt,
                            unsigned long long *counts, int grid, void *stream) {
  Args A{text, n, t2, bl, b_words, items, item_cnt, counts};
  const size_t lds = (size_t)proto_k1_lds(b_words);
  hipError_t e = hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k1, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, A);
  return (int)hipGetLastError();
}
