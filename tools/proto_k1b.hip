// proto_k1b.hip -- PROTOTYPE (not the product path): brute-force position-parallel filter pass.
//
// Every byte position j is a candidate start.  Its window is the first TWO UNITS of the text at j,
// a unit being 1, 2 or 3 bytes long as told by the high bits of its first byte (UTF-8 lead classes;
// any self-delimiting length function works as long as keys and text use the same one).  One probe of
// a blocked Bloom filter (one bit in each byte of a 32-bit word) over {first two units of every key}
// answers "may a key start here / may the trie walk from here be longer than two units?".
// Negative => the start is boring (no key starts here, walk shorter than 6 bytes).  Positives are
// appended to the chunk's item list for the exact resolver.  No rings, no per-lane state: straight-
// line code with 16 independent chains per lane.
//
// Build: hipcc -O3 --offload-arch=gfx950 -DPK_WAVES=16 -shared -fPIC -o tools/libproto_k1b.so tools/proto_k1b.hip
#include <hip/hip_runtime.h>
#include <cstdint>

#ifndef PK_ABL
#define PK_ABL 0
#endif
namespace {
constexpr int kWaves = PK_WAVES;
constexpr int kThreads = kWaves * 64;
constexpr int kTile = 1024;
constexpr uint32_t kChunk = 4096;
constexpr uint32_t kOutCap = kChunk / 8;
constexpr uint32_t K1 = 0x9E3779u, K2 = 0x85EBCBu, K3 = 0xC2B2AFu, K4 = 0x27D4EBu;

struct Args {
  const uint8_t *text;
  uint64_t n;
  const uint32_t *bloom;  // [b_words] power of two
  uint32_t b_words;
  uint32_t lut_lo, lut_hi;  // 8 x (8 * unit length) indexed by byte >> 5
  uint16_t *items;          // [n_chunks * kOutCap] chunk-relative positions
  uint32_t *item_cnt;       // [n_chunks]
  unsigned long long *counts;  // [0] positives, [2] overflowed chunks
};

__global__ __launch_bounds__(kThreads) void k1b(Args A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *bl = reinterpret_cast<uint32_t *>(smem);
  uint16_t *outs = reinterpret_cast<uint16_t *>(bl + A.b_words);
  for (uint32_t i = threadIdx.x; i < A.b_words; i += kThreads) bl[i] = A.bloom[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint16_t *obuf = outs + wave * kOutCap;
  const uint32_t amask = (A.b_words - 1) << 2;
  const uint64_t n_chunks = (A.n + kChunk - 1) / kChunk;
  const uint64_t wave_id = (uint64_t)blockIdx.x * kWaves + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
  unsigned long long n_pos = 0;

  for (uint64_t chunk = wave_id; chunk < n_chunks; chunk += n_waves) {
    const uint64_t c0 = chunk * kChunk;
    uint32_t ocnt = 0;
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
    uint4 cur = load16(c0 + (uint64_t)lane * 16);
    for (uint32_t t = 0; t < kChunk / kTile; t++) {
      const uint64_t t0 = c0 + (uint64_t)t * kTile;
      if (t0 >= A.n) break;
      const uint4 nxt = load16(t0 + kTile + (uint64_t)lane * 16);
      uint32_t n0 = __shfl_down(cur.x, 1, 64), n1 = __shfl_down(cur.y, 1, 64);
      const uint32_t x0 = __builtin_amdgcn_readfirstlane(nxt.x), x1 = __builtin_amdgcn_readfirstlane(nxt.y);
      if (lane == 63) { n0 = x0; n1 = x1; }
      const uint32_t d[6] = {cur.x, cur.y, cur.z, cur.w, n0, n1};
      if (PK_ABL == 3) { n_pos += __popc(d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5]); cur = nxt; continue; }
      // 8 * unit length of every byte (4 bytes per v_perm)
      uint32_t U[6];
#pragma unroll
      for (int i = 0; i < 6; i++) U[i] = __builtin_amdgcn_perm(A.lut_hi, A.lut_lo, (d[i] >> 5) & 0x07070707u);
      uint32_t pm = 0;  // positive positions of this lane
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int q = k >> 2, r = k & 3;
        const uint32_t lo = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], r) : d[q];
        const uint32_t hi = r ? __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], r) : d[q + 1];
        const uint32_t ulo = r ? __builtin_amdgcn_alignbyte(U[q + 1], U[q], r) : U[q];
        const uint32_t uhi = r ? __builtin_amdgcn_alignbyte(U[q + 2], U[q + 1], r) : U[q + 1];
        // byte 0 of ulo = 8 * len(unit 1) (8..24) = bit offset of unit 2 in the window
        const uint32_t u2 = __builtin_amdgcn_alignbit(uhi, ulo, ulo);
        const uint32_t n8 = (ulo & 0xFFu) + (u2 & 0xFFu);  // window bits (16..48)
        const uint64_t W = ((uint64_t)hi << 32) | lo;
        const uint64_t Wl = W << (64u - n8);                // left-aligned: bytes beyond the window dropped
        const uint32_t a1 = (uint32_t)(Wl >> 32), a0 = (uint32_t)Wl;
        uint32_t m = (a1 & 0xFFFFFFu) * K1;
        m = (a1 >> 8) * K2 + m;
        m = (a0 >> 8) * K3 + m;
        if (PK_ABL == 2) { pm |= (m >> 31) << k; continue; }
        const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(bl) + ((m >> 15) & amask));
        const uint32_t g = (m & 0xFFFFFFu) * K4 + a1;
        const uint32_t bm = __builtin_amdgcn_perm(0x80402010u, 0x08040201u, (g >> 4) & 0x07070707u);
        pm |= ((word & bm) == bm ? 1u : 0u) << k;
      }
      if (PK_ABL == 1) { n_pos += __popc(pm); cur = nxt; continue; }
      // output: lane-local lists, wave prefix of the counts
      const uint32_t cnt = __popc(pm);
      uint32_t incl = cnt;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t o = __shfl_up(incl, dd, 64);
        if (lane >= dd) incl += o;
      }
      const uint32_t total = __shfl(incl, 63, 64);
      uint32_t w = ocnt + incl - cnt;
      uint32_t rest = pm;
      const uint32_t pbase = t * kTile + lane * 16;
      while (rest) {
        const uint32_t k = __builtin_ctz(rest);
        rest &= rest - 1;
        if (w < kOutCap) obuf[w] = (uint16_t)(pbase + k);
        w++;
      }
      ocnt += total;
      cur = nxt;
    }
    n_pos += (lane == 0) ? ocnt : 0;
    const uint32_t nw = min(ocnt, kOutCap);
    uint16_t *dst = A.items + chunk * kOutCap;
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = obuf[i];
    if (lane == 0) {
      A.item_cnt[chunk] = ocnt;
      if (ocnt > kOutCap) atomicAdd(A.counts + 2, 1ull);
    }
  }
  for (int dd = 32; dd >= 1; dd >>= 1) n_pos += __shfl_down(n_pos, dd, 64);
  if (lane == 0) atomicAdd(A.counts + 0, n_pos);
}
}  // namespace

extern "C" int proto_k1b_lds(uint32_t b_words) { return (int)((size_t)b_words * 4 + (size_t)kWaves * kOutCap * 2); }

extern "C" int proto_k1b_run(const uint8_t *text, uint64_t n, const uint32_t *bl, uint32_t b_words, uint32_t lut_lo,
                             uint32_t lut_hi, uint16_t *items, uint32_t *item_cnt, unsigned long long *counts, int grid,
                             void *stream) {
  Args A{text, n, bl, b_words, lut_lo, lut_hi, items, item_cnt, counts};
  const size_t lds = (size_t)proto_k1b_lds(b_words);
  hipError_t e = hipFuncSetAttribute((const void *)k1b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k1b, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, A);
  return (int)hipGetLastError();
}
