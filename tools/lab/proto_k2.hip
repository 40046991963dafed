// proto_k2.hip -- lab prototype (not product code): the start-parallel walk behind the pair filter (proto_k1.hip).
// A wave takes a 4 KiB tile: the candidate bits of its 64 pieces (+ the piece before, for walks that reach into the tile) become
// batches of 64 candidates in position order; a lane walks ONE candidate's goto path through the hash image (pair: perfect
// hash, deeper: cuckoo table) over the tile's text in LDS; an END step is a hit of the reference exactly when no earlier
// start's walk is still alive at its end (exclusive prefix maximum of the walks' reaches).  Counts events and hits only:
// what the walks cost, and that the formulation is exact on the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kTile = 4096, kWarm = 64, kAhead = 64;
constexpr int kRowBytes = kWarm + kTile + kAhead;  // text [a - 64, a + 4096 + 64)
constexpr uint32_t kHTag = 1u << 24, kHK2 = 0x85EBCBu, kHMix = 0x2545F491u, kHMix2 = 0x9E3779B1u, kSalt = 0x5BD1E995u;
constexpr int kThreads = 256;
constexpr int kMaxEnds = 4;

struct K2P {
  const uint8_t *text;
  uint64_t n_bytes;
  const uint64_t *bitmap;  // bit b of word p: byte 64 p + b starts a candidate pair
  const uint8_t *disp;
  const uint4 *pairs, *deep;
  uint32_t n_groups, pair_log2, deep_log2, k1, max_len;
  int max_steps;
  unsigned long long *out;  // [0] events [1] hits [2] checksum [3] starts with more than kMaxEnds END steps [4] candidates [5] pair hits
};

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t rot11(uint32_t g) { return (g >> 11) | (g << 21); }

__global__ __launch_bounds__(kThreads) void k2_walk(K2P P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *dispb = smem;
  uint8_t *lentab = dispb + ((P.n_groups + 15u) & ~15u);
  uint8_t *rows = lentab + 256;
  for (uint32_t i = threadIdx.x; i < P.n_groups; i += kThreads) dispb[i] = P.disp[i];
  lentab[threadIdx.x] = (threadIdx.x & 0xE0u) == 0xC0u ? 16 : ((threadIdx.x & 0xF0u) == 0xE0u ? 24 : 8);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *row = rows + (size_t)wave * (kRowBytes + 64 * 2 + 16);
  uint16_t *list = reinterpret_cast<uint16_t *>(row + kRowBytes);
  const uint32_t *row32 = reinterpret_cast<const uint32_t *>(row);
  const uint32_t gmask = P.n_groups - 1u, psh = 32u - P.pair_log2, pmask = (1u << P.pair_log2) - 1u, dsh = 32u - P.deep_log2;
  const uint64_t n_tiles = (P.n_bytes + kTile - 1) / kTile;
  const uint64_t wid = (uint64_t)blockIdx.x * (kThreads / 64) + wave, nw = (uint64_t)gridDim.x * (kThreads / 64);
  const int warm = P.max_len > 1 ? (int)min(P.max_len - 1u, 63u) : 0;
  unsigned long long n_ev = 0, n_hit = 0, csum = 0, n_over = 0, n_cand = 0, n_pair = 0;

  auto char_at = [&](uint32_t o, uint32_t dend, uint32_t &c, uint32_t &L) {  // o: LDS offset of the unit
    const uint32_t lo = row32[o >> 2], hi = row32[(o >> 2) + 1];
    const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, o & 3u);
    const uint32_t s = lentab[w4 & 0xFFu];
    const uint32_t cm = __builtin_amdgcn_ubfe(0xC0C000u, 0u, s);
    const bool whole = (o + (s >> 3) <= dend) & ((w4 ^ 0x808000u) & cm) == 0u;
    const uint32_t se = whole ? s : 8u;
    c = __builtin_amdgcn_ubfe(w4, 0u, se);
    L = se >> 3;
  };

  for (uint64_t tile = wid; tile < n_tiles; tile += nw) {
    const int64_t a = (int64_t)tile * kTile;
    // ---- the tile's text: LDS offset o <-> text position a - 64 + o
    {
      const int64_t g0 = a - kWarm;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int idx = k * 64 + lane;  // 16-byte piece
        if (idx * 16 < kRowBytes) {
          const int64_t g = g0 + (int64_t)idx * 16;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (g >= 0 && g + 16 <= (int64_t)P.n_bytes) v = *reinterpret_cast<const uint4 *>(P.text + g);
          *reinterpret_cast<uint4 *>(row + idx * 16) = v;
        }
      }
    }
    const uint32_t dend = (uint32_t)min<int64_t>((int64_t)P.n_bytes - (a - kWarm), kRowBytes);  // (one document)
    // ---- candidate bits: lane l its own piece, lane 0 also the piece before (starts that can reach into the tile)
    const uint64_t p0 = (uint64_t)a / 64;
    unsigned long long m = (p0 + lane) * 64 < P.n_bytes ? P.bitmap[p0 + lane] : 0ull;
    unsigned long long mw = 0;
    if (lane == 0 && p0 > 0 && warm > 0) mw = P.bitmap[p0 - 1] & (~0ull << (64 - warm));
    uint32_t carry = 0;  // furthest LDS offset (exclusive) an earlier start's walk is alive at
    for (;;) {
      // ---- the next 64 candidates, in position order
      const uint32_t cnt = (uint32_t)__popcll(m) + (uint32_t)__popcll(mw);
      uint32_t incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(incl, d, 64);
        if (lane >= d) incl += u;
      }
      const uint32_t total = __shfl(incl, 63, 64);
      if (total == 0) break;
      uint32_t rank = incl - cnt;
      for (int it = 0; it < 64; it++) {  // (bounded: a lane gives at most 64 candidates to a batch)
        const bool go = rank < 64u && (mw | m) != 0ull;
        if (!__builtin_amdgcn_ballot_w64(go)) break;
        if (go) {
          const bool fromw = mw != 0ull;
          const unsigned long long cur = fromw ? mw : m;
          const uint32_t b = (uint32_t)__builtin_ctzll(cur);
          list[rank] = (uint16_t)((fromw ? 0u : (uint32_t)(kWarm + lane * 64)) + b);
          if (fromw) mw &= mw - 1; else m &= m - 1;
          rank++;
        }
      }
      const uint32_t nb = min(total, 64u);
      const bool have = (uint32_t)lane < nb;
      const uint32_t q = have ? list[lane] : 0u;
      // ---- the walk of candidate q
      uint32_t reach = 0;   // LDS offset (exclusive) of the last character the walk is alive at; 0: no pair
      uint32_t ej[kMaxEnds], ex[kMaxEnds];
      uint32_t ne = 0, over = 0;
#pragma unroll
      for (int k = 0; k < kMaxEnds; k++) ej[k] = ex[k] = 0;
      uint32_t E = 0, CF = 0, p = q;
      bool alive = false;
      {
        uint32_t c1, L1, c2, L2;
        char_at(q, dend, c1, L1);
        char_at(q + L1, dend, c2, L2);
        const uint32_t part = rot11(mul24(c1, kHK2));
        uint32_t h = mul24(c2, P.k1) + part;
        h ^= h >> 16;
        const uint32_t t = h * kHMix;
        const uint32_t d = dispb[(h >> 7) & gmask];
        const uint32_t sl = ((t >> psh) + d * ((t << 1) | 1u)) & pmask;
        const uint4 e = P.pairs[have ? sl : 0u];
        alive = have & e.x == (kHTag | c1) & (e.y & 0xFFFFFFu) == c2;
        E = e.z;
        CF = e.w;
        p = q + L1 + L2;
        if (alive) {
          reach = p;
          if (E >> 31) {
            ej[0] = p;
            ex[0] = (E & 0x3FFFFFu) | (e.y >> 24) << 22;
            ne = 1;
          }
        }
      }
      for (int step = 0; step < P.max_steps; step++) {  // (bounded: no key has more characters)
        uint32_t c, L;
        char_at(min(p, (uint32_t)(kRowBytes - 8)), dend, c, L);
        const bool go = alive & p < dend & ((CF >> (mul24(c, kHK2) >> 27)) & 1u) != 0u;
        if (!__builtin_amdgcn_ballot_w64(go)) break;
        const uint32_t B = E & 0x3FFFFFu;
        uint32_t h = mul24(c, P.k1) + (rot11(mul24(B, kHK2)) ^ kSalt);
        h ^= h >> 16;
        const uint32_t t = h * kHMix;
        const uint4 e1 = P.deep[go ? (t >> dsh) : 0u];
        const uint4 e2 = P.deep[go ? ((t * kHMix2) >> dsh) : 0u];
        const bool h1 = e1.x == B & (e1.y & 0xFFFFFFu) == c, h2 = e2.x == B & (e2.y & 0xFFFFFFu) == c;
        const bool hit = go & (h1 | h2);
        alive = hit;
        if (hit) {
          E = h1 ? e1.z : e2.z;
          CF = h1 ? e1.w : e2.w;
          const uint32_t ey = h1 ? e1.y : e2.y;
          p += L;
          reach = p;
          if (E >> 31) {
            if (ne < (uint32_t)kMaxEnds) {
#pragma unroll
              for (int k = 0; k < kMaxEnds; k++)
                if ((uint32_t)k == ne) {
                  ej[k] = p;
                  ex[k] = (E & 0x3FFFFFu) | (ey >> 24) << 22;
                }
              ne++;
            } else {
              over = 1;
            }
          }
        }
      }
      // ---- an END step at offset j is the reference's event when no earlier start is alive there
      uint32_t pm = reach;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(pm, d, 64);
        if (lane >= d) pm = max(pm, u);
      }
      const uint32_t last = __shfl(pm, 63, 64);
      uint32_t before = __shfl_up(pm, 1, 64);
      before = lane == 0 ? carry : max(before, carry);
      carry = max(carry, last);
#pragma unroll
      for (int k = 0; k < kMaxEnds; k++) {
        // ownership: the event's last byte lies in the tile: offsets (64, 64 + 4096]
        const bool ok = (uint32_t)k < ne && ej[k] > before && ej[k] > (uint32_t)kWarm && ej[k] <= (uint32_t)(kWarm + kTile);
        if (ok) {
          n_ev++;
          n_hit += ex[k] >> 22;
          csum += (unsigned long long)(a - kWarm + ej[k]) * 0x9E3779B97F4A7C15ull + (ex[k] & 0x3FFFFFu);
        }
      }
      n_over += over;
      n_cand += have ? 1 : 0;
      n_pair += reach ? 1 : 0;
    }
  }
  // ---- totals
  for (int d = 32; d >= 1; d >>= 1) {
    n_ev += __shfl_xor(n_ev, d, 64);
    n_hit += __shfl_xor(n_hit, d, 64);
    csum += __shfl_xor(csum, d, 64);
    n_over += __shfl_xor(n_over, d, 64);
    n_cand += __shfl_xor(n_cand, d, 64);
    n_pair += __shfl_xor(n_pair, d, 64);
  }
  if (lane == 0) {
    atomicAdd(&P.out[0], n_ev);
    atomicAdd(&P.out[1], n_hit);
    atomicAdd(&P.out[2], csum);
    atomicAdd(&P.out[3], n_over);
    atomicAdd(&P.out[4], n_cand);
    atomicAdd(&P.out[5], n_pair);
  }
}

extern "C" int proto_k2_run(const uint8_t *text, uint64_t n_bytes, const uint64_t *bitmap, const uint8_t *disp, const void *pairs,
                            const void *deep, uint32_t n_groups, uint32_t pair_log2, uint32_t deep_log2, uint32_t k1,
                            uint32_t max_len, unsigned long long *out, int grid, int reps, float *ms_out, int max_steps) {
  K2P P{text, n_bytes, bitmap, disp, (const uint4 *)pairs, (const uint4 *)deep, n_groups, pair_log2, deep_log2, k1, max_len, max_steps, out};
  const size_t lds = ((n_groups + 15u) & ~15u) + 256 + (size_t)(kThreads / 64) * (kRowBytes + 64 * 2 + 16);
  if (hipFuncSetAttribute((const void *)k2_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    (void)hipMemsetAsync(out, 0, 8 * 8, 0);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k2_walk, dim3(grid), dim3(kThreads), lds, 0, P);
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  *ms_out = best;
  return (int)hipGetLastError();
}
