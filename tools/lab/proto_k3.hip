// proto_k3.hip -- lab prototype (not product code): the start-parallel walk, second form (proto_k2.hip is the first).
// A wave takes a 4 KiB tile; every lane walks the candidates of ITS 64-byte piece, two per iteration (both pair probes in
// flight, no extraction scans); a pair that may continue becomes a WALKER in a queue in LDS, and every iteration a lane that
// holds a walker also advances it by one step (its two cuckoo loads fly beside the pair probes).  A walker that ends leaves
// its interval (start, reach); at the end of the tile every END step is counted unless a walker with an earlier start reaches
// it.  Counts events and hits only (exactness against the product; what the walks cost in this form).
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kTile = 4096, kWarm = 64, kAhead = 64;
constexpr int kRowBytes = kWarm + kTile + kAhead;
constexpr uint32_t kHTag = 1u << 24, kHK2 = 0x85EBCBu, kHMix = 0x2545F491u, kHMix2 = 0x9E3779B1u, kSalt = 0x5BD1E995u;
constexpr int kThreads = 256;
constexpr int kEvPerLane = 8;    // pair events a lane keeps per tile
constexpr int kQueue = 128;      // walkers per tile (queue entries)
constexpr int kDeepEv = 256;     // END steps of walkers per tile

struct K3P {
  const uint8_t *text;
  uint64_t n_bytes;
  const uint64_t *bitmap;
  const uint8_t *disp;
  const uint4 *pairs, *deep;
  uint32_t n_groups, pair_log2, deep_log2, k1, max_len;
  unsigned long long *out;  // [0] events [1] hits [2] overflows [3] candidates [4] walkers [5] iterations
};

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
__device__ __forceinline__ uint32_t rot11(uint32_t g) { return (g >> 11) | (g << 21); }

// per wave in LDS
struct WaveLds {
  uint8_t row[kRowBytes];
  uint2 ev[64 * kEvPerLane];   // pair events by lane: {end offset (exclusive), start offset | hits << 16}
  uint4 queue[kQueue];         // walkers waiting / intervals of finished ones: {start, p, E, CF} -> {start, reach, 0, 0}
  uint2 dev[kDeepEv];          // END steps of walkers: {end offset, start | hits << 16}
};

__global__ __launch_bounds__(kThreads) void k3_walk(K3P P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *dispb = smem;
  uint8_t *lentab = dispb + ((P.n_groups + 15u) & ~15u);
  WaveLds *W = reinterpret_cast<WaveLds *>(lentab + 256) + (threadIdx.x >> 6);
  for (uint32_t i = threadIdx.x; i < P.n_groups; i += kThreads) dispb[i] = P.disp[i];
  lentab[threadIdx.x] = (threadIdx.x & 0xE0u) == 0xC0u ? 16 : ((threadIdx.x & 0xF0u) == 0xE0u ? 24 : 8);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t *row32 = reinterpret_cast<const uint32_t *>(W->row);
  const uint32_t gmask = P.n_groups - 1u, psh = 32u - P.pair_log2, pmask = (1u << P.pair_log2) - 1u, dsh = 32u - P.deep_log2;
  const uint64_t n_tiles = (P.n_bytes + kTile - 1) / kTile;
  const uint64_t wid = (uint64_t)blockIdx.x * (kThreads / 64) + wave, nw = (uint64_t)gridDim.x * (kThreads / 64);
  const int warm = P.max_len > 1 ? (int)min(P.max_len - 1u, 63u) : 0;
  unsigned long long n_ev = 0, n_hit = 0, n_over = 0, n_cand = 0, n_walk = 0, n_iter = 0;

  auto char_at = [&](uint32_t o, uint32_t dend, uint32_t &c, uint32_t &L) {
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
    {
      const int64_t g0 = a - kWarm;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int idx = k * 64 + lane;
        if (idx * 16 < kRowBytes) {
          const int64_t g = g0 + (int64_t)idx * 16;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (g >= 0 && g + 16 <= (int64_t)P.n_bytes) v = *reinterpret_cast<const uint4 *>(P.text + g);
          *reinterpret_cast<uint4 *>(W->row + idx * 16) = v;
        }
      }
    }
    const uint32_t dend = (uint32_t)min<int64_t>((int64_t)P.n_bytes - (a - kWarm), kRowBytes);
    const uint64_t p0 = (uint64_t)a / 64;
    unsigned long long m = (p0 + lane) * 64 < P.n_bytes ? P.bitmap[p0 + lane] : 0ull;
    unsigned long long mw = 0;
    if (lane == 0 && p0 > 0 && warm > 0) mw = P.bitmap[p0 - 1] & (~0ull << (64 - warm));
    n_cand += (unsigned long long)__popcll(m) + (unsigned long long)__popcll(mw);
    uint32_t nev = 0;                   // pair events of this lane
    uint32_t q_head = 0, q_tail = 0;    // wave-uniform: walkers taken / appended
    uint32_t n_dev = 0;                 // wave-uniform: deep END steps
    // the lane's walker
    bool wk = false;
    uint32_t ws = 0, wp = 0, wE = 0, wCF = 0;
    uint32_t iv_cnt = 0;  // intervals are appended to W->queue from the END backwards: queue[kQueue - 1 - i]
    for (int it = 0; it < 512; it++) {  // (bounded; ends when no lane has a candidate, a walker or a queued walker)
      const bool more = (m | mw) != 0ull || wk;
      if (!__builtin_amdgcn_ballot_w64(more) && q_head == q_tail) break;
      n_iter++;
      // ---- up to two candidates of this lane: pair probes
      uint32_t cq[2], cc1[2], cc2[2], cp[2];
      bool chave[2];
      uint4 pe[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const bool fromw = mw != 0ull;
        const unsigned long long cur = fromw ? mw : m;
        chave[u] = cur != 0ull;
        const uint32_t b = chave[u] ? (uint32_t)__builtin_ctzll(cur) : 0u;
        cq[u] = (fromw ? 0u : (uint32_t)(kWarm + lane * 64)) + b;
        if (chave[u]) {
          if (fromw) mw &= mw - 1; else m &= m - 1;
        }
        uint32_t L1, L2;
        char_at(cq[u], dend, cc1[u], L1);
        char_at(cq[u] + L1, dend, cc2[u], L2);
        cp[u] = cq[u] + L1 + L2;
        uint32_t h = mul24(cc2[u], P.k1) + rot11(mul24(cc1[u], kHK2));
        h ^= h >> 16;
        const uint32_t t = h * kHMix;
        const uint32_t d = dispb[(h >> 7) & gmask];
        const uint32_t sl = ((t >> psh) + d * ((t << 1) | 1u)) & pmask;
        pe[u] = P.pairs[chave[u] ? sl : 0u];
      }
      // ---- one step of this lane's walker
      uint32_t dc = 0, dL = 0;
      bool dgo = false;
      uint4 d1 = make_uint4(0, 0, 0, 0), d2 = d1;
      {
        char_at(min(wp, (uint32_t)(kRowBytes - 8)), dend, dc, dL);
        dgo = wk & wp < dend & ((wCF >> (mul24(dc, kHK2) >> 27)) & 1u) != 0u;
        const uint32_t B = wE & 0x3FFFFFu;
        uint32_t h = mul24(dc, P.k1) + (rot11(mul24(B, kHK2)) ^ kSalt);
        h ^= h >> 16;
        const uint32_t t = h * kHMix;
        d1 = P.deep[dgo ? (t >> dsh) : 0u];
        d2 = P.deep[dgo ? ((t * kHMix2) >> dsh) : 0u];
      }
      // ---- resolve the pairs: an END pair is an event of this lane; a pair that may go on joins the queue
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const bool hit = chave[u] & pe[u].x == (kHTag | cc1[u]) & (pe[u].y & 0xFFFFFFu) == cc2[u];
        const uint32_t E = pe[u].z, CF = pe[u].w;
        if (hit && (E >> 31)) {
          if (nev < (uint32_t)kEvPerLane) W->ev[lane * kEvPerLane + nev] = make_uint2(cp[u], cq[u] | (pe[u].y >> 24) << 16);
          else n_over++;
          nev++;
        }
        // may it go on?  (the next character's class against the child filter)
        uint32_t c3, L3;
        char_at(min(cp[u], (uint32_t)(kRowBytes - 8)), dend, c3, L3);
        const bool cont = hit & cp[u] < dend & ((CF >> (mul24(c3, kHK2) >> 27)) & 1u) != 0u;
        const uint64_t cm = __builtin_amdgcn_ballot_w64(cont);
        if (cm) {
          const uint32_t r = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
          if (cont) {
            if (r < (uint32_t)kQueue - iv_cnt) W->queue[r] = make_uint4(cq[u], cp[u], E, CF);
            else n_over++;
          }
          q_tail = min(q_tail + (uint32_t)__popcll(cm), (uint32_t)kQueue - iv_cnt);
        }
      }
      // ---- resolve the walker's step (votes outside the divergent part: the counters are wave-uniform)
      {
        const uint32_t B = wE & 0x3FFFFFu;
        const bool h1 = d1.x == B & (d1.y & 0xFFFFFFu) == dc, h2 = d2.x == B & (d2.y & 0xFFFFFFu) == dc;
        const bool hit = wk & dgo & (h1 | h2);
        if (hit) {
          wE = h1 ? d1.z : d2.z;
          wCF = h1 ? d1.w : d2.w;
          wp += dL;
        }
        const bool endstep = hit & (wE >> 31) != 0u;
        const uint64_t em = __builtin_amdgcn_ballot_w64(endstep);
        if (em) {
          const uint32_t r = n_dev + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
          if (endstep) {
            const uint32_t c4 = (h1 ? d1.y : d2.y) >> 24;
            if (r < (uint32_t)kDeepEv) W->dev[r] = make_uint2(wp, ws | c4 << 16);
            else n_over++;
          }
          n_dev = min(n_dev + (uint32_t)__popcll(em), (uint32_t)kDeepEv);
        }
        const bool done = wk & !hit;  // the walk ends: its interval (start, reach)
        const uint64_t dm = __builtin_amdgcn_ballot_w64(done);
        if (dm) {
          const uint32_t r = iv_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
          if (done) {
            if (r < (uint32_t)kQueue - q_tail) W->queue[kQueue - 1 - r] = make_uint4(ws, wp, 0u, 0u);
            else n_over++;
            wk = false;
          }
          iv_cnt = min(iv_cnt + (uint32_t)__popcll(dm), (uint32_t)kQueue - q_tail);
        }
      }
      // ---- free lanes take queued walkers
      {
        const uint64_t fm = __builtin_amdgcn_ballot_w64(!wk);
        const uint32_t r = q_head + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
        if (!wk && r < q_tail) {
          const uint4 e = W->queue[r];
          ws = e.x;
          wp = e.y;
          wE = e.z;
          wCF = e.w;
          wk = true;
          n_walk++;
        }
        q_head = min(q_tail, q_head + (uint32_t)__popcll(fm));
      }
    }
    // ---- count: an END step at offset j of start s is the reference's event unless a walker with an earlier start reaches j
    auto blocked = [&](uint32_t j, uint32_t s) {
      bool b = false;
      for (uint32_t i = 0; i < iv_cnt; i++) {
        const uint4 iv = W->queue[kQueue - 1 - i];
        b |= iv.x < s & iv.y >= j;
      }
      return b;
    };
    for (uint32_t k = 0; k < (uint32_t)kEvPerLane; k++) {
      const bool have = k < nev;
      if (!__builtin_amdgcn_ballot_w64(have)) break;
      const uint2 e = have ? W->ev[lane * kEvPerLane + k] : make_uint2(0, 0);
      const bool ok = have && e.x > (uint32_t)kWarm && e.x <= (uint32_t)(kWarm + kTile) && !blocked(e.x, e.y & 0xFFFFu);
      if (ok) {
        n_ev++;
        n_hit += e.y >> 16;
      }
    }
    for (uint32_t i0 = 0; i0 < n_dev; i0 += 64) {
      const bool have = i0 + lane < n_dev;
      const uint2 e = have ? W->dev[i0 + lane] : make_uint2(0, 0);
      const bool ok = have && e.x > (uint32_t)kWarm && e.x <= (uint32_t)(kWarm + kTile) && !blocked(e.x, e.y & 0xFFFFu);
      if (ok) {
        n_ev++;
        n_hit += e.y >> 16;
      }
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    n_ev += __shfl_xor(n_ev, d, 64);
    n_hit += __shfl_xor(n_hit, d, 64);
    n_over += __shfl_xor(n_over, d, 64);
    n_cand += __shfl_xor(n_cand, d, 64);
    n_walk += __shfl_xor(n_walk, d, 64);
  }
  if (lane == 0) {
    atomicAdd(&P.out[0], n_ev);
    atomicAdd(&P.out[1], n_hit);
    atomicAdd(&P.out[2], n_over);
    atomicAdd(&P.out[3], n_cand);
    atomicAdd(&P.out[4], n_walk);
    atomicAdd(&P.out[5], n_iter);
  }
}

extern "C" int proto_k3_run(const uint8_t *text, uint64_t n_bytes, const uint64_t *bitmap, const uint8_t *disp, const void *pairs,
                            const void *deep, uint32_t n_groups, uint32_t pair_log2, uint32_t deep_log2, uint32_t k1,
                            uint32_t max_len, unsigned long long *out, int grid, int reps, float *ms_out) {
  K3P P{text, n_bytes, bitmap, disp, (const uint4 *)pairs, (const uint4 *)deep, n_groups, pair_log2, deep_log2, k1, max_len, out};
  const size_t lds = ((n_groups + 15u) & ~15u) + 256 + (size_t)(kThreads / 64) * sizeof(WaveLds) + 64;
  if (hipFuncSetAttribute((const void *)k3_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    (void)hipMemsetAsync(out, 0, 8 * 8, 0);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k3_walk, dim3(grid), dim3(kThreads), lds, 0, P);
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  *ms_out = best;
  return (int)hipGetLastError();
}
