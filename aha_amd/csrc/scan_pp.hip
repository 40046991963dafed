// scan_pp.hip -- position-parallel match engine for gfx950 (MI355X); see pp.hpp for the exactness
// argument.  Replaces the same reference path as scan_v2.hip (src/aha/ac.cr:176-192 match_,
// src/aha/cedar.cr:441-447 child, :657-660 is_end?) for automata that meet pp.hpp's preconditions:
//
//   k_pp_filter   every start position is classified independently: a direct-table lookup on its two
//                 bytes, then (through a wave-private LDS ring that compacts the live starts into full
//                 64-lane batches) three Bloom probes on its 3 / 4 / 5 bytes.  All tables live in LDS;
//                 the input is read once with coalesced 16-byte loads and stays in registers.  Starts
//                 that are not proven boring become 16-bit items of their 4 KiB chunk.
//   k_pp_walk / k_pp_deep / k_pp_order   the exact pass: trie walks of the items, the boring starts right in
//                 front of an END re-walked exactly, long walks finished in a kernel of their own, events
//                 written in position order into the chunk's region {state base, end offset in the document}.
// The records, the per-chunk counts and the per-document ranks are those of k2_traverse's region
// pipeline, so k2d_count / scan / k2d_expand / k2d_doc_offsets (scan_v2.hip) finish the call.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"
#include "pp.hpp"

namespace aha {

namespace {

constexpr int kFWaves = 16;                 // waves per filter workgroup (one workgroup per CU)
constexpr int kFThreads = kFWaves * 64;
constexpr int kFTile = 1024;                // bytes per wave step (64 lanes x 16 B)
constexpr uint32_t kFRing = 128;            // ring entries per wave (8 B each): at most 63 + 64 are pending

// ------------------------------------------------------------------ pass 1
__global__ __launch_bounds__(kFThreads) void k_pp_filter(PpArgs P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *t2 = reinterpret_cast<uint32_t *>(smem);
  uint32_t *bl = t2 + kPpT2Words;
  uint2 *rings = reinterpret_cast<uint2 *>(bl + P.b_words);
  uint16_t *outs = reinterpret_cast<uint16_t *>(rings + kFWaves * kFRing);
  for (uint32_t i = threadIdx.x; i < kPpT2Words; i += kFThreads) t2[i] = P.t2[i];
  for (uint32_t i = threadIdx.x; i < P.b_words; i += kFThreads) bl[i] = P.bloom[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint2 *ring = rings + wave * kFRing;
  uint16_t *obuf = outs + wave * kPpItemCap;
  const uint32_t b_scale = (P.b_words << 8) & 0xFFFFFFu;
  const uint64_t N = P.n_bytes;
  const uint64_t wave_id = (uint64_t)blockIdx.x * kFWaves + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * kFWaves;

  for (uint64_t chunk = wave_id; chunk < P.n_chunks; chunk += n_waves) {
    const uint64_t c0 = chunk * kPpChunk;
    uint32_t head = 0, tail = 0;  // ring cursors (wave uniform)
    uint32_t ocnt = 0;            // items in obuf (wave uniform)
    unsigned long long tile_end = 0;  // 4 x 16 bit: items up to the end of each 1 KiB tile
    auto load16 = [&](uint64_t g) -> uint4 {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g + 16 <= N) {
        v = *reinterpret_cast<const uint4 *>(P.text + g);
      } else if (g < N) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16 && g + j < N; j++) w[j >> 2] |= (uint32_t)P.text[g + j] << ((j & 3) * 8);
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      return v;
    };
    // one batch of up to 64 ring items: Bloom probes, survivors appended to obuf
    auto batch = [&](uint32_t navail) {
      const bool valid = (uint32_t)lane < navail;
      const uint2 it = ring[(head + lane) & (kFRing - 1)];
      const uint32_t lo = it.x, hi = it.y;
      const PpHash h = pp_hash(lo, hi);
      const uint32_t g = h.m1 ^ (h.m1 >> 11);
      const uint32_t bm = __builtin_amdgcn_perm(0x80402010u, 0x08040201u, g & 0x07070707u);  // = pp_mask(h.m1)
      const uint32_t w3 = bl[(uint32_t)(((uint64_t)(h.h3 >> 8) * b_scale) >> 32)];
      const uint32_t w4 = bl[(uint32_t)(((uint64_t)(h.h4 >> 8) * b_scale) >> 32)];
      const uint32_t w5 = bl[(uint32_t)(((uint64_t)(h.h5 >> 8) * b_scale) >> 32)];
      const bool deep = (hi >> 30) & 1u;
      const bool p3 = deep && (w3 & bm) == bm, p4 = deep && (w4 & bm) == bm, p5 = deep && (w5 & bm) == bm;
      const bool end2 = (hi >> 31) & 1u;
      const bool keep = valid && (p3 || p4 || p5 || end2);
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        const uint32_t my =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, ocnt));
        if (my < kPpItemCap)
          obuf[my] = (uint16_t)(((hi >> 8) & 0xFFFu) | (p3 ? kPpItemE3 : 0u) | (p4 ? kPpItemE4 : 0u) |
                                (p5 ? kPpItemP5 : 0u) | (end2 ? kPpItemEnd2 : 0u));
      }
      ocnt += (uint32_t)__popcll(mask);
      head += navail;
    };

    uint4 cur = load16(c0 + (uint64_t)lane * 16);
    for (uint32_t t = 0; t < kPpChunk / kFTile; t++) {
      const uint64_t t0 = c0 + (uint64_t)t * kFTile;
      if (t0 >= N) break;
      const uint4 nxt = load16(t0 + kFTile + (uint64_t)lane * 16);  // next tile; its first bytes are lane 63's halo
      uint32_t n0 = __shfl_down(cur.x, 1, 64), n1 = __shfl_down(cur.y, 1, 64);
      const uint32_t x0 = __builtin_amdgcn_readfirstlane(nxt.x), x1 = __builtin_amdgcn_readfirstlane(nxt.y);
      if (lane == 63) {
        n0 = x0;
        n1 = x1;
      }
      const uint32_t d[6] = {cur.x, cur.y, cur.z, cur.w, n0, n1};
      const uint32_t posbase = (t * kFTile + lane * 16) << 8;
      // phase A: the lane's 16 T2 lookups (independent LDS reads), codes packed 2 bits per position
      uint32_t codes = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int q = k >> 2, r = k & 3;
        const uint32_t lo = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], r) : d[q];
        const uint32_t word = t2[lo & 0xFFFu];  // = t2[pp_t2_word(b0, b1)]
        codes |= ((word >> ((lo >> 11) & 30u)) & 3u) << (2 * k);
      }
      // phase B: live starts go to the ring; a full batch of 64 runs the Bloom probes
#pragma unroll
      for (int k = 0; k < 16; k++) {
        const int q = k >> 2, r = k & 3;
        const uint32_t code = (codes >> (2 * k)) & 3u;
        const bool live = code != 0;
        const unsigned long long m = __ballot(live);
        if (live) {
          const uint32_t lo = r ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], r) : d[q];
          const uint32_t hi = r ? __builtin_amdgcn_alignbyte(d[q + 2], d[q + 1], r) : d[q + 1];
          const uint32_t idx =
              __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail)) &
              (kFRing - 1);
          // {bytes 0..3, byte 4 | position << 8 | deep << 30 | end2 << 31}
          ring[idx] = make_uint2(lo, (hi & 0xFFu) | (posbase + ((uint32_t)k << 8)) | (code << 30));
        }
        tail += (uint32_t)__popcll(m);
        if (tail - head >= 64) batch(64);
      }
      // the items of a tile stay together: pass 2 resolves tile by tile
      if (tail != head) batch(tail - head);
      tile_end |= (unsigned long long)min(ocnt, kPpItemCap) << (16 * t);
      cur = nxt;
    }
    for (uint32_t t = (uint32_t)((min(c0 + kPpChunk, N) - c0 + kFTile - 1) / kFTile); t < kPpChunk / kFTile; t++)
      tile_end |= (unsigned long long)min(ocnt, kPpItemCap) << (16 * t);
    const uint32_t nw = min(ocnt, kPpItemCap);
    uint16_t *dst = P.items + chunk * kPpItemCap;
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = obuf[i];
    if (lane == 0) {
      P.tile_end[chunk] = tile_end;
      if (ocnt > kPpItemCap) P.flags[1] = 3ull;  // hit-dense input: the host takes the single-traversal engine
    }
  }
}

// ------------------------------------------------------------------ pass 2
// Three kernels, none of which waits for a long dependent chain in lockstep:
//   k_pp_walk   one lane per item, workgroups that keep the shallow part of the image in LDS.  The walk from the
//               root is taken to depth kWalkCap; END nodes found on the way become raw candidates unless one of
//               the (at most two) starts right in front is still alive at that position (walks of those starts
//               are at most 4 long when they matter here, so they are re-walked exactly, mostly in LDS).  A walk
//               still alive at depth kWalkCap goes to the deep list.
//   k_pp_deep   one lane per deep entry, full occupancy: the rest of the walk in HBM/L2; more raw candidates, and
//               the start's reach goes to the long list of the chunk(s) it covers.
//   k_pp_order  one wave per chunk: drops the raw candidates covered by a long start in front of them, sorts the
//               rest by position (bitmap ranks) and rewrites the chunk's region as final event records.
constexpr int kWWaves = 16;
constexpr int kWThreads = kWWaves * 64;
constexpr uint32_t kWalkCap = kPpGuard;  // lockstep depth of k_pp_walk; walks alive here are "long" (L >= kPpGuard)
constexpr int kWUnroll = 4;              // items per lane per round (their probes are in flight together)
constexpr uint32_t kRBnd = 4;            // document boundaries kept in registers per chunk window

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in order; this only keeps the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// up to 8 text bytes at g (little endian); bytes at and beyond N read as 0
__device__ __forceinline__ uint64_t load8(const uint8_t *text, uint64_t g, uint64_t N) {
  uint64_t v = 0;
  if (g + 8 <= N) {
    __builtin_memcpy(&v, text + g, 8);
  } else {
    for (int j = 0; j < 8 && g + j < N; j++) v |= (uint64_t)text[g + j] << (8 * j);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_pp_chunk_doc(const uint64_t *doc_off, uint64_t D, uint64_t n_chunks,
                                                       uint32_t *out) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t cs = c * kPpChunk, lo = cs >= kPpHalo ? cs - kPpHalo : 0;
  out[2 * c] = (uint32_t)first_boundary(doc_off, D, cs);
  out[2 * c + 1] = (uint32_t)first_boundary(doc_off, D, lo + 1);
}

// Document boundaries q of a chunk's window, lo < q <= hi (lo = cs - kPpHalo or 0), wave uniform in registers.
struct ChunkDocs {
  uint64_t bq[kRBnd];
  uint64_t ds0, hi;  // start of the document that covers lo; end of the window
  bool slow;         // more than kRBnd boundaries: every query searches doc_off
  const uint64_t *doc_off;
  uint64_t D;
  __device__ __forceinline__ void init(const V2Args &M, const PpArgs &P, uint64_t c, int lane) {
    doc_off = M.doc_off;
    D = M.n_docs;
    const uint64_t cs = c * kPpChunk, ce = min(cs + kPpChunk, M.n_bytes);
    hi = min(ce + kPpHalo, M.n_bytes);
    const uint64_t dnA = P.chunk_doc[2 * c + 1];  // first document that starts after lo (>= 1: doc_off[0] = 0)
    const uint64_t dn = dnA + (uint64_t)lane;
    const uint64_t q = (lane <= (int)kRBnd && dn <= D) ? doc_off[dn] : ~0ull;
    const unsigned long long inm = __ballot(q <= hi);
    const uint32_t nb = (uint32_t)__popcll(inm & ((1ull << kRBnd) - 1));
    slow = (inm >> kRBnd) & 1ull;
#pragma unroll
    for (int k = 0; k < (int)kRBnd; k++) {
      const uint64_t qk = __shfl(q, k, 64);
      bq[k] = (uint32_t)k < nb ? qk : ~0ull;
    }
    ds0 = doc_off[dnA - 1];
  }
  // end (exclusive) of the document of absolute position g, clamped to hi
  __device__ __forceinline__ uint64_t end_of(uint64_t g) const {
    if (slow) return min(doc_off[first_boundary(doc_off, D, g + 1)], hi);
    uint64_t e = hi;
#pragma unroll
    for (int k = (int)kRBnd - 1; k >= 0; k--) e = bq[k] > g && bq[k] < e ? bq[k] : e;
    return e;
  }
  __device__ __forceinline__ uint64_t start_of(uint64_t g) const {
    if (slow) return doc_off[first_boundary(doc_off, D, g + 1) - 1];
    uint64_t s = ds0;
#pragma unroll
    for (int k = 0; k < (int)kRBnd; k++) s = bq[k] <= g ? bq[k] : s;
    return s;
  }
};

// appends the raw candidate {state base | depth << 22, position in the chunk} to the region of the END's chunk
__device__ __forceinline__ void push_cand(const V2Args &M, uint64_t i_abs, uint32_t base, uint32_t depth) {
  const uint64_t ci = i_abs / kPpChunk;
  const uint32_t k = atomicAdd(&M.ev_cnt[ci], 1u);
  if (k < M.ev_stride) M.evd[ci * M.ev_stride + k] = make_uint2(base | (depth << 22), (uint32_t)(i_abs - ci * kPpChunk));
}
constexpr uint32_t kWTextPad = 16;                                     // zero bytes in front of the staged window
constexpr uint32_t kWText = kWTextPad + kPpHalo + kPpChunk + kPpHalo + 16;  // staged text per wave (LDS)

__global__ __launch_bounds__(kWThreads) void k_pp_walk(DevAut A, V2Args M, PpArgs P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *lt = reinterpret_cast<uint32_t *>(smem);
  const uint32_t *gt = reinterpret_cast<const uint32_t *>(A.slots);
  const uint32_t T = P.lds_slots;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(gt);
    uint4 *dst = reinterpret_cast<uint4 *>(lt);
    for (uint32_t i = threadIdx.x; i < T / 4; i += kWThreads) dst[i] = src[i];
  }
  __syncthreads();  // the only workgroup barrier: every wave is on its own from here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *tb = smem + (size_t)T * 4 + (size_t)wave * kWText;  // this wave's text window
  const uint32_t *tbw = reinterpret_cast<const uint32_t *>(tb);
  const uint64_t N = M.n_bytes;
  const uint32_t Lmax = A.max_len;
  const uint64_t wave_id = (uint64_t)blockIdx.x * kWWaves + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * kWWaves;
  auto probe = [&](uint32_t idx) -> uint32_t {
    uint32_t e;
    if (idx < T)
      e = lt[idx];
    else
      e = gt[idx];
    return e;
  };
  if (__builtin_amdgcn_readfirstlane((uint32_t)M.cursor[1]) != 0) return;  // pass 1 overflowed
  if (lane < (int)(kWTextPad / 4)) reinterpret_cast<uint32_t *>(tb)[lane] = 0;

  for (uint64_t c = wave_id; c < M.n_chunks; c += n_waves) {
    const uint64_t te64 = P.tile_end[c];
    const uint32_t n_own = (uint32_t)(te64 >> 48);
    // items of the previous chunk's last tile: those in its last kPpHalo bytes may end inside this chunk
    uint32_t hb = 0, he = 0;
    if (c) {
      const uint64_t pe64 = P.tile_end[c - 1];
      hb = (uint32_t)(pe64 >> 32) & 0xFFFFu;
      he = (uint32_t)(pe64 >> 48);
    }
    if (n_own == 0 && hb == he) continue;  // ev_cnt[c] stays 0 (zeroed by the host)
    const uint64_t cs = c * kPpChunk, ce = min(cs + kPpChunk, N);
    // ---- stage text[wb .. we) in LDS with coalesced loads (one exposed memory latency per chunk)
    const uint64_t wb = cs >= kPpHalo ? cs - kPpHalo : 0, we = min(ce + kPpHalo, N);
    const uint32_t wl = (uint32_t)(we - wb);
    for (uint32_t i = lane * 16; i < wl + 16; i += 64 * 16) {
      const uint64_t gg = wb + i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gg + 16 <= N) {
        v = *reinterpret_cast<const uint4 *>(M.text + gg);
      } else if (gg < N) {
        uint32_t ww[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16 && gg + j < N; j++) ww[j >> 2] |= (uint32_t)M.text[gg + j] << ((j & 3) * 8);
        v = make_uint4(ww[0], ww[1], ww[2], ww[3]);
      }
      *reinterpret_cast<uint4 *>(tb + kWTextPad + i) = v;
    }
    ChunkDocs docs;
    docs.init(M, P, c, lane);
    const uint16_t *own = P.items + c * kPpItemCap;
    const uint16_t *prev = P.items + (c ? c - 1 : 0) * kPpItemCap;
    uint2 *reg = M.evd + c * M.ev_stride;
    uint32_t cnt = 0;  // raw candidates of this chunk so far (wave uniform): only this wave writes the region now
    uint32_t dcnt = 0; // long starts of this chunk so far
    // one virtual list: the previous chunk's last tile (its items in the last kPpHalo bytes count), then the own items
    const uint32_t n_halo = he - hb, n_all = n_halo + n_own;
    const uint32_t n_rounds = (n_all + 64 * kWUnroll - 1) / (64 * kWUnroll);
    wave_sync();
    for (uint32_t rd = 0; rd < n_rounds; rd++) {
      bool live[kWUnroll], halo[kWUnroll];
      uint64_t g[kWUnroll], V[kWUnroll];        // V = text[g-2 .. g+6)
      uint32_t lim[kWUnroll], nback[kWUnroll];
      uint32_t B[kWUnroll][3], L[kWUnroll][3];  // walks from g (0), g-1 (1), g-2 (2)
      bool alive[kWUnroll][3];
      uint32_t eb[kWUnroll][kWalkCap + 1];      // state base of the END node of the walk from g at each depth
#pragma unroll
      for (int u = 0; u < kWUnroll; u++) {
        const uint32_t r = (rd * kWUnroll + u) * 64 + lane;
        halo[u] = r < n_halo;
        live[u] = r < n_all;
        uint32_t pos = 0;
        if (live[u]) pos = (halo[u] ? prev[hb + r] : own[r - n_halo]) & 0xFFFu;
        if (halo[u]) {
          live[u] = pos >= kPpChunk - kPpHalo;
          g[u] = live[u] ? cs - kPpChunk + pos : cs;  // c >= 1 here
        } else {
          g[u] = cs + pos;
        }
      }
#pragma unroll
      for (int u = 0; u < kWUnroll; u++) {
        const uint32_t a = kWTextPad + (uint32_t)(g[u] - wb) - 2;  // g - 2 in the staged window (the pad is zero)
        const uint32_t d0 = tbw[a >> 2], d1 = tbw[(a >> 2) + 1], d2 = tbw[(a >> 2) + 2];
        const uint32_t sh = (a & 3u) * 8u;
        V[u] = ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | __builtin_amdgcn_alignbit(d1, d0, sh);
        const uint64_t dend = docs.end_of(g[u]), dstart = docs.start_of(g[u]);
        lim[u] = live[u] ? (uint32_t)min<uint64_t>(dend - g[u], Lmax) : 0u;
        nback[u] = live[u] ? (uint32_t)min<uint64_t>(g[u] - dstart, 2) : 0u;  // starts in front, same document
#pragma unroll
        for (int k = 0; k < 3; k++) {
          B[u][k] = A.root;
          L[u][k] = 0;
          alive[u][k] = live[u] && (uint32_t)k <= nback[u];
        }
#pragma unroll
        for (int d = 0; d <= (int)kWalkCap; d++) eb[u][d] = 0;
      }
      // ---- lockstep walks (goto probes from the root, cedar.cr:441-447): from g to depth kWalkCap, and from the two
      // starts in front of it to depth kPpGuard - 1 (boring starts have no longer walks; long ones are k_pp_order's)
#pragma unroll
      for (int dpt = 1; dpt <= (int)kWalkCap; dpt++) {
#pragma unroll
        for (int u = 0; u < kWUnroll; u++) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            if (k > 0 && dpt > (int)kPpGuard - 1) continue;
            const uint32_t b = (uint32_t)(V[u] >> (8 * (2 - k + dpt - 1))) & 0xFFu;  // text[g - k + dpt - 1]
            const bool in = alive[u][k] && (uint32_t)dpt <= lim[u] + k && b != 0;      // keys hold no NUL (cedar.cr:235)
            uint32_t e = 0;
            if (dpt <= 2) {  // the root's and the depth-1 states' rows are always in the LDS prefix (host check)
              if (in) e = lt[B[u][k] ^ b];
            } else {
              if (in) e = probe(B[u][k] ^ b);
            }
            alive[u][k] = in && (e & 0xFFu) == b;
            if (alive[u][k]) {
              B[u][k] = (e >> C_BASE_SHIFT) & C_BASE_MASK;
              L[u][k] = dpt;
              if (k == 0 && (e & C_END)) eb[u][dpt] = B[u][0] | 0x80000000u;
            }
          }
        }
      }
      // ---- END nodes -> raw candidates of this chunk.  An END at depth d reports unless an earlier start of the
      // document is still alive at its last byte: the start k bytes in front covers it iff its walk has k + d bytes
      // (short starts: decided here, exactly; starts with walks of kPpGuard bytes and more: k_pp_order).
      uint32_t keep[kWUnroll];
      uint32_t mine = 0;
#pragma unroll
      for (int u = 0; u < kWUnroll; u++) {
        keep[u] = 0;
#pragma unroll
        for (int d = 2; d <= (int)kWalkCap; d++) {
          const uint64_t i_abs = g[u] + d - 1;
          bool ok = eb[u][d] != 0 && i_abs >= cs && i_abs < ce;
          ok = ok && !(L[u][1] >= (uint32_t)d + 1) && !(L[u][2] >= (uint32_t)d + 2);
          keep[u] |= ok ? 1u << d : 0u;
        }
        mine += __popc(keep[u]);
      }
      {
        const uint32_t inc = wave_incl_scan(mine);
        uint32_t wpos = cnt + inc - mine;
        cnt += __shfl(inc, 63, 64);
#pragma unroll
        for (int u = 0; u < kWUnroll; u++) {
#pragma unroll
          for (int d = 2; d <= (int)kWalkCap; d++) {
            if (keep[u] & (1u << d)) {
              if (wpos < M.ev_stride)
                reg[wpos] = make_uint2((eb[u][d] & C_BASE_MASK) | ((uint32_t)d << 22), (uint32_t)(g[u] + d - 1 - cs));
              wpos++;
            }
          }
        }
      }
      // ---- long starts of this chunk: the deep pass finishes the walk and publishes the reach.  The chunk's own
      // list, filled by this wave alone (one global cursor for all waves serialises at ~11 ns per atomic).
#pragma unroll
      for (int u = 0; u < kWUnroll; u++) {
        const bool deep = !halo[u] && live[u] && L[u][0] == kWalkCap;
        const unsigned long long dm = __ballot(deep);
        if (dm) {
          if (deep) {
            const uint32_t k = dcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
            if (k < kPpDeepCap)
              P.deep[c * kPpDeepCap + k] = make_uint2((uint32_t)(g[u] - cs) | (lim[u] << 16), B[u][0]);
          }
          dcnt += (uint32_t)__popcll(dm);
        }
      }
    }
    if (lane == 0) {
      M.ev_cnt[c] = min(cnt, M.ev_stride);  // k_pp_deep adds to it
      P.deep_cnt[c] = min(dcnt, kPpDeepCap);
      if (cnt > M.ev_stride || dcnt > kPpDeepCap) M.cursor[1] = 3ull;
    }
    wave_sync();  // the next chunk overwrites the staged text
  }
}

__global__ __launch_bounds__(256) void k_pp_deep(DevAut A, V2Args M, PpArgs P) {
  const uint32_t *gt = reinterpret_cast<const uint32_t *>(A.slots);
  const uint64_t N = M.n_bytes;
  if (M.cursor[1] != 0) return;
  static_assert(kPpDeepCap == 64, "one wave per chunk list");
  const uint64_t n = M.n_chunks * kPpDeepCap;
  for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) {
    const uint64_t c = k / kPpDeepCap;
    if ((uint32_t)(k % kPpDeepCap) >= P.deep_cnt[c]) continue;
    const uint2 it = P.deep[k];
    const uint64_t g = c * kPpChunk + (it.x & 0xFFFFu);
    const uint32_t lim = it.x >> 16;
    uint32_t B = it.y, L = kWalkCap;
    uint64_t w = 0;
    for (uint32_t dpt = kWalkCap + 1; dpt <= lim; dpt++) {
      if (((dpt - kWalkCap - 1) & 7u) == 0) w = load8(M.text, g + dpt - 1, N);
      const uint32_t b = (uint32_t)w & 0xFFu;
      w >>= 8;
      if (b == 0) break;
      const uint32_t e = gt[B ^ b];
      if ((e & 0xFFu) != b) break;
      B = (e >> C_BASE_SHIFT) & C_BASE_MASK;
      L = dpt;
      if (e & C_END) push_cand(M, g + dpt - 1, B, dpt);  // no short start can cover an END this deep
    }
    // reach of a long start: to its own chunk's list, and to the next chunk's when it covers bytes of it
    for (uint64_t ck = c; ck <= (g + L - 1) / kPpChunk && ck < M.n_chunks; ck++) {
      const uint32_t q = atomicAdd(&P.long_cnt[ck], 1u);
      if (q < kPpLongCap) P.longs[ck * kPpLongCap + q] = (uint32_t)(g + kPpHalo - ck * kPpChunk) | (L << 16);
    }
  }
}

struct OScratch {
  uint32_t ebm[kPpChunk / 32];
  uint16_t epre[kPpChunk / 32];
  uint32_t longs[kPpLongCap];
};

__global__ __launch_bounds__(256) void k_pp_order(DevAut A, V2Args M, PpArgs P) {
  __shared__ OScratch Sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  OScratch &S = Sh[wave];
  const uint64_t N = M.n_bytes, D = M.n_docs;
  if (__builtin_amdgcn_readfirstlane((uint32_t)M.cursor[1]) != 0) return;
  constexpr int kPer = kPpEvStride / 64;  // raw candidates per lane
  for (uint64_t c = (uint64_t)blockIdx.x * 4 + wave; c < M.n_chunks; c += (uint64_t)gridDim.x * 4) {
    const uint64_t cs = c * kPpChunk, ce = min(cs + kPpChunk, N);
    const uint32_t n_raw = M.ev_cnt[c], n_long = P.long_cnt[c];
    if (n_raw > M.ev_stride || n_long > kPpLongCap) {
      if (lane == 0) M.cursor[1] = 3ull;  // hit-dense or deeply nested: the host takes the single-traversal engine
      continue;
    }
    uint2 *reg = M.evd + c * M.ev_stride;
    const uint64_t dnc = P.chunk_doc[2 * c];  // first document that starts at or after cs
    uint32_t n_ev = 0;
    if (n_raw) {
      ChunkDocs docs;
      docs.init(M, P, c, lane);
      for (uint32_t k = lane; k < kPpChunk / 32; k += 64) S.ebm[k] = 0;
      for (uint32_t k = lane; k < n_long; k += 64) S.longs[k] = P.longs[c * kPpLongCap + k];
      wave_sync();
      uint2 cd[kPer];
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        const uint32_t k = u * 64 + lane;
        cd[u] = k < n_raw ? reg[k] : make_uint2(0u, 0xFFFFFFFFu);
        if (k < n_raw) {
          const int32_t i = (int32_t)cd[u].y, d = (int32_t)(cd[u].x >> 22), j = i - d + 1;
          bool ok = true;
          for (uint32_t q = 0; q < n_long; q++) {  // a long start in front of it that is still alive at i
            const uint32_t e = S.longs[q];
            const int32_t jl = (int32_t)(e & 0xFFFFu) - (int32_t)kPpHalo, ll = (int32_t)(e >> 16);
            ok = ok && !(jl < j && jl + ll > i);
          }
          if (ok)
            atomicOr(&S.ebm[i >> 5], 1u << (i & 31));
          else
            cd[u].y = 0xFFFFFFFFu;
        }
      }
      wave_sync();
      // exclusive prefix of the popcounts (128 words: two per lane)
      {
        const uint32_t a = __popc(S.ebm[2 * lane]), b = __popc(S.ebm[2 * lane + 1]);
        const uint32_t inc = wave_incl_scan(a + b);
        n_ev = __shfl(inc, 63, 64);
        S.epre[2 * lane] = (uint16_t)(inc - a - b);
        S.epre[2 * lane + 1] = (uint16_t)(inc - b);
      }
      wave_sync();
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        if (cd[u].y == 0xFFFFFFFFu) continue;
        const uint32_t i = cd[u].y;
        const uint32_t rank = S.epre[i >> 5] + __popc(S.ebm[i >> 5] & ((1u << (i & 31)) - 1u));
        const uint64_t gi = cs + i;
        reg[rank] = make_uint2(cd[u].x & C_BASE_MASK, (uint32_t)(gi + 1 - docs.start_of(gi)));
      }
    }
    if (lane == 0) M.ev_cnt[c] = n_ev;
    // documents that start inside the chunk: events of the chunk before the document start
    for (uint64_t d0 = dnc;; d0 += 64) {
      const uint64_t dn = d0 + (uint64_t)lane;
      const uint64_t q = dn <= D ? M.doc_off[dn] : ~0ull;
      const bool in = q < ce;
      if (in) {
        const uint32_t qr = (uint32_t)(q - cs);
        M.doc_ev_rank[dn] = n_raw ? S.epre[qr >> 5] + __popc(S.ebm[qr >> 5] & ((1u << (qr & 31)) - 1u)) : 0u;
      }
      if (__popcll(__ballot(in)) < 64) break;
    }
    wave_sync();
  }
}

}  // namespace

size_t pp_filter_lds(uint32_t b_words) {
  return (size_t)kPpT2Words * 4 + (size_t)b_words * 4 + (size_t)kFWaves * kFRing * 8 + (size_t)kFWaves * kPpItemCap * 2;
}

size_t pp_walk_lds(uint32_t lds_slots) { return (size_t)lds_slots * 4 + (size_t)kWWaves * kWText; }
uint32_t pp_walk_max_slots() { return (uint32_t)((160 * 1024 - kWWaves * kWText) / 4) & ~3u; }

int pp_prepare(uint32_t b_words, uint32_t lds_slots) {
  int e = (int)hipFuncSetAttribute((const void *)k_pp_filter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)pp_filter_lds(b_words));
  if (e) return e;
  return (int)hipFuncSetAttribute((const void *)k_pp_walk, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pp_walk_lds(lds_slots));
}

void pp_launch_filter(const PpArgs &P, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(k_pp_filter, dim3(grid), dim3(kFThreads), pp_filter_lds(P.b_words), (hipStream_t)stream, P);
}

void pp_launch_resolve(const DevAut &A, const V2Args &M, const PpArgs &P, uint32_t grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pp_chunk_doc, dim3((uint32_t)((M.n_chunks + 255) / 256)), dim3(256), 0, s, M.doc_off, M.n_docs,
                     M.n_chunks, P.chunk_doc);
  hipLaunchKernelGGL(k_pp_walk, dim3(grid), dim3(kWThreads), pp_walk_lds(P.lds_slots), s, A, M, P);
  const dim3 gc((uint32_t)std::min<uint64_t>((M.n_chunks + 3) / 4, 1u << 16));
  hipLaunchKernelGGL(k_pp_deep, gc, dim3(256), 0, s, A, M, P);
  hipLaunchKernelGGL(k_pp_order, gc, dim3(256), 0, s, A, M, P);
}

}  // namespace aha
