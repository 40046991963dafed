// scan_pp.hip -- position-parallel match engine for gfx950 (MI355X); see pp.hpp for the exactness
// argument.  Replaces the same reference path as scan_v2.hip (src/aha/ac.cr:176-192 match_,
// src/aha/cedar.cr:441-447 child, :657-660 is_end?) for automata that meet pp.hpp's preconditions:
//
//   k_pp_filter   every start position is classified independently: a direct-table lookup on its two
//                 bytes, then (through a wave-private LDS ring that compacts the live starts into full
//                 64-lane batches) three Bloom probes on its 3 / 4 / 5 bytes.  All tables live in LDS;
//                 the input is read once with coalesced 16-byte loads and stays in registers.  Starts
//                 that are not proven boring become 16-bit items of their 4 KiB chunk.
//   k_pp_resolve  one workgroup per chunk: exact trie walks of the items (and of the items in the
//                 256-byte halo in front of the chunk), prefix maximum of their reaches, exact check
//                 of the boring starts right in front of a candidate, events written in position
//                 order into the chunk's region {state base, end offset in the document}.
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
// Persistent 1024-thread workgroups like k2_traverse: the first lds_slots slots of the image (the shallow, hot
// states) are copied into LDS once; every wave then resolves whole chunks on its own, tile by tile (a tile is the
// 1 KiB piece whose items pass 1 appended contiguously), with wave-private scratch and no workgroup barrier.
constexpr int kRWaves = 16;
constexpr int kRThreads = kRWaves * 64;
constexpr uint32_t kRHalo = 256;                        // bytes in front of a tile whose items are re-walked
constexpr uint32_t kRItems = 384;                       // items of a tile + its halo
constexpr uint32_t kRCands = 192;                       // END nodes that end inside the tile
constexpr uint32_t kRIWords = (kFTile + kRHalo) / 32;   // item bitmap words (halo + tile)
constexpr uint32_t kREWords = kFTile / 32;              // event bitmap words
constexpr uint32_t kRBnd = 4;                           // document boundaries kept in registers per chunk window

struct WScratch {
  uint2 cand[kRCands];       // x = state base, y = end position (window relative) | item rank << 16
  uint16_t ipos[kRItems];    // item positions (window relative), ascending
  uint16_t icov[kRItems];    // max over earlier items of (position + reach)
  uint8_t ireach[kRItems];
  uint32_t ibm[kRIWords];    // item bitmap over [base0, te)
  uint32_t ebm[kREWords];    // event bitmap over [ts, te)
  uint16_t wpre[kRIWords];   // exclusive prefix of popcounts of ibm
  uint16_t epre[kREWords];
  uint32_t n_cands;
  uint32_t pad[3];
};

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
  const uint64_t cs = c * kPpChunk, lo = cs >= kRHalo ? cs - kRHalo : 0;
  out[2 * c] = (uint32_t)first_boundary(doc_off, D, cs);
  out[2 * c + 1] = (uint32_t)first_boundary(doc_off, D, lo + 1);
}

__global__ __launch_bounds__(kRThreads) void k_pp_resolve(DevAut A, V2Args M, PpArgs P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *lt = reinterpret_cast<uint32_t *>(smem);
  const uint32_t *gt = reinterpret_cast<const uint32_t *>(A.slots);
  const uint32_t T = P.lds_slots;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(gt);
    uint4 *dst = reinterpret_cast<uint4 *>(lt);
    for (uint32_t i = threadIdx.x; i < T / 4; i += kRThreads) dst[i] = src[i];
  }
  __syncthreads();  // the only workgroup barrier: every wave is on its own from here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WScratch &S = *reinterpret_cast<WScratch *>(smem + (size_t)T * 4 + (size_t)wave * sizeof(WScratch));
  const uint64_t N = M.n_bytes, D = M.n_docs;
  const uint32_t Lmax = A.max_len;
  const uint64_t wave_id = (uint64_t)blockIdx.x * kRWaves + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * kRWaves;
  auto probe = [&](uint32_t idx) -> uint32_t {
    uint32_t e;
    if (idx < T)
      e = lt[idx];
    else
      e = gt[idx];
    return e;
  };
  // pass 1 overflowed somewhere: the host discards everything (read once per wave: uniform in the wave)
  if (__builtin_amdgcn_readfirstlane((uint32_t)M.cursor[1]) != 0) return;

  for (uint64_t c = wave_id; c < M.n_chunks; c += n_waves) {
    const uint64_t cs = c * kPpChunk, ce = min(cs + kPpChunk, N);
    // ---- document boundaries q of the chunk's window, cs - kRHalo < q <= hi, in registers (wave uniform)
    const uint64_t hi = min(ce + kRHalo, N);  // window (lo, hi], lo = cs - kRHalo (or 0)
    const uint64_t dnc = P.chunk_doc[2 * c];      // first document that starts at or after cs
    const uint64_t dnA = P.chunk_doc[2 * c + 1];  // first document that starts after lo (>= 1: doc_off[0] = 0)
    uint64_t bq[kRBnd];
    uint32_t nb = 0;
    bool slow;
    {
      const uint64_t dn = dnA + (uint64_t)lane;
      const uint64_t q = (lane <= (int)kRBnd && dn <= D) ? M.doc_off[dn] : ~0ull;
      const unsigned long long inm = __ballot(q <= hi);
      nb = (uint32_t)__popcll(inm & 0xFull);
      slow = (inm >> kRBnd) & 1ull;  // a fifth boundary: many small documents, take the searching path
#pragma unroll
      for (int k = 0; k < (int)kRBnd; k++) {
        const uint64_t qk = __shfl(q, k, 64);
        bq[k] = (uint32_t)k < nb ? qk : ~0ull;
      }
    }
    const uint64_t ds0 = M.doc_off[dnA - 1];  // start of the document that covers lo
    // end (exclusive) of the document of absolute position g, clamped to hi
    auto doc_end = [&](uint64_t g) -> uint64_t {
      if (slow) return min(M.doc_off[first_boundary(M.doc_off, D, g + 1)], hi);
      uint64_t e = hi;
#pragma unroll
      for (int k = (int)kRBnd - 1; k >= 0; k--) e = bq[k] > g && bq[k] < e ? bq[k] : e;
      return e;
    };
    auto doc_start = [&](uint64_t g) -> uint64_t {
      if (slow) return M.doc_off[first_boundary(M.doc_off, D, g + 1) - 1];
      uint64_t s = ds0;
#pragma unroll
      for (int k = 0; k < (int)kRBnd; k++) s = bq[k] <= g ? bq[k] : s;
      return s;
    };
    // tile ends of this chunk's and of the previous chunk's item lists
    const uint64_t te64 = P.tile_end[c];
    const uint64_t pe64 = c ? P.tile_end[c - 1] : 0ull;
    const uint16_t *own = P.items + c * kPpItemCap;
    uint2 *reg = M.evd + c * M.ev_stride;
    uint32_t ev_base = 0;
    uint64_t dcur = dnc;  // next document whose start has not been ranked yet
    bool bad = false;

    for (uint32_t t = 0; t < kPpChunk / kFTile; t++) {
      const uint64_t ts = cs + (uint64_t)t * kFTile;
      if (ts >= N) break;
      const uint64_t te = min(ts + kFTile, N);
      const uint64_t base0 = ts >= kRHalo ? ts - kRHalo : 0;
      const uint32_t off = (uint32_t)(ts - base0), tl = (uint32_t)(te - ts);
      // ---- item bitmap: the tile's own items and the halo part of the previous tile's
      if (lane < (int)kRIWords) S.ibm[lane] = 0;
      if (lane < (int)kREWords) S.ebm[lane] = 0;
      if (lane == 0) S.n_cands = 0;
      wave_sync();
      {
        const uint32_t ib = t ? (uint32_t)(te64 >> (16 * (t - 1))) & 0xFFFFu : 0u;
        const uint32_t ie = (uint32_t)(te64 >> (16 * t)) & 0xFFFFu;
        for (uint32_t i = ib + lane; i < ie; i += 64) {
          const uint32_t p = (own[i] & 0xFFFu) - t * kFTile + off;
          atomicOr(&S.ibm[p >> 5], 1u << (p & 31));
        }
        if (off) {
          const uint16_t *pl = t ? own : P.items + (c - 1) * kPpItemCap;
          const uint64_t e64 = t ? te64 : pe64;
          const uint32_t tp = t ? t - 1 : kPpChunk / kFTile - 1;
          const uint32_t pb = tp ? (uint32_t)(e64 >> (16 * (tp - 1))) & 0xFFFFu : 0u;
          const uint32_t pe = (uint32_t)(e64 >> (16 * tp)) & 0xFFFFu;
          for (uint32_t i = pb + lane; i < pe; i += 64) {
            const uint32_t q = (pl[i] & 0xFFFu) - tp * kFTile;
            if (q >= kFTile - off) {
              const uint32_t p = q - (kFTile - off);
              atomicOr(&S.ibm[p >> 5], 1u << (p & 31));
            }
          }
        }
      }
      wave_sync();
      // ---- ranks, sorted positions
      uint32_t n_items;
      {
        const uint32_t bits0 = lane < (int)kRIWords ? S.ibm[lane] : 0u;
        const uint32_t v = __popc(bits0);
        const uint32_t inc = wave_incl_scan(v);
        n_items = __shfl(inc, 63, 64);
        if (lane < (int)kRIWords) S.wpre[lane] = (uint16_t)(inc - v);
        if (n_items > kRItems) {
          bad = true;
          break;
        }
        uint32_t bits = bits0, r = inc - v;
        while (bits) {
          const uint32_t b = __builtin_ctz(bits);
          bits &= bits - 1;
          S.ipos[r++] = (uint16_t)(lane * 32 + b);
        }
      }
      wave_sync();
      // ---- exact walks of the items (goto probes from the root, cedar.cr:441-447)
      for (uint32_t r = lane; r < n_items; r += 64) {
        const uint32_t p = S.ipos[r];
        const uint64_t g = base0 + p;
        const uint32_t lim = (uint32_t)min<uint64_t>(doc_end(g) - g, Lmax);
        uint64_t w = load8(M.text, g, N);
        uint32_t B = A.root, L = 0;
        for (uint32_t dpt = 1; dpt <= lim; dpt++) {
          if (dpt > 1 && ((dpt - 1) & 7u) == 0) w = load8(M.text, g + dpt - 1, N);
          const uint32_t b = (uint32_t)w & 0xFFu;
          w >>= 8;
          if (b == 0) break;  // keys hold no NUL (cedar.cr:235)
          const uint32_t e = probe(B ^ b);
          if ((e & 0xFFu) != b) break;
          B = (e >> C_BASE_SHIFT) & C_BASE_MASK;
          L = dpt;
          const uint32_t i = p + dpt - 1;  // end position, window relative
          if ((e & C_END) && i >= off && i < off + tl) {
            const uint32_t k = atomicAdd(&S.n_cands, 1u);
            if (k < kRCands) S.cand[k] = make_uint2(B, i | (r << 16));
          }
        }
        S.ireach[r] = (uint8_t)L;
      }
      wave_sync();
      // ---- exclusive prefix maximum of (position + reach) over the sorted items
      {
        constexpr uint32_t per = kRItems / 64;
        const uint32_t r0 = lane * per;
        uint32_t mx = 0;
#pragma unroll
        for (uint32_t k = 0; k < per; k++)
          if (r0 + k < n_items) mx = max(mx, (uint32_t)S.ipos[r0 + k] + S.ireach[r0 + k]);
        uint32_t inc = mx;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
          const uint32_t o = __shfl_up(inc, dd, 64);
          if (lane >= dd) inc = max(inc, o);
        }
        uint32_t run = __shfl_up(inc, 1, 64);
        if (lane == 0) run = 0;
#pragma unroll
        for (uint32_t k = 0; k < per; k++) {
          if (r0 + k < n_items) {
            S.icov[r0 + k] = (uint16_t)run;
            run = max(run, (uint32_t)S.ipos[r0 + k] + S.ireach[r0 + k]);
          }
        }
      }
      wave_sync();
      const uint32_t n_cands = S.n_cands;
      if (n_cands > kRCands) {
        bad = true;
        break;
      }
      // ---- a candidate reports iff no earlier start of its document is still alive at its end
      for (uint32_t k = lane; k < n_cands; k += 64) {
        const uint2 cd = S.cand[k];
        const uint32_t i = cd.y & 0xFFFFu, r = cd.y >> 16;
        const uint32_t p = S.ipos[r];
        bool ok = S.icov[r] <= i;  // items in front of it: exact reaches
        if (ok && p > 0) {
          // boring starts have walks shorter than kPpGuard: only the starts jj >= i - (kPpGuard - 2) can cover i
          const uint64_t dstart = doc_start(base0 + p);
          const uint32_t dlen = i - p;  // jj = p - back may cover i only if i - jj + 1 <= kPpGuard - 1
          const uint32_t nback = dlen >= kPpGuard - 2 ? 0u : min(p, kPpGuard - 2 - dlen);
          const uint64_t w = load8(M.text, base0 + p - nback, N);                   // text[p - nback .. ]
          for (uint32_t back = 1; back <= nback && ok; back++) {
            const uint32_t jj = p - back;
            if (base0 + jj < dstart) break;
            if ((S.ibm[jj >> 5] >> (jj & 31)) & 1u) continue;  // an item: already in the prefix maximum
            const uint32_t len = i - jj + 1;
            uint64_t ww = w >> (8 * (nback - back));
            uint32_t B = A.root;
            bool path = true;
            for (uint32_t dpt = 0; dpt < len; dpt++) {
              const uint32_t b = (uint32_t)ww & 0xFFu;
              ww >>= 8;
              const uint32_t e = probe(B ^ b);  // b == 0 reads the state's own slot: harmless, rejected below
              if (b == 0 || (e & 0xFFu) != b) {
                path = false;
                break;
              }
              B = (e >> C_BASE_SHIFT) & C_BASE_MASK;
            }
            if (path) ok = false;
          }
        }
        if (ok) {
          const uint32_t ir = i - off;
          atomicOr(&S.ebm[ir >> 5], 1u << (ir & 31));
        } else {
          S.cand[k].y = 0xFFFFFFFFu;
        }
      }
      wave_sync();
      // ---- events in position order
      uint32_t n_ev;
      {
        const uint32_t v = lane < (int)kREWords ? __popc(S.ebm[lane]) : 0u;
        const uint32_t inc = wave_incl_scan(v);
        n_ev = __shfl(inc, 63, 64);
        if (lane < (int)kREWords) S.epre[lane] = (uint16_t)(inc - v);
      }
      wave_sync();
      for (uint32_t k = lane; k < n_cands; k += 64) {
        const uint2 cd = S.cand[k];
        if (cd.y == 0xFFFFFFFFu) continue;
        const uint32_t i = cd.y & 0xFFFFu, ir = i - off;
        const uint32_t rank = ev_base + S.epre[ir >> 5] + __popc(S.ebm[ir >> 5] & ((1u << (ir & 31)) - 1u));
        const uint64_t g = base0 + i;
        if (rank < M.ev_stride) reg[rank] = make_uint2(cd.x, (uint32_t)(g + 1 - doc_start(g)));
      }
      // documents that start inside the tile: events of the chunk before the document start
      for (;;) {
        const uint64_t dn = dcur + (uint64_t)lane;
        const uint64_t q = dn <= D ? M.doc_off[dn] : ~0ull;
        const bool in = q < te;
        if (in) {
          const uint32_t qr = (uint32_t)(q - ts);
          M.doc_ev_rank[dn] = ev_base + S.epre[qr >> 5] + __popc(S.ebm[qr >> 5] & ((1u << (qr & 31)) - 1u));
        }
        const uint32_t n = (uint32_t)__popcll(__ballot(in));
        dcur += n;
        if (n < 64) break;
      }
      ev_base += n_ev;
      wave_sync();
    }
    if (lane == 0) {
      M.ev_cnt[c] = ev_base;
      if (bad || ev_base > M.ev_stride) M.cursor[1] = 3ull;  // nested or hit-dense: the host takes the other engine
    }
  }
}

}  // namespace

size_t pp_filter_lds(uint32_t b_words) {
  return (size_t)kPpT2Words * 4 + (size_t)b_words * 4 + (size_t)kFWaves * kFRing * 8 + (size_t)kFWaves * kPpItemCap * 2;
}

size_t pp_resolve_lds(uint32_t lds_slots) { return (size_t)lds_slots * 4 + (size_t)kRWaves * sizeof(WScratch); }
uint32_t pp_resolve_max_slots() { return (uint32_t)((160 * 1024 - kRWaves * sizeof(WScratch)) / 4) & ~3u; }

int pp_prepare(uint32_t b_words, uint32_t lds_slots) {
  int e = (int)hipFuncSetAttribute((const void *)k_pp_filter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)pp_filter_lds(b_words));
  if (e) return e;
  return (int)hipFuncSetAttribute((const void *)k_pp_resolve, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pp_resolve_lds(lds_slots));
}

void pp_launch_filter(const PpArgs &P, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(k_pp_filter, dim3(grid), dim3(kFThreads), pp_filter_lds(P.b_words), (hipStream_t)stream, P);
}

void pp_launch_resolve(const DevAut &A, const V2Args &M, const PpArgs &P, uint32_t grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pp_chunk_doc, dim3((uint32_t)((M.n_chunks + 255) / 256)), dim3(256), 0, s, M.doc_off, M.n_docs,
                     M.n_chunks, P.chunk_doc);
  const uint64_t waves = (M.n_chunks + kRWaves - 1) / kRWaves;
  hipLaunchKernelGGL(k_pp_resolve, dim3((uint32_t)std::min<uint64_t>(grid, std::max<uint64_t>(waves, 1))), dim3(kRThreads),
                     pp_resolve_lds(P.lds_slots), s, A, M, P);
}

}  // namespace aha
