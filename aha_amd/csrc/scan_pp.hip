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
      cur = nxt;
    }
    if (tail != head) batch(tail - head);
    const uint32_t nw = min(ocnt, kPpItemCap);
    uint16_t *dst = P.items + chunk * kPpItemCap;
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = obuf[i];
    if (lane == 0) {
      P.item_cnt[chunk] = nw;
      if (ocnt > kPpItemCap) P.flags[1] = 3ull;  // hit-dense input: the host takes the single-traversal engine
    }
  }
}

// ------------------------------------------------------------------ pass 2
constexpr int kRThreads = 256;
constexpr uint32_t kRHalo = 256;                          // bytes in front of the chunk whose items are re-walked
constexpr uint32_t kRWin = kPpChunk + 2 * kRHalo;         // text window: halo + chunk + look-ahead of the walks
constexpr uint32_t kRItems = kPpItemCap + kRHalo;         // positions are unique: at most kRHalo items in the halo
constexpr uint32_t kRWords = (kPpChunk + kRHalo) / 32;    // bitmap words over halo + chunk
constexpr uint32_t kRCands = 1024;                        // END nodes that end inside the chunk
constexpr uint32_t kRBnd = 64;                            // document boundaries cached per window

struct RShared {
  alignas(16) uint8_t txt[kRWin + 16];
  uint32_t ibm[kRWords];      // item bitmap over [base0, ce)
  uint32_t ebm[kPpChunk / 32];  // event bitmap over [cs, ce)
  uint16_t wpre[kRWords];     // exclusive prefix of popcounts of ibm
  uint16_t epre[kPpChunk / 32];
  uint16_t ipos[kRItems];     // item positions (relative to base0), ascending
  uint16_t icov[kRItems];     // max over earlier items of (position + reach), relative to base0
  uint8_t ireach[kRItems];
  uint2 cand[kRCands];        // x = state base, y = i (relative to base0) | item rank << 16
  uint64_t bnd[kRBnd];        // document boundaries in (base0, tend]
  uint32_t scan[kRThreads / 64];
  uint32_t n_items, n_cands, n_bnd, bnd_slow, abort;
};

__global__ __launch_bounds__(kRThreads) void k_pp_resolve(DevAut A, V2Args M, PpArgs P) {
  __shared__ RShared S;
  const uint32_t *slots = reinterpret_cast<const uint32_t *>(A.slots);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t N = M.n_bytes, D = M.n_docs;
  const uint32_t Lmax = A.max_len;
  // pass 1 overflowed: results are discarded by the host.  One lane reads the flag for the whole workgroup: other
  // workgroups of THIS launch may set it meanwhile, and a barrier must never see only part of a workgroup.
  if (tid == 0) S.abort = M.cursor[1] != 0;
  __syncthreads();
  if (S.abort) return;

  for (uint64_t c = blockIdx.x; c < M.n_chunks; c += gridDim.x) {
    const uint64_t cs = c * kPpChunk, ce = min(cs + kPpChunk, N);
    const uint64_t base0 = cs >= kRHalo ? cs - kRHalo : 0;
    const uint64_t tend = min(ce + kRHalo, N);
    const uint32_t off = (uint32_t)(cs - base0);   // chunk start inside the window
    const uint32_t wl = (uint32_t)(tend - base0);  // window bytes
    // ---- window text, bitmaps, counters
    for (uint32_t i = tid * 16; i < wl + 16; i += kRThreads * 16) {
      const uint64_t g = base0 + i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (g + 16 <= N && (reinterpret_cast<uintptr_t>(M.text + g) & 15) == 0) {
        v = *reinterpret_cast<const uint4 *>(M.text + g);
      } else if (g < N) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16 && g + j < N; j++) w[j >> 2] |= (uint32_t)M.text[g + j] << ((j & 3) * 8);
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      *reinterpret_cast<uint4 *>(S.txt + i) = v;
    }
    for (uint32_t i = tid; i < kRWords; i += kRThreads) S.ibm[i] = 0;
    for (uint32_t i = tid; i < kPpChunk / 32; i += kRThreads) S.ebm[i] = 0;
    if (tid == 0) {
      S.n_cands = 0;
      S.bnd_slow = 0;
    }
    // document boundaries q with base0 < q <= tend; ds0 = start of the document that covers base0
    const uint64_t dn0 = first_boundary(M.doc_off, D, base0 + 1);  // >= 1 because doc_off[0] = 0
    const uint64_t ds0 = M.doc_off[dn0 - 1];
    if (tid < (int)kRBnd) {
      const uint64_t dn = dn0 + tid;
      S.bnd[tid] = (dn <= D && M.doc_off[dn] <= tend) ? M.doc_off[dn] : ~0ull;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t n = 0;
      while (n < kRBnd && S.bnd[n] != ~0ull) n++;
      S.n_bnd = n;
      if (n == kRBnd && dn0 + kRBnd <= D && M.doc_off[dn0 + kRBnd] <= tend) S.bnd_slow = 1;  // many tiny documents
    }
    // ---- item bitmap: own chunk, and the halo part of the previous chunk's list
    {
      const uint32_t n_own = P.item_cnt[c];
      const uint16_t *own = P.items + c * kPpItemCap;
      for (uint32_t i = tid; i < n_own; i += kRThreads) {
        const uint32_t p = (own[i] & 0xFFFu) + off;
        atomicOr(&S.ibm[p >> 5], 1u << (p & 31));
      }
      if (off) {
        const uint32_t n_prev = P.item_cnt[c - 1];
        const uint16_t *prev = P.items + (c - 1) * kPpItemCap;
        for (uint32_t i = tid; i < n_prev; i += kRThreads) {
          const uint32_t q = prev[i] & 0xFFFu;
          if (q >= kPpChunk - off) {
            const uint32_t p = q - (kPpChunk - off);
            atomicOr(&S.ibm[p >> 5], 1u << (p & 31));
          }
        }
      }
    }
    __syncthreads();
    // ---- word prefix, sorted positions
    {
      uint32_t v = 0;
      if (tid < (int)kRWords) v = __popc(S.ibm[tid]);
      uint32_t inc = wave_incl_scan(v);
      if (lane == 63) S.scan[wave] = inc;
      __syncthreads();
      uint32_t basew = 0;
      for (int w = 0; w < wave; w++) basew += S.scan[w];
      if (tid < (int)kRWords) S.wpre[tid] = (uint16_t)(basew + inc - v);
      if (tid == kRThreads - 1) S.n_items = basew + inc;
      __syncthreads();
      if (tid < (int)kRWords) {
        uint32_t bits = S.ibm[tid], r = S.wpre[tid];
        while (bits) {
          const uint32_t b = __builtin_ctz(bits);
          bits &= bits - 1;
          S.ipos[r++] = (uint16_t)(tid * 32 + b);
        }
      }
      __syncthreads();
    }
    const uint32_t n_items = S.n_items, n_bnd = S.n_bnd;
    const bool slow = S.bnd_slow != 0;
    // end of the document of window position p (exclusive, window relative, clamped to the window)
    auto doc_end = [&](uint32_t p) -> uint32_t {
      const uint64_t g = base0 + p;
      if (slow) {
        const uint64_t dn = first_boundary(M.doc_off, D, g + 1);
        return (uint32_t)(min(M.doc_off[dn], tend) - base0);
      }
      for (uint32_t k = 0; k < n_bnd; k++)
        if (S.bnd[k] > g) return (uint32_t)(S.bnd[k] - base0);
      return wl;
    };
    // start of the document of absolute position g (g >= base0)
    auto doc_start = [&](uint64_t g) -> uint64_t {
      if (slow) {
        const uint64_t dn = first_boundary(M.doc_off, D, g + 1);
        return M.doc_off[dn - 1];
      }
      uint64_t s = ds0;
      for (uint32_t k = 0; k < n_bnd; k++)
        if (S.bnd[k] <= g) s = S.bnd[k];
      return s;
    };
    // ---- exact walks of the items (goto probes from the root, cedar.cr:441-447)
    for (uint32_t r = tid; r < n_items; r += kRThreads) {
      const uint32_t p = S.ipos[r];
      const uint32_t lim = min(doc_end(p) - p, Lmax);
      uint32_t B = A.root, L = 0;
      for (uint32_t dpt = 1; dpt <= lim; dpt++) {
        const uint32_t b = S.txt[p + dpt - 1];
        if (b == 0) break;  // keys hold no NUL (cedar.cr:235)
        const uint32_t e = slots[B ^ b];
        if ((e & 0xFFu) != b) break;
        B = (e >> C_BASE_SHIFT) & C_BASE_MASK;
        L = dpt;
        const uint32_t i = p + dpt - 1;  // end position, window relative
        if ((e & C_END) && i >= off && i < off + (uint32_t)(ce - cs)) {
          const uint32_t k = atomicAdd(&S.n_cands, 1u);
          if (k < kRCands) S.cand[k] = make_uint2(B, i | (r << 16));
        }
      }
      S.ireach[r] = (uint8_t)L;
    }
    __syncthreads();
    // ---- exclusive prefix maximum of (position + reach) over the sorted items
    {
      constexpr uint32_t per = (kRItems + kRThreads - 1) / kRThreads;
      const uint32_t r0 = tid * per;
      uint32_t mx = 0;
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        if (r < n_items) mx = max(mx, (uint32_t)S.ipos[r] + S.ireach[r]);
      }
      uint32_t inc = mx;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) {
        const uint32_t o = __shfl_up(inc, dd, 64);
        if (lane >= dd) inc = max(inc, o);
      }
      if (lane == 63) S.scan[wave] = inc;
      __syncthreads();
      uint32_t run = 0;
      for (int w = 0; w < wave; w++) run = max(run, S.scan[w]);
      const uint32_t prev = __shfl_up(inc, 1, 64);
      if (lane > 0) run = max(run, prev);
      for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        if (r < n_items) {
          S.icov[r] = (uint16_t)run;
          run = max(run, (uint32_t)S.ipos[r] + S.ireach[r]);
        }
      }
      __syncthreads();
    }
    const uint32_t n_cands = S.n_cands;
    if (n_cands > kRCands) {
      if (tid == 0) M.cursor[1] = 3ull;  // deeply nested keys: the host takes the single-traversal engine
      return;
    }
    // ---- a candidate reports iff no earlier start of its document is still alive at its end
    for (uint32_t k = tid; k < n_cands; k += kRThreads) {
      const uint2 cd = S.cand[k];
      const uint32_t i = cd.y & 0xFFFFu, r = cd.y >> 16;
      const uint32_t p = S.ipos[r];
      bool ok = S.icov[r] <= i;  // items in front of it: exact reaches
      if (ok) {
        // boring starts in front of it have walks shorter than kPpGuard: only jj >= i - (kPpGuard - 2) can cover i
        const uint64_t dstart = doc_start(base0 + p);
        for (uint32_t back = 1; back < kPpGuard - 1 && ok; back++) {
          if (p < back) break;
          const uint32_t jj = p - back;
          const uint32_t len = i - jj + 1;
          if (len > kPpGuard - 1) break;
          if (base0 + jj < dstart) break;
          if ((S.ibm[jj >> 5] >> (jj & 31)) & 1u) continue;  // an item: already in the prefix maximum
          uint32_t B = A.root;
          bool path = true;
          for (uint32_t dpt = 0; dpt < len; dpt++) {
            const uint32_t b = S.txt[jj + dpt];
            const uint32_t e = b ? slots[B ^ b] : 0u;
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
    __syncthreads();
    // ---- events in position order
    {
      uint32_t v = 0;
      if (tid < (int)(kPpChunk / 32)) v = __popc(S.ebm[tid]);
      uint32_t inc = wave_incl_scan(v);
      if (lane == 63) S.scan[wave] = inc;
      __syncthreads();
      uint32_t basew = 0;
      for (int w = 0; w < wave; w++) basew += S.scan[w];
      if (tid < (int)(kPpChunk / 32)) S.epre[tid] = (uint16_t)(basew + inc - v);
      uint32_t n_ev = 0;
      for (int w = 0; w < kRThreads / 64; w++) n_ev += S.scan[w];
      __syncthreads();
      uint2 *reg = M.evd + c * M.ev_stride;
      for (uint32_t k = tid; k < n_cands; k += kRThreads) {
        const uint2 cd = S.cand[k];
        if (cd.y == 0xFFFFFFFFu) continue;
        const uint32_t i = cd.y & 0xFFFFu, ir = i - off;
        const uint32_t rank = S.epre[ir >> 5] + __popc(S.ebm[ir >> 5] & ((1u << (ir & 31)) - 1u));
        const uint64_t g = base0 + i;
        if (rank < M.ev_stride) reg[rank] = make_uint2(cd.x, (uint32_t)(g + 1 - doc_start(g)));
      }
      if (tid == 0) {
        M.ev_cnt[c] = n_ev;
        if (n_ev > M.ev_stride) M.cursor[1] = 3ull;
      }
      // documents that start inside the chunk: events of the chunk before the document start
      const uint64_t dfirst = first_boundary(M.doc_off, D, cs);
      for (uint64_t dn = dfirst + tid; dn <= D; dn += kRThreads) {
        const uint64_t q = M.doc_off[dn];
        if (q >= ce) break;
        const uint32_t qr = (uint32_t)(q - cs);
        M.doc_ev_rank[dn] = S.epre[qr >> 5] + __popc(S.ebm[qr >> 5] & ((1u << (qr & 31)) - 1u));
      }
    }
    __syncthreads();
  }
}

}  // namespace

size_t pp_filter_lds(uint32_t b_words) {
  return (size_t)kPpT2Words * 4 + (size_t)b_words * 4 + (size_t)kFWaves * kFRing * 8 + (size_t)kFWaves * kPpItemCap * 2;
}

int pp_prepare(uint32_t b_words) {
  return (int)hipFuncSetAttribute((const void *)k_pp_filter, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)pp_filter_lds(b_words));
}

void pp_launch_filter(const PpArgs &P, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(k_pp_filter, dim3(grid), dim3(kFThreads), pp_filter_lds(P.b_words), (hipStream_t)stream, P);
}

void pp_launch_resolve(const DevAut &A, const V2Args &M, const PpArgs &P, void *stream) {
  const uint32_t grid = (uint32_t)std::min<uint64_t>(M.n_chunks, 1u << 20);
  hipLaunchKernelGGL(k_pp_resolve, dim3(grid), dim3(kRThreads), 0, (hipStream_t)stream, A, M, P);
}

}  // namespace aha
