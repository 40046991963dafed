// scan_hash.hip -- the character-level traversal over the HASH image (hash.hpp) for gfx950: the walk of ku_traverse
// (scan_unit.hip: persistent workgroups, a lane per chunk, wave-private input windows, wave-buffered events, the same
// per-chunk outputs -- every post pass is shared) with a trip that never turns bytes into symbols.  A character is its raw
// UTF-8 bytes; "is the previous character followed by this one on some key's first two characters" is a blocked Bloom
// filter in LDS; the transitions are entries of two hash tables keyed by (parent, character).  Replaces
// src/aha/ac.cr:176-192 (match_) exactly like ku_traverse does (unit.hpp has the exactness argument: same automaton,
// same state identities).
//
// LDS: the pair filter (64 KiB), the displacement bytes of the pairs' perfect hash (<= 16 KiB), a 256-byte table of unit
// lengths, the waves' input windows and event buffers.  No root table, no decode tables.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "hash.hpp"
#include "image.hpp"
#include "unit.hpp"

namespace aha {

namespace {

constexpr int kHPiece = 16;
constexpr int kHWin = 3 * kHPiece;
constexpr int kHRow = kHWin + 4;  // bytes per lane in LDS (13 dwords: odd stride)
constexpr int kHWave = 64 * kHRow;
constexpr int BB = 22;

__device__ __forceinline__ uint4 hload16(const uint8_t *text, int64_t g, int64_t N) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (g >= 0 && g + 16 <= N) {
    v = *reinterpret_cast<const uint4 *>(text + g);
  } else if (g >= 0 && g < N) {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // (rare: the text's last bytes)
#pragma unroll 1
    for (int j = 0; j < 16 && g + j < N; j++) {
      const uint32_t b = (uint32_t)text[g + j] << ((j & 3) * 8);
      w0 |= (j >> 2) == 0 ? b : 0u;
      w1 |= (j >> 2) == 1 ? b : 0u;
      w2 |= (j >> 2) == 2 ? b : 0u;
      w3 |= (j >> 2) == 3 ? b : 0u;
    }
    v = make_uint4(w0, w1, w2, w3);
  }
  return v;
}

__device__ __forceinline__ uint64_t hballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool hany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool hall(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

__host__ __device__ inline size_t h_lds_tabs(uint32_t n_groups) {  // filter, displacement bytes, unit lengths
  return ((size_t)4 << kHBloomLog2) + (((size_t)n_groups + 15) & ~(size_t)15) + 256;
}
__host__ __device__ inline size_t h_lds(uint32_t n_groups) {
  return h_lds_tabs(n_groups) + (size_t)(kV2Threads / 64) * kHWave + (size_t)(kV2Threads / 64) * 64 * 12 + 16;
}

template <bool CHARS>
__global__ __launch_bounds__(kV2Threads) void kh_traverse(HashDev H, V2Args M) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;  // the doc offsets are not what the call says (k_check_docs ran in front): index nothing
  uint32_t *bl = reinterpret_cast<uint32_t *>(smem);
  uint8_t *dispb = smem + ((size_t)4 << kHBloomLog2);
  const uint32_t n_groups = H.n_groups;
  uint8_t *lentab = dispb + ((n_groups + 15u) & ~15u);
  for (uint32_t i = threadIdx.x; i < (1u << kHBloomLog2); i += kV2Threads) bl[i] = H.bloom[i];
  for (uint32_t i = threadIdx.x; i < n_groups; i += kV2Threads) dispb[i] = H.disp[i];
  // 8 * bytes the unit a byte starts would have: one byte unless it is a lead byte of a two- or three-byte unit (unit.hpp)
  if (threadIdx.x < 256) lentab[threadIdx.x] = (threadIdx.x & 0xE0u) == 0xC0u ? 16 : ((threadIdx.x & 0xF0u) == 0xE0u ? 24 : 8);
  __syncthreads();
  uint8_t *in_base = smem + h_lds_tabs(n_groups);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *inl = in_base + wave * kHWave + lane * kHRow;
  const uint32_t *row = reinterpret_cast<const uint32_t *>(inl);
  typedef uint32_t v3u __attribute__((ext_vector_type(3)));
  const uint32_t wbo = (uint32_t)(h_lds_tabs(n_groups) + (size_t)(kV2Threads / 64) * kHWave) + (uint32_t)wave * (64u * 12u);
  const uint4 *pairs = H.pairs, *deept = H.deep;
  const uint32_t k1 = H.k1, gmask = n_groups - 1u, psh = 32u - H.pair_log2, pmask = (1u << H.pair_log2) - 1u, dsh = 32u - H.deep_log2;
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int64_t S = (int64_t)M.S;
  const int rounds = (int)(M.S / kHPiece);
  const int warm = H.max_len > 1 ? (int)H.max_len - 1 : 0;
  const int R = (warm + kHPiece - 1) / kHPiece;  // warm-up rounds before the chunk

  const uint64_t n_tiles = (M.n_chunks + kV2Threads - 1) / kV2Threads;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t chunk = tile * kV2Threads + threadIdx.x;
    const uint32_t ev_stride = M.ev_stride;
    const bool live = chunk < M.n_chunks;
    const int64_t a = (int64_t)chunk * S;
    const int64_t e = live ? min(a + S, N) : a;
    uint64_t dn = 0;
    int64_t nb = INT64_MAX, doc_start = a, pos = e;
    // events: a buffer of 64 records per wave in LDS, stored 768 bytes at a time into the wave's part of the event space
    // (scan_unit.hip has the why; same records: unit.hpp u_rec_*)
    uint32_t wfill = 0, wout = 0;
    const uint64_t wchunk0 = tile * kV2Threads + (uint64_t)wave * 64;
    uint32_t *wreg = M.evg + wchunk0 * M.ev_stride * 3;
    const uint32_t wcap = (uint32_t)min<uint64_t>(64, M.n_chunks > wchunk0 ? M.n_chunks - wchunk0 : 0) * M.ev_stride;
    uint32_t hits = 0;
    uint32_t lead_total = 0;
    uint32_t lc = 0, lc_exact = 0;
    uint32_t seq = 0;
    // the state: E = the unit image's word of a state of two characters or more (hash.hpp), 0 = the root or a one-character
    // state -- that of the character consumed last, pc; CF = E's child filter; hp / hpc = the hash parts of E / of pc
    uint32_t E = 0, CF = 0, hp = 0, pc = 0, hpc = 0;
    uint4 q1 = make_uint4(0, 0, 0, 0), q2 = q1, q3 = q1;
    uint32_t w[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      dn = first_boundary(M.doc_off, D, (uint64_t)a);
      nb = (int64_t)M.doc_off[dn];
      pos = a;
      if (nb != a) {
        doc_start = (int64_t)M.doc_off[dn - 1];
        pos = a - min<int64_t>(a - doc_start, warm);
        if (CHARS) M.chunk_doc0[chunk] = (uint32_t)(dn - 1);
      } else if (CHARS) {
        M.chunk_doc0[chunk] = (uint32_t)dn;
      }
    }

    for (int r = -((R + 3) / 4) * 4 - 2; r <= rounds; r++) {
      const int64_t pb = a + (int64_t)r * kHPiece;
      const int64_t pl = pb + 2 * kHPiece;  // the piece loaded in this round
      const int64_t pend = min(pb + kHWin, e);
      {
        uint4 v;
        if ((pl & 63) == 0) {
          v = hload16(M.text, pl, N);
          q1 = hload16(M.text, pl + 16, N);
          q2 = hload16(M.text, pl + 32, N);
          q3 = hload16(M.text, pl + 48, N);
        } else {
          v = q1;
          q1 = q2;
          q2 = q3;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) w[i] = w[i + 4];
        w[9] = v.x;
        w[10] = v.y;
        w[11] = v.z;
        w[12] = v.w;
        uint32_t *dst = reinterpret_cast<uint32_t *>(inl);
#pragma unroll
        for (int i = 0; i < 13; i++) dst[i] = w[i];
      }
      const bool need = live && pos < pend;
      if (!hany(need)) continue;
      uint32_t rel = kHRow, lim = 0;
      if (need) {
        rel = (uint32_t)(pos - pb + 4);
        lim = (uint32_t)(pend - pb + 4);  // units that START before pend
      }
      uint32_t nb_rel = (nb >= pb - 4 && nb <= pb + kHWin) ? (uint32_t)(nb - pb + 4) : ~0u;
      const uint32_t n_rel = (N - pb) <= (int64_t)kHWin ? (uint32_t)max<int64_t>(N - pb + 4, 0) : ~0u;
      const int32_t a_rel = (int32_t)max<int64_t>(a - pb + 4, -128);
      const int32_t e_rel = (int32_t)min<int64_t>(e - pb + 4, 128);
      int32_t docrel = (int32_t)(pb - 4 - doc_start);

      for (;;) {  // outer: resolve document boundaries, then run the trips up to the next one
        const bool bnd = rel < lim && rel == nb_rel;
        if (hany(bnd)) {  // rare: a document starts here (ac.cr:177: the state is per sequence)
          if (bnd) {
            const int64_t here = pb - 4 + rel;
            do {
              M.doc_ev_rank[dn] = seq;
              M.doc_hit_rank[dn] = hits;
              if (CHARS) M.doc_lead_rank[dn] = lead_total;
              dn++;
              nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            } while (nb == here);
            asm volatile("" : "+v"(nb));
            nb_rel = (nb <= pb + kHWin) ? (uint32_t)(nb - pb + 4) : ~0u;
            E = 0;
            pc = 0;
            hpc = 0;
            lc = 0;
            lc_exact = 1;
            doc_start = here;
            docrel = -(int32_t)rel;
          }
        }
        uint32_t lim2 = min(lim, nb_rel);          // lanes park at the next boundary
        const uint32_t dend = min(nb_rel, n_rel);  // first byte that is not this document's
        // The unit at row position `at` (unit.hpp: what a position offers as one symbol), as a raw character: its first byte
        // announces its length; a lead byte whose continuation bytes are missing, malformed or beyond the document is a
        // one-byte unit of its own -- its byte value is no character of any key, so it matches nothing anywhere.
        auto decode = [&](uint32_t at, uint32_t &o_c, uint32_t &o_L, uint32_t &o_g, bool &o_later) {
          const uint32_t lo = row[at >> 2], hi = row[(at >> 2) + 1];
          const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, at & 3u);
          const uint32_t b0 = w4 & 0xFFu;
          const uint32_t s = lentab[b0];
          const uint32_t want = s >> 3;
          const bool in_doc = at + want <= dend;
          o_later = in_doc & at + want > (uint32_t)kHRow;  // its bytes are not all staged yet: next round
          const uint32_t cm = __builtin_amdgcn_ubfe(0xC0C000u, 0u, s);  // the continuation bytes' top bits must read 10
          const bool whole = in_doc & ((w4 ^ 0x808000u) & cm) == 0u;
          const uint32_t se = whole ? s : 8u;
          o_c = __builtin_amdgcn_ubfe(w4, 0u, se);
          o_L = se >> 3;
          if (CHARS) o_L |= (b0 & 0xC0u) != 0x80u ? 0x100u : 0u;
          o_g = h_mul24(o_c, kHK2);
        };
        uint32_t c, L, g;
        {
          bool later;
          decode(min(rel, (uint32_t)kHRow), c, L, g, later);
          lim2 = later ? min(lim2, rel) : lim2;
          lim = later ? min(lim, rel) : lim;
        }
        bool all_left = false;
        for (;;) {
          const bool act = rel < lim2;
          all_left = hall(rel >= (uint32_t)(4 + kHPiece) || rel >= lim);
          if (all_left || !hany(act)) break;
          uint32_t evc = 0;
          {
            const uint32_t Bq = E & 0x3FFFFFu;
            const bool deep = Bq != 0u;
            const bool hdr = u_hdr_pending(E), nfr = u_nfr(E), f1 = u_f1(E);
            // the state's child filter answers most "does it continue on this character" without a probe
            const bool dpass = deep & !hdr & ((CF >> (g >> 27)) & 1u) != 0u;
            const bool fall = deep & !hdr & !dpass;  // a miss known without a probe
            // A miss falls to the root (no NFR: the character is consumed there), to the one-character state of the
            // character that led here (F1: the pair (pc, c) is asked in this very trip when the miss needed no probe), or to
            // the state's fail state, whose word its header holds.
            const bool pairmode = !deep | (fall & nfr & f1);
            const bool needhdr = hdr | (fall & nfr & !f1);
            const uint32_t ck = needhdr ? kHHdr : c;
            const uint32_t h = h_key(pairmode ? hpc : hp, ck, k1);
            const uint32_t bw = bl[h_bloom_word(h)];
            const uint32_t dd = dispb[(h >> 7) & gmask];
            const uint32_t bm = h_bloom_mask(h);
            const bool probe = act & ((pairmode & (bw & bm) == bm) | dpass | needhdr);
            const uint32_t t = h * kHMix;
            const uint32_t ps = ((t >> psh) + dd * ((t << 1) | 1u)) & pmask;
            const uint4 *a1 = pairmode ? pairs + ps : deept + (t >> dsh);
            const uint4 *a2 = pairmode ? a1 : deept + ((t * kHMix2) >> dsh);
            a1 = probe ? a1 : pairs;
            a2 = probe ? a2 : pairs;
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            v4u e1, e2;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e1) : "v"(a1) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e2) : "v"(a2) : "memory");
            if (act) {
              uint32_t n_c, n_L, n_g;
              bool n_later;
              const uint32_t Lb = CHARS ? (L & 0xFFu) : L;
              decode(rel + Lb, n_c, n_L, n_g, n_later);  // rows are padded: rel + 3 + 8 bytes stay inside LDS
              asm volatile("s_waitcnt vmcnt(0)" : "+v"(e1), "+v"(e2) : : "memory");
              const uint32_t Pk = pairmode ? (kHTag | pc) : Bq;
              const bool hit1 = e1.x == Pk & (e1.y & 0xFFFFFFu) == ck;
              const bool hit2 = e2.x == Pk & (e2.y & 0xFFFFFFu) == ck;
              const bool hit = probe & (hit1 | hit2);
              const uint32_t ey = hit1 ? e1.y : e2.y, ez = hit1 ? e1.z : e2.z, ew = hit1 ? e1.w : e2.w;
              const bool child = hit & !needhdr;  // a transition: the character is consumed
              // (a header that is not there cannot happen; consuming the character then keeps the walk finite)
              const bool consumed = child | (!hit & (pairmode | !nfr | needhdr));
              uint32_t En = (!hit & !consumed & !f1) ? (Bq | 0x20000000u) : 0u;  // fetch the header next / fall to pc's state
              En = hit ? ez : En;
              E = En;
              CF = hit ? ew : CF;
              hp = h_part_base(En & 0x3FFFFFu);
              const bool end = child & u_end(En);
              hpc = consumed ? h_rot(g) : hpc;
              pc = consumed ? c : pc;
              const uint32_t adv = consumed ? Lb : 0u;
              if (CHARS) {
                const uint32_t isl = (consumed & (int32_t)rel >= a_rel) ? (L >> 8) : 0u;
                lc += isl;
                lead_total += isl;
              }
              rel += adv;
              const bool park = consumed & n_later;
              lim2 = park ? rel : lim2;
              lim = park ? rel : lim;
              c = consumed ? n_c : c;
              g = consumed ? n_g : g;
              L = consumed ? n_L : L;
              const int32_t last = (int32_t)rel - 1;
              // is_end? -> fetch later (ac.cr:183-185); this lane reports the end positions in [a, e)
              evc = (end & last >= a_rel & last < e_rel) ? (ey >> 24) : 0u;
            }
          }
          const uint64_t evm = hballot(evc != 0u);
          if (evm) {
            const bool ev = evc != 0u;
            const uint32_t my = wfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(evm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)evm, 0u));
            const uint32_t rx = u_rec_x(u_child(E, BB), (uint32_t)lane, evc, BB), rz = u_rec_z(hits, evc, BB);
            const uint32_t ry = CHARS ? ((lc << 1) | lc_exact) : (uint32_t)(docrel + (int32_t)rel);
            if (ev && my < 64u) {
              uint32_t *d = reinterpret_cast<uint32_t *>(smem + (wbo + __umul24(my, 12u)));
              d[0] = rx;
              d[1] = ry;
              d[2] = rz;
            }
            const uint32_t kp = __popcll(evm);
            if (wfill + kp >= 64u) {
              const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + (wbo + __umul24((uint32_t)lane, 12u)));
              const v3u rr = {q[0], q[1], q[2]};
              if (wout + 64u <= wcap) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = rr;
              wout += 64u;
              if (ev && my >= 64u) {
                uint32_t *d = reinterpret_cast<uint32_t *>(smem + (wbo + __umul24(my - 64u, 12u)));
                d[0] = rx;
                d[1] = ry;
                d[2] = rz;
              }
            }
            wfill = (wfill + kp) & 63u;
            seq += ev ? 1u : 0u;
            hits += evc;
          }
        }
        if (all_left || !hany(rel < lim)) break;
      }
      if (need) pos = pb - 4 + rel;
    }
    {  // the rest of the wave's buffer
      const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + (wbo + __umul24((uint32_t)lane, 12u)));
      const v3u rr = {q[0], q[1], q[2]};
      if ((uint32_t)lane < wfill && wout + (uint32_t)lane < wcap) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = rr;
    }
    if (live) {
      M.ev_cnt[chunk] = seq;
      if (seq > ev_stride) M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
      if (CHARS) M.lead_cnt[chunk] = lead_total;
      M.chunk_hits[chunk] = hits;
      if (e == N) {  // documents that start at N (empty tail documents, and d = D)
        while (dn <= D) {
          M.doc_ev_rank[dn] = seq;
          M.doc_hit_rank[dn] = hits;
          if (CHARS) M.doc_lead_rank[dn] = lead_total;
          dn++;
        }
      }
    }
  }
}

}  // namespace

size_t hash_lds_bytes(uint32_t n_groups) { return h_lds(n_groups); }

int hash_prepare(uint32_t n_groups) {
  const int lds = (int)h_lds(n_groups);
  int e = (int)hipFuncSetAttribute((const void *)kh_traverse<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (!e) e = (int)hipFuncSetAttribute((const void *)kh_traverse<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  return e;
}

void hash_launch_traverse(const HashDev &H, const V2Args &M, uint32_t grid, void *stream) {
  const size_t lds = h_lds(H.n_groups);
  if (M.chars)
    hipLaunchKernelGGL(kh_traverse<true>, dim3(grid), dim3(kV2Threads), lds, (hipStream_t)stream, H, M);
  else
    hipLaunchKernelGGL(kh_traverse<false>, dim3(grid), dim3(kV2Threads), lds, (hipStream_t)stream, H, M);
}

}  // namespace aha
