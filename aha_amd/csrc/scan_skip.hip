// scan_skip.hip -- the skip-ahead traversal for gfx950 (engine 6): ku_traverse's walk over the unit image (unit.hpp), started
// only where a two-unit trie path can start.
//
// Why (profiles/r05_hash_walk_proto.txt): ku_traverse is bound by instruction issue -- ~175
// instructions per unit and wave, 0.52 trips per byte on cfg 3 -- while a STATELESS look at every unit (raw bytes, one hash,
// one Bloom-filter word in LDS) costs a fifth of that.  And the walk's state is shallow most of the time: for 86 % of cfg 3's
// characters it is the root or a one-unit state.  In such a state the next state depends on the next two units alone -- the
// two-unit state when they spell a trie path, else the one-unit state (or the root) of the second -- and, when no key is a
// single unit, nothing is reported on the way (src/aha/ac.cr:183-185 reports END states only).  So:
//
//   ks_mark      every unit start p whose two units pass the filter over the image's two-unit paths gets a bit (unit.hpp,
//                MARKS): a superset of the positions where a two-unit path starts.  Coalesced, stateless, a lane per 64-byte
//                piece; the chain of units is followed by validated lengths, so malformed text is segmented like unit.hpp says.
//   ks_traverse  a lane per chunk like ku_traverse, the same state word, probes, fail links, event records and per-chunk
//                outputs -- but a lane whose state is shallow JUMPS to the next mark at or behind its position, takes the unit
//                there from the root's table and goes on with ordinary trips until the state is shallow again.  Lanes run
//                free: no rounds, no LDS input window -- a lane reads the 16 text bytes at its jump target and its piece of the
//                bitmap straight from memory, prefetched a trip ahead.
//
// Exactness: the automaton started at the root at any position x visits, from x + Lmax - 1 on, the states of
// src/aha/ac.cr:176-192 (the chunks' warm-up).  Skipping from a shallow state at q (the start of the last
// consumed unit) to the first mark m >= q changes nothing: no two-unit path starts in [q, m), so every state in between is
// the root or a one-unit state -- which reports nothing and is forgotten when the unit at m is consumed, because the pair
// (unit before m, unit at m) is not a path either.  tests/skipsim.py is the CPU twin of both kernels.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"
#include "unit.hpp"

namespace aha {

namespace {

constexpr int kSkPiece = 64;             // bytes of text per lane of ks_mark
constexpr int kSkRow = 80;               // its LDS row: the piece + 16 bytes of look-ahead (two units behind a unit that starts at byte 63)
constexpr int kSkPad = 4;                // bitmap words behind the text's last one (ks_traverse reads two ahead)
constexpr uint32_t kSkLimit = 124;       // a lane looks for marks in [.., 64 bw + 124): the jump's unit ends inside its two words

__host__ __device__ inline size_t sk_mark_lds(uint32_t log2w) { return ((size_t)4 << log2w) + 1024 + (size_t)kV2Threads * kSkRow; }
__host__ __device__ inline size_t sk_walk_lds(uint32_t n_syms) {  // decode tables, root table, the waves' event buffers
  return (size_t)(kUTabWords + ((n_syms + 3u) & ~3u)) * 4 + 16 + (size_t)(kV2Threads / 64) * 64 * 12;
}

typedef uint32_t sk_v4u __attribute__((ext_vector_type(4)));

// 16 text bytes from any byte address (gfx950 global loads take unaligned addresses); bytes beyond the text read as 0
__device__ __forceinline__ sk_v4u sk_text16(const uint8_t *__restrict__ text, int64_t g, int64_t N) {
  sk_v4u v = {0u, 0u, 0u, 0u};
  if (g + 16 <= N) {
    __builtin_memcpy(&v, text + g, 16);
  } else if (g < N) {  // (the last bytes of the batch)
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll 1
    for (int j = 0; j < 16 && g + j < N; j++) {
      const uint32_t b = (uint32_t)text[g + j] << ((j & 3) * 8);
      const int k = j >> 2;
      w0 |= k == 0 ? b : 0u;
      w1 |= k == 1 ? b : 0u;
      w2 |= k == 2 ? b : 0u;
      w3 |= k == 3 ? b : 0u;
    }
    v = sk_v4u{w0, w1, w2, w3};
  }
  return v;
}

__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// ---- the marks.  A wave takes 64 pieces of 64 bytes (a tile of 4 KiB, read coalesced); a lane walks its piece unit by unit:
// the unit's length from its first byte, checked against its continuation bytes (a lead byte without them is a one-byte unit:
// unit.hpp), its bytes as an integer, the pair hash with the unit before (whose share is carried), one word of the filter.
// A lane marks the pairs whose FIRST unit starts in its piece, reading up to 16 bytes into the next one; the first bytes of a
// piece may belong to a unit that started in the piece before: they walk as one-byte units that match nothing until the
// first byte outside 0x80..0xBF -- a unit start whatever precedes it -- puts the chain in step.
__global__ __launch_bounds__(kV2Threads) void ks_mark(SkipDev K, V2Args M, unsigned long long *__restrict__ bitmap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;  // the doc offsets are not what the call says: the call fails, nothing reads the bitmap
  const uint32_t words = 1u << K.log2;
  uint32_t *bl = reinterpret_cast<uint32_t *>(smem);
  uint32_t *lt = bl + words;  // by first byte: continuation-byte mask of the unit (bits 8..23) | its bits (8, 16, 24)
  uint8_t *rows = reinterpret_cast<uint8_t *>(lt + 256);
  for (uint32_t i = threadIdx.x; i < words; i += kV2Threads) bl[i] = K.bloom[i];
  if (threadIdx.x < 256) {
    const uint32_t b = threadIdx.x;
    lt[b] = (b & 0xE0u) == 0xC0u ? (0xC000u | 16u) : ((b & 0xF0u) == 0xE0u ? (0xC0C000u | 24u) : 8u);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lb = (uint32_t)(rows - smem) + (uint32_t)(wave * 64 + lane) * kSkRow;
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t n_pieces = (M.n_bytes + kSkPiece - 1) / kSkPiece + kSkPad;
  const uint64_t n_tiles = (n_pieces + 63) / 64;
  const uint64_t wave_id = (uint64_t)blockIdx.x * (kV2Threads / 64) + wave, n_waves = (uint64_t)gridDim.x * (kV2Threads / 64);
  const uint32_t wshift = 32u - K.log2;
  for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    const uint64_t piece = tile * 64 + lane;
    const int64_t g0 = (int64_t)piece * kSkPiece;
    {
      sk_v4u v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0, va = v0;
      if (g0 + kSkRow <= N) {
        const sk_v4u *p = reinterpret_cast<const sk_v4u *>(M.text + g0);  // (the corpus is 16-byte aligned: capi.cpp)
        v0 = p[0];
        v1 = p[1];
        v2 = p[2];
        v3 = p[3];
        va = p[4];
      } else if (g0 < N) {
        v0 = sk_text16(M.text, g0, N);
        v1 = sk_text16(M.text, g0 + 16, N);
        v2 = sk_text16(M.text, g0 + 32, N);
        v3 = sk_text16(M.text, g0 + 48, N);
        va = sk_text16(M.text, g0 + 64, N);
      }
      sk_v4u *d = reinterpret_cast<sk_v4u *>(smem + lb);
      d[0] = v0;
      d[1] = v1;
      d[2] = v2;
      d[3] = v3;
      d[4] = va;
    }
    uint32_t o = 0, po = kSkPiece, gp = 0, clo = 0, chi = 0;
    for (;;) {
      if (!wany(po < (uint32_t)kSkPiece || o == 0u)) break;
      const uint32_t at = min(o, (uint32_t)(kSkRow - 8));
      const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + lb + (at & ~3u));
      const uint32_t x = __builtin_amdgcn_alignbyte(q[1], q[0], at & 3u);
      const uint32_t e = lt[x & 0xFFu];
      const uint32_t cm = e & 0xFFFF00u;
      const uint32_t s = (x & cm) == (cm & 0x808080u) ? (e & 0xFFu) : 8u;
      const uint32_t c = __builtin_amdgcn_ubfe(x, 0u, s);
      uint32_t h = __umul24(c, K.k1) + gp;
      const uint32_t g = __umul24(c, kSkipKB);
      gp = __builtin_amdgcn_alignbit(g, g, 11);
      h ^= h >> 16;
      const uint32_t w = bl[h >> wshift];
      const uint32_t m = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
      const bool pass = po < (uint32_t)kSkPiece & (w & m) == m;
      const unsigned long long bit = (unsigned long long)(pass ? 1u : 0u) << (po & 63u);
      clo |= (uint32_t)bit;
      chi |= (uint32_t)(bit >> 32);
      po = o;
      o += s >> 3;
    }
    if (piece < n_pieces) {
      unsigned long long v = (unsigned long long)chi << 32 | clo;
      if (g0 >= N)
        v = 0ull;
      else if (g0 + kSkPiece > N)
        v &= ~0ull >> (64 - (uint32_t)(N - g0));  // (no marks behind the text)
      bitmap[piece] = v;
    }
  }
}

// ---- the walk.  Positions are 32-bit offsets from `org`, a multiple of 64 at or below the lane's first byte.
// LDS: decode tables, root table, the waves' event buffers (scan_unit.hip's, without the input rows).
template <int BB>
__global__ __launch_bounds__(kV2Threads) void ks_traverse(UnitDev U, V2Args M, const unsigned long long *__restrict__ bitmap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;
  uint32_t *tabw = reinterpret_cast<uint32_t *>(smem);
  uint32_t *rlw = tabw + kUTabWords;
  const uint32_t n_root = (U.n_syms + 3u) & ~3u;
  for (uint32_t i = threadIdx.x; i < kUTabWords; i += kV2Threads) tabw[i] = U.tables[i];
  for (uint32_t i = threadIdx.x; i < n_root; i += kV2Threads) rlw[i] = i < U.n_syms ? U.root[i] : 0u;
  __syncthreads();
  const uint32_t *rl = rlw;
  const uint4 *t0a = reinterpret_cast<const uint4 *>(tabw + kUT0a);
  const uint2 *t0b = reinterpret_cast<const uint2 *>(tabw + kUT0b);
  const uint8_t *tabb = reinterpret_cast<const uint8_t *>(tabw);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef uint32_t v3u __attribute__((ext_vector_type(3)));
  const uint32_t wbo = (uint32_t)(kUTabWords + n_root) * 4u + 16u + (uint32_t)wave * (64u * 12u);  // the wave's event buffer (byte offset)
  const uint2 *slots = U.slots;
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int64_t S = (int64_t)M.S;
  const int warm = U.max_len > 1 ? (int)U.max_len - 1 : 0;

  // the unit that starts `at` bytes into the window w (at <= 9: a window starts at a multiple of 4 at or below a position, the
  // units at offsets <= 3, <= 6 and <= 9 are what a trip and the jump behind it look at);
  // dend = first byte, in window coordinates, that is not the document's (scan_unit.hip decode)
  auto wdecode = [&](const sk_v4u &w, uint32_t at, int32_t dend, uint32_t &o_code, uint32_t &o_L) {
    const uint32_t lo = at < 4u ? w.x : (at < 8u ? w.y : w.z), hi = at < 4u ? w.y : (at < 8u ? w.z : w.w);
    const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, at & 3u);
    const uint32_t b0 = w4 & 0xFFu;
    const uint4 q0 = t0a[b0];
    const uint2 q1 = t0b[b0];
    const uint32_t s1 = *reinterpret_cast<const uint32_t *>(tabb + q0.x + ((w4 >> 6) & 0x3FCu));
    const uint32_t s2 = *reinterpret_cast<const uint32_t *>(tabb + q0.y + ((w4 >> 14) & 0x3FCu));
    const uint32_t sum = q0.z + s1 + s2;
    const uint32_t want = q1.y;
    const bool in_doc = (int32_t)(at + want) <= dend;
    const bool whole = in_doc & sum < kUPoison;
    const bool good = whole & (sum - q0.w) < q1.x;
    o_L = whole ? want : 1u;
    o_code = good ? sum - kUBias : 0u;
  };

  const uint64_t n_tiles = (M.n_chunks + kV2Threads - 1) / kV2Threads;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t chunk = tile * kV2Threads + threadIdx.x;
    const uint32_t ev_stride = M.ev_stride;
    const bool live = chunk < M.n_chunks;
    const int64_t a = (int64_t)chunk * S;
    const int64_t e = live ? min(a + S, N) : a;
    uint64_t dn = 0;
    int64_t nb = INT64_MAX, doc_start = a, pos0 = e;
    // events: scan_unit.hip's wave buffer (64 records in LDS, 768-byte stores into the wave's part of the event space)
    uint32_t wfill = 0, wout = 0;
    const uint64_t wchunk0 = tile * kV2Threads + (uint64_t)wave * 64;
    uint32_t *wreg = M.evg + wchunk0 * M.ev_stride * 3;
    const uint32_t wcap = (uint32_t)min<uint64_t>(64, M.n_chunks > wchunk0 ? M.n_chunks - wchunk0 : 0) * M.ev_stride;
    uint32_t hits = 0, seq = 0;
    if (live) {
      dn = first_boundary(M.doc_off, D, (uint64_t)a);
      nb = (int64_t)M.doc_off[dn];
      pos0 = a;
      if (nb != a) {
        doc_start = (int64_t)M.doc_off[dn - 1];
        pos0 = a - min<int64_t>(a - doc_start, warm);
      }
    }
    const int64_t org = live ? pos0 & ~(int64_t)63 : 0;  // (a lane without a chunk asks for the text's first bytes)
    const uint8_t *tx = M.text + org;
    const int64_t Nr = N - org;                                   // text bytes from org on
    const uint32_t lim16 = (uint32_t)min<int64_t>(max<int64_t>(Nr - 16, 0), 0x7FFFFFFF);  // last offset a 16-byte window may start at
    const unsigned long long *bmp = bitmap + (org >> 6);
    const uint32_t er = live ? (uint32_t)(e - org) : 0u;          // the lane takes the units that start below it
    const int32_t a_rel = (int32_t)(a - org);
    uint32_t r = (uint32_t)(pos0 - org);                          // position of the next unit
    uint32_t nbr = (uint32_t)min<int64_t>(nb - org, 0x7FFFFFFF);  // next document boundary
    int32_t dbase = (int32_t)(org - doc_start);                   // end offset in the document = dbase + r
    uint32_t E = 0, Ek = 0, pc = 0;  // the state as one word (unit.hpp), the state an event is for, the symbol that led to E
    uint32_t code = 0, L = 0;        // the unit at r
    sk_v4u W = {0, 0, 0, 0};         // sixteen text bytes; the unit at r starts `off` bytes into them
    uint32_t off = 0;
    unsigned long long bm0 = 0, bm1 = 0;  // marks of the positions [64 bw, 64 bw + 128)
    uint32_t bw = 0;
    bool d1 = false;     // E is the one-unit state of the last consumed unit
    bool mql = false;    // ... whose start is marked (a two-unit path may start there)
    bool fresh = true;   // no state, no unit: the lane is about to jump from r (the chunk's start, a document's start)
    bool bmok = false;   // bm0 / bm1 hold the words bw, bw + 1

#ifdef AHA_SK_STATS
    uint32_t st_trips = 0, st_wtrips = 0, st_jumps = 0, st_probes = 0, st_fresh = 0;
#endif
    for (;;) {
      const bool act = live & r < er;
      if (!wany(act)) break;
#ifdef AHA_SK_STATS
      st_wtrips++;
      st_trips += act ? 1u : 0u;
      st_fresh += (act & fresh) ? 1u : 0u;
#endif
      {  // rare: a document starts at r (ac.cr:177: the state is per sequence)
        const bool bnd = act & r == nbr;
        if (wany(bnd)) {
          if (bnd) {
            const int64_t here = org + r;
            do {
              M.doc_ev_rank[dn] = seq;
              M.doc_hit_rank[dn] = hits;
              dn++;
              nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            } while (nb == here);
            asm volatile("" : "+v"(nb));
            nbr = (uint32_t)min<int64_t>(nb - org, 0x7FFFFFFF);
            E = 0;
            pc = 0;
            d1 = false;
            fresh = true;
            doc_start = here;
            dbase = -(int32_t)r;
          }
        }
      }
      {  // rare: a lane without marks in its registers (the chunk's start, behind a document boundary)
        const bool ld = act & fresh & (!bmok | (r >> 6) != bw);
        if (wany(ld)) {
          if (ld) {
            bmok = true;
            bw = r >> 6;
            bm0 = bmp[bw];
            bm1 = bmp[bw + 1];
            asm volatile("" : "+v"(bm0), "+v"(bm1));
          }
        }
      }
      // ---- where the lane goes if this trip leaves it shallow: the first mark at or behind the end of the unit at r
      // (fresh: at or behind r), inside the lane's two words of marks; no mark: their end, a jump like any other (its
      // target is unmarked, so nothing is probed there); never beyond the document or the chunk
      const uint32_t Le = fresh ? 0u : L;
      const uint32_t rel6 = r - (bw << 6);                 // < 64
      const bool mp = ((bm0 >> rel6) & 1ull) != 0ull;      // the unit at r starts at a mark
      const uint32_t sfrom = rel6 + Le;                    // <= 66
      const unsigned long long m0 = sfrom < 64u ? bm0 & (~0ull << sfrom) : 0ull;
      unsigned long long m1 = bm1 & ((1ull << (kSkLimit - 64u)) - 1ull);
      m1 = sfrom > 64u ? m1 & (~0ull << (sfrom - 64u)) : m1;
      uint32_t t = m0 ? (uint32_t)__builtin_ctzll(m0) : (m1 ? 64u + (uint32_t)__builtin_ctzll(m1) : kSkLimit);
      const bool tmark = (((t < 64u ? bm0 >> t : bm1 >> (t - 64u))) & 1ull) != 0ull;
      const uint32_t stop = min(nbr, er);
      const uint32_t T1 = min((bw << 6) + t, stop);
      // ---- requests, issued by hand and waited for ONCE in front of their first use (scan_unit.hip's way with its probe:
      // left to hipcc every load is followed by its own wait -- four memory round trips per trip, 4.0 ms per GiB instead of
      // what is measured now): the text at the jump target, the text behind the unit at r, the next word of marks, the probe.
      // Every lane asks (the idle ones for the chunk's first bytes); a window that would reach beyond the text is asked for
      // 16 bytes early and put right below (rare: the last bytes of the batch).
      // Only the lanes that need them ask, and the windows start at multiples of 4 (measured, profiles/r06_skip_engine.txt: every
      // lane asking for all four costs 6.5 ms per GiB against 3.9; a 16-byte load from an odd address costs the memory pipeline
      // ~8.6 cycles per lane, from a multiple of 4 ~3.6).
      sk_v4u Bw = {0, 0, 0, 0}, Aw = Bw;
      unsigned long long Cw = 0;
      const uint32_t gB = min(T1, lim16) & ~3u, gA = min(r + Le, lim16) & ~3u;
      if (act) {
        const uint8_t *pB = tx + gB;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Bw) : "v"(pB) : "memory");
        if (!fresh) {
          const uint8_t *pA = tx + gA;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(Aw) : "v"(pA) : "memory");
        }
        if (t + 3u >= 64u | sfrom >= 64u) {
          const unsigned long long *pC = bmp + (bw + 2u);
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(Cw) : "v"(pC) : "memory");
        }
      }
      const bool trip = act & !fresh;
      uint32_t evc = 0, endr = 0;
      bool jump = act & fresh;
      {
        const bool good = code != 0u;
        const uint32_t Bq = u_child(E, BB);
        const bool hdr = u_hdr_pending(E);
        const bool grp = Bq >= U.big_lo & code >= U.n_low & !hdr;
        uint32_t se = grp ? (code >> 5) + U.g0 : code;
        se = hdr ? 0u : se;
        const uint32_t fc = code & 7u;
        const bool probe = trip & good & (((E | 0x20000000u) >> ((uint32_t)BB + fc)) & 1u) != 0u & Bq != 0u;
        unsigned long long enw;
        {
          const uint2 *ap = slots + (probe ? (Bq ^ se) : 0u);
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(enw) : "v"(ap) : "memory");
        }
        uint32_t n_code = 0, n_L = 1, rt = 0, rf = 0;
        if (trip) {
          wdecode(W, off + L, (int32_t)(nbr - r + off), n_code, n_L);  // the next unit, while the requests are in flight
          rt = rl[code];
          rf = rl[pc];
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(enw), "+v"(Bw), "+v"(Aw), "+v"(Cw) : : "memory");
        {  // rare: a window at the very end of the text was asked for up to 15 bytes early
          const bool fixB = act & T1 > lim16, fixA = trip & r + Le > lim16;
          if (wany(fixB | fixA)) {
            if (fixB) Bw = sk_text16(tx, (int64_t)(T1 & ~3u), Nr);
            if (fixA) Aw = sk_text16(tx, (int64_t)((r + Le) & ~3u), Nr);
          }
        }
        if (trip) {
          const uint32_t enx = (uint32_t)enw, eny = (uint32_t)(enw >> 32);
          const bool symhit = probe & u_sym(eny) == se & !grp;
          const bool hit = symhit & !hdr;
          const bool redir = grp & probe & ((enx >> (code & 31u)) & 1u) != 0u;
          const uint32_t rE = (__builtin_popcount(enx & ~(~0u << (code & 31u))) + eny ^ code) | u_all_filter(BB);
          // a miss where the fail link is the root, where the unit matches nothing, or where the fail state is the one-unit
          // state of an UNMARKED unit (no two-unit path starts there, so that state misses too and fails to the root): the
          // root's table answers in this trip.  Else the unit is tried again in the fail state (scan_unit.hip).
          const bool viaroot = !symhit & !redir & (!u_nfr(E) | !good | (u_f1(E) & !mql));
          const bool f1fail = !symhit & !redir & !viaroot & u_f1(E);
          const uint32_t ft = u_f1(E) ? (rf & 0x7FFFFFFFu) : (Bq | u_all_filter(BB) | 0x20000000u);
          uint32_t missE = viaroot ? rt : ft;
          missE = redir ? rE : missE;
          E = symhit ? enx : missE;
          const bool consumed = hit | viaroot;
          const bool end = consumed & u_end(E);
          const uint32_t c4 = hit ? u_c4(eny) : 1u;
          d1 = consumed ? (viaroot & E != 0u) : (f1fail ? true : (hdr ? false : d1));
          if (consumed) {
            pc = code;
            mql = mp;
            endr = r + L;
            const int32_t last = (int32_t)endr - 1;
            evc = (end & last >= a_rel & last < (int32_t)er) ? c4 : 0u;
            Ek = E;
            // the early fail: the state just entered fails to a one-unit state (NFR + F1) and its own filter says the next
            // unit does not continue it -- the next trip would miss without a probe and fall to root[the unit just consumed]
            const uint32_t nfc = n_code & 7u;
            const uint32_t fm = ((1u << BB) << nfc) | 0x60000000u;
            const bool ef = hit & nfc < 7u & (E & fm) == 0x60000000u;
            E = ef ? (rt & 0x7FFFFFFFu) : E;
            d1 = d1 | ef;
            jump = (E == 0u | d1) & !(d1 & mql);
            r = endr;
            W = Aw;
            off = endr & 3u;
            code = n_code;
            L = n_L;
          }
        }
      }
      // ---- the jump: the unit at the target comes from the root's table, the one behind it is the next trip's
#ifdef AHA_SK_STATS
      st_jumps += jump ? 1u : 0u;
#endif
      if (wany(jump)) {
        if (jump) {
          r = T1;
          E = 0;
          pc = 0;
          d1 = false;
          fresh = T1 >= stop;  // the document's or the chunk's end: nothing to take (a boundary makes the lane fresh anyway)
          if (!fresh) {
            uint32_t c0, l0;
            const uint32_t o0 = T1 & 3u;  // (the window starts at T1 & ~3)
            const int32_t dendB = (int32_t)(nbr - T1 + o0);
            wdecode(Bw, o0, dendB, c0, l0);
            E = rl[c0] & 0x7FFFFFFFu;  // (no key is a single unit: nothing to report)
            pc = c0;
            d1 = E != 0u;
            mql = tmark;
            r = T1 + l0;
            W = Bw;
            off = o0 + l0;
            wdecode(Bw, o0 + l0, dendB, code, L);
          }
        }
      }
      if ((r >> 6) > bw) {  // (at most one word: the jump's unit ends below 64 bw + 128)
        bm0 = bm1;
        bm1 = Cw;
        bw++;
      }
      const uint64_t evm = wballot(evc != 0u);
      if (evm) {
        const bool ev = evc != 0u;
        const uint32_t my = wfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(evm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)evm, 0u));
        const uint32_t rx = u_rec_x(u_child(Ek, BB), (uint32_t)lane, evc, BB), rz = u_rec_z(hits, evc, BB);
        const uint32_t ry = (uint32_t)(dbase + (int32_t)endr);
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
    {  // the rest of the wave's buffer
      const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + (wbo + __umul24((uint32_t)lane, 12u)));
      const v3u rr = {q[0], q[1], q[2]};
      if ((uint32_t)lane < wfill && wout + (uint32_t)lane < wcap) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = rr;
    }
#ifdef AHA_SK_STATS
    atomicAdd(&M.cursor[8], (unsigned long long)st_trips);
    atomicAdd(&M.cursor[9], (unsigned long long)st_jumps);
    atomicAdd(&M.cursor[11], (unsigned long long)st_fresh);
    if (lane == 0) atomicAdd(&M.cursor[10], (unsigned long long)st_wtrips);
#endif
    if (live) {
      M.ev_cnt[chunk] = seq;
      if (seq > ev_stride) M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
      M.chunk_hits[chunk] = hits;
      if (e == N) {  // documents that start at N (empty tail documents, and d = D)
        while (dn <= D) {
          M.doc_ev_rank[dn] = seq;
          M.doc_hit_rank[dn] = hits;
          dn++;
        }
      }
    }
  }
}

}  // namespace

size_t skip_bitmap_bytes(uint64_t n_bytes) { return ((n_bytes + kSkPiece - 1) / kSkPiece + kSkPad) * 8; }

int skip_prepare(uint32_t n_syms, uint32_t log2_words) {
  int e = (int)hipFuncSetAttribute((const void *)ks_mark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk_mark_lds(log2_words));
  if (!e) e = (int)hipFuncSetAttribute((const void *)ks_traverse<22>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sk_walk_lds(n_syms));
  return e;
}

void skip_launch_mark(const SkipDev &K, const V2Args &M, void *bitmap, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(ks_mark, dim3(grid), dim3(kV2Threads), sk_mark_lds(K.log2), (hipStream_t)stream, K, M,
                     reinterpret_cast<unsigned long long *>(bitmap));
}

void skip_launch_traverse(const UnitDev &U, const V2Args &M, const void *bitmap, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(ks_traverse<22>, dim3(grid), dim3(kV2Threads), sk_walk_lds(U.n_syms), (hipStream_t)stream, U, M,
                     reinterpret_cast<const unsigned long long *>(bitmap));
}

}  // namespace aha
