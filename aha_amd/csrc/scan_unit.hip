// scan_unit.hip -- the character-level traversal for gfx950: the same persistent-workgroup walk as scan_v2.hip's
// k2_traverse, but over the UNIT image (unit.hpp): a trip consumes one UTF-8-shaped unit -- one, two or three bytes --
// with at most ONE probe of the double array, and a miss needs no header load because the entry that led to the
// current state already told where it fails to.  Replaces src/aha/ac.cr:176-192 (match_) for key sets that are
// sequences of whole units (unit.hpp has the exactness argument); reports through the same per-chunk event regions,
// so count / scan / expansion (scan_v2.hip, k2d_*) are shared.
//
// Why it exists (profiles/r03_trip_anatomy.txt): the byte-level trip is bound by VALU issue, 63 vector instructions
// per byte.  This trip costs about as many per CHARACTER: the unit is decoded by table lookups (LDS reads are nearly
// free: unit.hpp, SYMBOLS) into a symbol of a dense alphabet, the root's transitions are one directly indexed LDS
// table over that alphabet, and an 8-bit filter in the root entry answers most "does the character after this one
// continue a key" questions without a probe (0.2 probes per byte leave LDS instead of 0.5).
//
// LDS: the root's transitions (4 bytes per symbol) + the decode tables (11 KiB) + a wave-private input window (32
// bytes + the last 4 bytes of the previous piece per lane, rows of 9 dwords: odd stride, no bank conflicts).
// Everything else is probed in HBM/L2, 8 bytes per probe.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"
#include "unit.hpp"

namespace aha {

namespace {

constexpr int kUPiece = 32;         // input bytes staged per lane per round (64-byte rounds with a 3-byte root table were
                                    // measured too: 3 % faster, but the table needs its fourth byte for the filter)
constexpr int kURow = kUPiece + 4;  // bytes per lane in the LDS input window (9 dwords: odd stride)
constexpr int kUWave = 64 * kURow;

// 16 text bytes at g (any alignment of the corpus end; the corpus itself is 16-byte aligned, g is a multiple of 16)
__device__ __forceinline__ uint4 load16(const uint8_t *text, int64_t g, int64_t N) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (g >= 0 && g + 16 <= N) {
    v = *reinterpret_cast<const uint4 *>(text + g);
  } else if (g >= 0 && g < N) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16 && g + j < N; j++) w[j >> 2] |= (uint32_t)text[g + j] << ((j & 3) * 8);
    v = make_uint4(w[0], w[1], w[2], w[3]);
  }
  return v;
}

__global__ __launch_bounds__(kV2Threads) void ku_traverse(UnitDev U, V2Args M) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // LDS: decode tables (16-byte aligned), the root's transitions (child base | filter << 21 | END << 31), input rows
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
  uint8_t *in_base = smem + (size_t)(kUTabWords + n_root) * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *inl = in_base + wave * kUWave + lane * kURow;
  const uint32_t *row = reinterpret_cast<const uint32_t *>(inl);
  const uint2 *slots = U.slots;
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int64_t S = (int64_t)M.S;
  const int rounds = (int)(M.S / kUPiece);
  const int warm = U.max_len > 1 ? (int)U.max_len - 1 : 0;
  const int R = (warm + kUPiece - 1) / kUPiece;  // warm-up rounds before the chunk

  const uint64_t n_tiles = (M.n_chunks + kV2Threads - 1) / kV2Threads;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t chunk = tile * kV2Threads + threadIdx.x;
    uint2 *evreg = M.evd + chunk * M.ev_stride;
    const uint32_t ev_stride = M.ev_stride;
    const bool live = chunk < M.n_chunks;
    const int64_t a = (int64_t)chunk * S;
    const int64_t e = live ? min(a + S, N) : a;
    uint64_t dn = 0;
    int64_t nb = INT64_MAX, doc_start = a, pos = e;
    uint32_t B = 0, fb = 0, seq = 0;  // state (0 = root), its fail state
    uint32_t flt = 0xFFu;             // filter of the state's transitions (0xFF: none known, probe)
    bool ffr = true;                  // the fail state's own fail link is the root
    uint4 half[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};  // second half of the current input line
    int64_t half_pb = INT64_MIN;                                       // the piece `half` holds
    uint32_t tail = 0;  // last four bytes of the piece staged in the previous round
    if (live) {
      dn = first_boundary(M.doc_off, D, (uint64_t)a);
      nb = (int64_t)M.doc_off[dn];
      pos = a;
      if (nb != a) {
        doc_start = (int64_t)M.doc_off[dn - 1];
        pos = a - min<int64_t>(a - doc_start, warm);
      }
    }

    // one round beyond the chunk: a unit that starts in the last bytes of the chunk and continues in the next one
    for (int r = -R; r <= rounds; r++) {
      const int64_t pb = a + (int64_t)r * kUPiece;
      const int64_t pend = min(pb + kUPiece, e);
      const bool need = live && pos < pend;
      if (!__any(need)) continue;
      // Row = the last 4 bytes of the previous piece + this piece: a unit whose bytes are not all here yet is left for
      // the next round (the lane parks), so no byte is requested ahead of its round.
      uint32_t rel = kURow, lim = 0;  // row coordinates: text position = pb - 4 + rel; inactive: rel >= lim
      if (need) {
        // a piece is half a 64-byte line: the round that starts a line also loads its second half into registers,
        // so both halves are requested while the line is in flight and every input line is fetched once
        const bool line_start = (pb & 63) == 0;
        uint4 v0, v1;
        if (!line_start && half_pb == pb) {
          v0 = half[0];
          v1 = half[1];
        } else {
          v0 = load16(M.text, pb, N);
          v1 = load16(M.text, pb + 16, N);
        }
        half_pb = INT64_MIN;
        if (line_start && pb >= 0 && pb + 2 * kUPiece <= N && pb + kUPiece < e) {
          half[0] = *reinterpret_cast<const uint4 *>(M.text + pb + kUPiece);
          half[1] = *reinterpret_cast<const uint4 *>(M.text + pb + kUPiece + 16);
          half_pb = pb + kUPiece;
        }
        uint32_t *dst = reinterpret_cast<uint32_t *>(inl);
        dst[0] = tail;
        dst[1] = v0.x;
        dst[2] = v0.y;
        dst[3] = v0.z;
        dst[4] = v0.w;
        dst[5] = v1.x;
        dst[6] = v1.y;
        dst[7] = v1.z;
        dst[8] = v1.w;
        tail = v1.w;
        rel = (uint32_t)(pos - pb + 4);
        lim = (uint32_t)(pend - pb + 4);  // units that START before pend
      }
      // limits in row coordinates: the next document boundary (when the row shows it), the end of the text, the
      // range of end positions this lane reports ([a, e): a unit that ends in the next chunk is that chunk's)
      uint32_t nb_rel = (nb >= pb - 4 && nb <= pb + kUPiece) ? (uint32_t)(nb - pb + 4) : ~0u;
      const uint32_t n_rel = (N - pb) <= (int64_t)kUPiece ? (uint32_t)max<int64_t>(N - pb + 4, 0) : ~0u;
      const int32_t a_rel = (int32_t)max<int64_t>(a - pb + 4, -128);
      const int32_t e_rel = (int32_t)min<int64_t>(e - pb + 4, 128);
      int32_t docrel = (int32_t)(pb - 4 - doc_start);  // end offset in the document = docrel + rel (after the unit)

      for (;;) {  // outer: resolve document boundaries, then run the trips up to the next one
        const bool bnd = rel < lim && rel == nb_rel;
        if (__any(bnd)) {  // rare: a document starts here (ac.cr:177: the state is per sequence)
          if (bnd) {
            const int64_t here = pb - 4 + rel;
            do {
              M.doc_ev_rank[dn] = seq;
              dn++;
              nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            } while (nb == here);
            asm volatile("" : "+v"(nb));  // retire the load inside this block
            nb_rel = (nb <= pb + kUPiece) ? (uint32_t)(nb - pb + 4) : ~0u;
            B = 0;
            fb = 0;
            ffr = true;
            flt = 0xFFu;
            doc_start = here;
            docrel = -(int32_t)rel;
          }
        }
        uint32_t lim2 = min(lim, nb_rel);          // lanes park at the next boundary
        const uint32_t dend = min(nb_rel, n_rel);  // first byte that is not this document's
        for (;;) {
          const bool act = rel < lim2;
          if (!__any(act)) break;
          bool ev = false;
          if (act) {
            // ---- the unit at rel (unit.hpp).  Every select below picks between values that are already computed
            // (plain locals): that keeps them v_cndmask instead of nested divergent branches, which cost more than
            // the work they skip.
            const uint32_t lo = row[rel >> 2], hi = row[(rel >> 2) + 1];
            const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, rel & 3u);
            // table-driven decode (unit.hpp, SYMBOLS): the first byte tells the length the unit would have, where its
            // second and third byte are looked up (a poison value unless they are continuation bytes) and the window
            // of its class; the sum is the symbol when it falls into that window
            const uint32_t b0 = w4 & 0xFFu;
            const uint4 q0 = t0a[b0];
            const uint2 q1 = t0b[b0];
            const uint32_t s1 = *reinterpret_cast<const uint32_t *>(tabb + q0.x + ((w4 >> 6) & 0x3FCu));
            const uint32_t s2 = *reinterpret_cast<const uint32_t *>(tabb + q0.y + ((w4 >> 14) & 0x3FCu));
            const uint32_t sum = q0.z + s1 + s2;
            const uint32_t want = q1.y;                  // bytes the first byte announces
            const bool in_doc = rel + want <= dend;      // else: a lead byte without its continuation bytes (bad)
            const bool later = in_doc & rel + want > (uint32_t)kURow;  // its bytes are not all staged yet: next round
            const bool whole = in_doc & sum < kUPoison;  // a well-formed unit
            const bool good = whole & (sum - q0.w) < q1.x;  // ... of the keys' alphabet
            const uint32_t L = whole ? want : 1u;
            const uint32_t code = good ? sum - kUBias : 0u;  // symbol 0 has no transition anywhere
            // ---- the root's transition on it (LDS)
            const uint32_t rt = rl[code];
            // ---- the state's own transition: one 8-byte probe
            // a depth-1 state brought a filter over the codes it continues on: a clear bit is a miss without the probe
            const bool probe = B != 0u & good & !later & ((flt >> (code & 7u)) & 1u) != 0u;
            const uint32_t pidx = B ^ code;
            const uint2 en = slots[probe ? pidx : 0u];
            const bool hit = probe & u_sym(en.y) == code;
            // the fail link is the root (or the unit takes the state to the root): the root's table answers
            const bool viaroot = !hit & !later & (!good | B == 0u | fb == 0u);
            const bool fall = !hit & !viaroot & !later;  // continue in the fail state, the unit is tried again there
            const bool needh = fall & !ffr;
            uint2 hd = make_uint2(0, 0);
            if (__any(needh)) hd = slots[needh ? fb : 0u];  // rare: the fail state's header says where IT fails to
            const bool keep0 = viaroot | ffr;  // (on a miss) the new fail state is the root
            const uint32_t en_child = u_child(en.x), en_fail = u_fail(en.x, en.y);
            const uint32_t hd_fail = u_fail(hd.x, hd.y);
            const uint32_t rt_child = good ? u_child(rt) : 0u;
            const bool en_ffr = u_ffr(en.y), hd_ffr = u_ffr(hd.y), en_end = u_end(en.x), rt_end = good & u_end(rt);
            const uint32_t missB = viaroot ? rt_child : fb;
            const uint32_t missF = keep0 ? 0u : hd_fail;
            const bool missR = keep0 | hd_ffr;
            const bool missE = viaroot & rt_end;
            const uint32_t newB = hit ? en_child : missB;
            const uint32_t newF = hit ? en_fail : missF;
            const bool newR = hit ? en_ffr : missR;
            const uint32_t rt_flt = u_filter(rt);
            const uint32_t missT = viaroot ? rt_flt : 0xFFu;
            const uint32_t newT = hit ? 0xFFu : missT;
            flt = later ? flt : newT;
            B = later ? B : newB;
            fb = later ? fb : newF;
            ffr = later ? ffr : newR;
            const bool end = hit ? en_end : missE;
            const bool consumed = hit | viaroot;
            const uint32_t adv = consumed ? L : 0u;
            rel += adv;
            lim2 = later ? rel : lim2;  // parked until the next round brings the rest of the unit
            lim = later ? rel : lim;
            const int32_t last = (int32_t)rel - 1;  // row index of the unit's last byte
            // is_end? -> fetch later (ac.cr:183-185); this lane reports the end positions in [a, e)
            ev = consumed & end & last >= a_rel & last < e_rel;
          }
          if (__any(ev)) {
            if (ev) {
              if (seq < ev_stride) {
                typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                const v2u rec = {B, (uint32_t)(docrel + (int32_t)rel)};
                *reinterpret_cast<v2u *>(evreg + seq) = rec;
              } else {
                M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
              }
              seq++;
            }
          }
        }
        if (!__any(rel < lim)) break;
      }
      if (need) pos = pb - 4 + rel;
    }
    if (live) {
      M.ev_cnt[chunk] = seq;
      if (e == N) {  // documents that start at N (empty tail documents, and d = D)
        while (dn <= D) {
          M.doc_ev_rank[dn] = seq;
          dn++;
        }
      }
    }
  }
}

}  // namespace

size_t unit_lds_bytes(uint32_t n_syms) {
  return (size_t)(kUTabWords + ((n_syms + 3u) & ~3u)) * 4 + (size_t)(kV2Threads / 64) * kUWave + 16;
}

int unit_prepare(uint32_t n_syms) {
  return (int)hipFuncSetAttribute((const void *)ku_traverse, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)unit_lds_bytes(n_syms));
}

void unit_launch_traverse(const UnitDev &U, const V2Args &M, uint32_t grid, void *stream) {
  hipLaunchKernelGGL(ku_traverse, dim3(grid), dim3(kV2Threads), unit_lds_bytes(U.n_syms), (hipStream_t)stream, U, M);
}

}  // namespace aha
