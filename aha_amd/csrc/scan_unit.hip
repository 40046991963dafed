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
// LDS: the root's transitions (4 bytes per symbol) + the decode tables (11 KiB) + a wave-private input window (three
// 16-byte pieces + the last 4 bytes before them per lane, rows of 13 dwords: odd stride, no bank conflicts).
// Everything else is probed in HBM/L2, 8 bytes per probe.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"
#include "unit.hpp"

namespace aha {

namespace {

// Input window of a lane: the last 4 bytes before it + three pieces of 16 bytes.  A round appends one piece and ends as
// soon as EVERY lane of the wave has left the oldest one, so a lane may run up to two pieces ahead of the slowest:
// characters have different lengths, and with one piece per round (the lanes of a wave in step at every piece) a
// wave ran 66 % of its lane-trips usefully on the cfg 3 mix; with this window 90 % (simulated, and measured).
constexpr int kUPiece = 16;
constexpr int kUWin = 3 * kUPiece;
constexpr int kURow = kUWin + 4;    // bytes per lane in LDS (13 dwords: odd stride, no bank conflicts)
constexpr int kUWave = 64 * kURow;

// 16 text bytes at g (any alignment of the corpus end; the corpus itself is 16-byte aligned, g is a multiple of 16)
__device__ __forceinline__ uint4 load16(const uint8_t *text, int64_t g, int64_t N) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (g >= 0 && g + 16 <= N) {
    v = *reinterpret_cast<const uint4 *>(text + g);
  } else if (g >= 0 && g < N) {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // (rare: the text's last bytes; kept as a loop)
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

// wave votes straight from the compare's mask (hipcc's __any/__all go through a 0/1 VGPR and a second compare)
__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool wall(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

__host__ __device__ inline size_t u_lds_rows(uint32_t n_syms) {  // decode tables, root table, input rows
  return (size_t)(kUTabWords + ((n_syms + 3u) & ~3u)) * 4 + (size_t)(kV2Threads / 64) * kUWave + 16;
}
__host__ __device__ inline size_t u_lds(uint32_t n_syms) {
  return u_lds_rows(n_syms) + (size_t)(kV2Threads / 64) * 64 * 12;  // + the waves' event buffers
}

// CHARS (String overload, matcher.cr:34-39): an event's second word is the number of characters -- bytes outside
// 0x80..0xBF: one per unit that does not start with a stray continuation byte -- of the lane's chunk up to and
// including the unit, << 1 | "counted from the start of the document" (the format of k2_traverse<.., CHARS>: the
// expansion adds the characters between the start of the document and the chunk).
// HB ("header beside"): the word of the fail state -- the state's header slot -- is requested beside the probe by the lanes whose
// state has one, and a miss continues in the fail state in the SAME trip.  A second load in every trip (+3.5 % on cfg 3, whose
// text rarely falls out of a deep match) against a trip per header (-8.5 % on cfg 5: 1.19 trips per byte); chosen per image when
// it is compiled (capi.cpp: the share of states that own a header), reported in aha_ac_info_t.unit_header_beside.
template <bool CHARS, int BB, bool HB>
__global__ __launch_bounds__(kV2Threads) void ku_traverse(UnitDev U, V2Args M) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;  // the doc offsets are not what the call says (k_check_docs ran in front): index nothing
  // LDS: decode tables (16-byte aligned), the root's transitions (state words, unit.hpp: base | filter << BB | F1 | NFR | END), input rows
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
  typedef uint32_t v3u __attribute__((ext_vector_type(3)));
  const uint32_t wbo = (uint32_t)u_lds_rows(U.n_syms) + (uint32_t)wave * (64u * 12u);  // the wave's event buffer (byte offset)
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
    const uint32_t ev_stride = M.ev_stride;
    const bool live = chunk < M.n_chunks;
    const int64_t a = (int64_t)chunk * S;
    const int64_t e = live ? min(a + S, N) : a;
    uint64_t dn = 0;
    int64_t nb = INT64_MAX, doc_start = a, pos = e;
    // Events leave through a buffer of 64 records per wave in LDS and go out 512 bytes at a time into the wave's part
    // of the event space (the regions of its 64 chunks, taken together), tagged with the lane: a store per lane and event
    // into the lane's own region put a scattered 8-byte store -- and its acknowledgement, which the next probe waits for
    // through the shared vmcnt -- into almost every trip (-0.28 ms per GiB on cfg 3, profiles/r03_unit_lab.txt).
    // ku_regroup sorts them back into the regions, chunk by chunk in order.
    uint32_t wfill = 0, wout = 0;  // records in the buffer / already written (wave-uniform)
    const uint64_t wchunk0 = tile * kV2Threads + (uint64_t)wave * 64;
    uint32_t *wreg = M.evg + wchunk0 * M.ev_stride * 3;
    const uint32_t wcap = (uint32_t)min<uint64_t>(64, M.n_chunks > wchunk0 ? M.n_chunks - wchunk0 : 0) * M.ev_stride;
    uint32_t hits = 0;        // hits the lane's events stand for (exact while no event stands for more than 15)
    uint32_t lead_total = 0;
    uint32_t lc = 0, lc_exact = 0;  // CHARS: characters of the chunk so far, or (lc_exact) of the document that started in it
    uint32_t E = 0, seq = 0;  // the state as one word (unit.hpp): base | filter << 22 | F1 | NFR | END; 0 = the root
    uint32_t pc = 0;          // the symbol that led to it
    uint4 q1 = make_uint4(0, 0, 0, 0), q2 = q1, q3 = q1;  // the rest of the input line whose first piece was staged last
    uint32_t w[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // the window's bytes (the LDS row is rewritten from them)
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

    // Round r: the window holds the pieces r, r + 1, r + 2 of the lane's chunk (piece r + 2 is loaded now; the walk
    // starts in piece -R, which the first round loads).  One round beyond the chunk: a unit that starts in the last
    // bytes of the chunk continues in the next one.
    // (the first piece loaded starts a 64-byte line, so that the line queue below is filled: R is rounded up to whole lines)
    for (int r = -((R + 3) / 4) * 4 - 2; r <= rounds; r++) {
      const int64_t pb = a + (int64_t)r * kUPiece;
      const int64_t pl = pb + 2 * kUPiece;  // the piece loaded in this round
      const int64_t pend = min(pb + kUWin, e);
      {
        // a 64-byte line is four pieces: the round that starts a line loads all of it (three pieces wait in registers),
        // so every input line is requested once while it is in flight.  Chunk starts are multiples of 64: uniform.
        uint4 v;
        if ((pl & 63) == 0) {
          v = load16(M.text, pl, N);
          q1 = load16(M.text, pl + 16, N);
          q2 = load16(M.text, pl + 32, N);
          q3 = load16(M.text, pl + 48, N);
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
      if (!wany(need)) continue;
      // row coordinates: text position = pb - 4 + rel; inactive: rel >= lim; a unit whose bytes are not all in the
      // window yet is left for a later round (the lane parks)
      uint32_t rel = kURow, lim = 0;
      if (need) {
        rel = (uint32_t)(pos - pb + 4);
        lim = (uint32_t)(pend - pb + 4);  // units that START before pend
      }
      // limits in row coordinates: the next document boundary (when the row shows it), the end of the text, the
      // range of end positions this lane reports ([a, e): a unit that ends in the next chunk is that chunk's)
      uint32_t nb_rel = (nb >= pb - 4 && nb <= pb + kUWin) ? (uint32_t)(nb - pb + 4) : ~0u;
      const uint32_t n_rel = (N - pb) <= (int64_t)kUWin ? (uint32_t)max<int64_t>(N - pb + 4, 0) : ~0u;
      const int32_t a_rel = (int32_t)max<int64_t>(a - pb + 4, -128);
      const int32_t e_rel = (int32_t)min<int64_t>(e - pb + 4, 128);
      int32_t docrel = (int32_t)(pb - 4 - doc_start);  // end offset in the document = docrel + rel (after the unit)

      for (;;) {  // outer: resolve document boundaries, then run the trips up to the next one
        const bool bnd = rel < lim && rel == nb_rel;
        if (wany(bnd)) {  // rare: a document starts here (ac.cr:177: the state is per sequence)
          if (bnd) {
            const int64_t here = pb - 4 + rel;
            do {
              M.doc_ev_rank[dn] = seq;
              M.doc_hit_rank[dn] = hits;
              if (CHARS) M.doc_lead_rank[dn] = lead_total;
              dn++;
              nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            } while (nb == here);
            asm volatile("" : "+v"(nb));  // retire the load inside this block
            nb_rel = (nb <= pb + kUWin) ? (uint32_t)(nb - pb + 4) : ~0u;
            E = 0;
            pc = 0;
            lc = 0;
            lc_exact = 1;
            doc_start = here;
            docrel = -(int32_t)rel;
          }
        }
        uint32_t lim2 = min(lim, nb_rel);          // lanes park at the next boundary
        const uint32_t dend = min(nb_rel, n_rel);  // first byte that is not this document's
        // The unit at row position `at` (unit.hpp): table-driven decode -- the first byte tells the length the unit
        // would have, where its second and third byte are looked up (a poison value unless they are continuation
        // bytes) and the window of its class; the sum is the symbol when it falls into that window.  Three dependent
        // LDS reads: the hot loop decodes the NEXT unit while the current one is probed, so they are off its chain.
        auto decode = [&](uint32_t at, uint32_t &o_code, uint32_t &o_L, bool &o_later) {
          const uint32_t lo = row[at >> 2], hi = row[(at >> 2) + 1];
          const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, at & 3u);
          const uint32_t b0 = w4 & 0xFFu;
          const uint4 q0 = t0a[b0];
          const uint2 q1 = t0b[b0];
          const uint32_t s1 = *reinterpret_cast<const uint32_t *>(tabb + q0.x + ((w4 >> 6) & 0x3FCu));
          const uint32_t s2 = *reinterpret_cast<const uint32_t *>(tabb + q0.y + ((w4 >> 14) & 0x3FCu));
          const uint32_t sum = q0.z + s1 + s2;
          const uint32_t want = q1.y;                 // bytes the first byte announces
          const bool in_doc = at + want <= dend;      // else: a lead byte without its continuation bytes (bad)
          o_later = in_doc & at + want > (uint32_t)kURow;  // its bytes are not all staged yet: next round
          const bool whole = in_doc & sum < kUPoison;  // a well-formed unit
          const bool o_good = whole & (sum - q0.w) < q1.x;  // ... of the keys' alphabet
          o_L = whole ? want : 1u;
          if (CHARS) o_L |= (b0 & 0xC0u) != 0x80u ? 0x100u : 0u;  // the unit is a character (it does not start with a stray continuation byte)
          o_code = o_good ? sum - kUBias : 0u;         // symbol 0 has no transition anywhere
        };
        uint32_t code, L;
        {
          bool later;  // the unit's bytes are not all staged yet: the lane parks until the next round brings the rest
          decode(min(rel, (uint32_t)kURow), code, L, later);
          lim2 = later ? min(lim2, rel) : lim2;
          lim = later ? min(lim, rel) : lim;
        }
        bool all_left = false;  // every lane has left the oldest piece (or has nothing more to do): next round
        for (;;) {
          const bool act = rel < lim2;
          all_left = wall(rel >= (uint32_t)(4 + kUPiece) || rel >= lim);
          if (all_left || !wany(act)) break;
          // The probe is issued and waited for by hand (cdna_hip_programming.md 5.7: an asm load hidden from hipcc): the load
          // first, by every lane (the idle ones ask for slot 0), then the trip's LDS work, then ONE wait right in front of the
          // first use.  Left to hipcc the same trip is 9 % slower (2.44 against 2.20 ms on one box): its wait insertion
          // is conservative around the exec-masked region.  The ISA is audited for reads or copies of the destination
          // between load and wait (tools/audit_probe.py, tests/test_host_logic.py).
          // (Tried on top and dropped: the store of a full event buffer behind the NEXT trip's probe, that probe consumed
          // at vmcnt(1) -- 2.33 against 2.20 ms: profiles/r04_two_walks.txt.)
          auto trip = [&]() -> uint32_t {
            uint32_t evc = 0;  // the lane reports: the hits the event stands for (a one-character state: its own key and no more)
            // ---- the state's own transition: at most one 8-byte probe.  The state word carries a filter over the
            // symbols the state continues on: a clear bit is a miss without the probe (the root's word is 0).
            // What the probe is keyed by (unit.hpp, IMAGE): the unit's symbol; the group of 32 symbols around it when the
            // state is a big one and the symbol a high one; 0 -- the header, which holds the word of the fail state --
            // when the trip before asked for it.  A big state's base is its block's first slot: ^ is + there.
            // (requested by every lane -- the idle ones ask for slot 0 -- so that the store below runs unmasked)
            const bool good = code != 0u;
            const uint32_t Bq = u_child(E, BB);
            const bool hdr = u_hdr_pending(E);
            const bool grp = Bq >= U.big_lo & code >= U.n_low & !hdr;
            uint32_t se = grp ? (code >> 5) + U.g0 : code;
            se = hdr ? 0u : se;
            // (bit 29 -- F1 -- stands in for the filter's eighth bit, which is always set)
            // (with 23-bit bases the filter has six bits: classes 6 and 7 share the forced one)
            const uint32_t fc = BB == 22 ? (code & 7u) : min(code & 7u, 6u);
            const bool probe = act & good & (((E | 0x20000000u) >> ((uint32_t)BB + fc)) & 1u) != 0u & Bq != 0u;
            unsigned long long enw = 0;
            uint32_t hdw = 0;
            {
              const uint2 *ap = slots + (probe ? (Bq ^ se) : 0u);
              asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(enw) : "v"(ap) : "memory");
              if (HB) {  // the header of a state whose fail state is neither the root nor a one-character state
                const uint2 *hp = slots + ((act & u_nfr(E) & !u_f1(E)) ? Bq : 0u);
                asm volatile("global_load_dword %0, %1, off" : "=v"(hdw) : "v"(hp) : "memory");
              }
            }
            if (act) {
              uint32_t n_code, n_L;
              bool n_later;
              const uint32_t Lb = CHARS ? (L & 0xFFu) : L;  // (CHARS: bit 8 = the unit is a character)
              decode(rel + Lb, n_code, n_L, n_later);  // rows are padded: rel + 3 + 8 bytes stay inside LDS
              // Every select below picks between values that are already computed (plain locals): that keeps them
              // v_cndmask instead of nested divergent branches, which cost more than the work they skip.
              // ---- the root's transitions (LDS) on the unit (symbol 0 -- a bad unit or one outside the alphabet -- has
              // none) and on the symbol that led to the current state: the one-character state a two-character state
              // fails to
              const uint32_t rt = rl[code];
              const uint32_t rf = rl[pc];
              if (HB)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(enw), "+v"(hdw) : : "memory");
              else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(enw) : : "memory");
              const uint32_t enx = (uint32_t)enw, eny = (uint32_t)(enw >> 32);
              const bool symhit = probe & u_sym(eny) == se & !grp;  // (a group record's second word is a slot number)
              const bool hit = symhit & !hdr;
              // a big state continues on this high symbol: its child's entry is the slot `first child of the group + set
              // bits below the symbol's`; the next trip probes it as the state "slot ^ symbol" and consumes the unit
              const bool redir = grp & probe & ((enx >> (code & 31u)) & 1u) != 0u;
              const uint32_t rE = (__builtin_popcount(enx & ~(~0u << (code & 31u))) + eny ^ code) | u_all_filter(BB);
              // a miss: the fail link is the root (or the unit matches nothing) -> the root's table answers in this trip;
              // else the unit is tried again in the fail state: root[the symbol that led here] (F1), or the state's header,
              // fetched by the next trip (falling into a state reports nothing: END is not carried)
              const bool viaroot = !symhit & !redir & (!u_nfr(E) | !good);
              // (HB: the header has come with the probe; else it is fetched by the next trip: "header pending")
              const uint32_t ft = u_f1(E) ? (rf & 0x7FFFFFFFu) : (HB ? hdw : (Bq | u_all_filter(BB) | 0x20000000u));
              uint32_t missE = viaroot ? rt : ft;
              missE = redir ? rE : missE;
              E = symhit ? enx : missE;
              const bool consumed = hit | viaroot;
              const bool end = consumed & u_end(E);
              const uint32_t c4 = hit ? u_c4(eny) : 1u;
              pc = consumed ? code : pc;
              const uint32_t adv = consumed ? Lb : 0u;
              if (CHARS) {  // characters that START in the lane's chunk (the rest of a chunk's bytes are continuation bytes)
                const uint32_t isl = (consumed & (int32_t)rel >= a_rel) ? (L >> 8) : 0u;
                lc += isl;
                lead_total += isl;
              }
              rel += adv;
              const bool park = consumed & n_later;  // the next unit waits for the next round
              lim2 = park ? rel : lim2;
              lim = park ? rel : lim;
              code = consumed ? n_code : code;  // (a trip that falls to the fail state tries the same unit again)
              L = consumed ? n_L : L;
              const int32_t last = (int32_t)rel - 1;  // row index of the unit's last byte
              // is_end? -> fetch later (ac.cr:183-185); this lane reports the end positions in [a, e)
              evc = (end & last >= a_rel & last < e_rel) ? c4 : 0u;
            }
            return evc;
          };
          const uint32_t evc = trip();
          const uint64_t evm = wballot(evc != 0u);
          if (evm) {
            // (a lane may send more than its region holds -- the call is repeated with larger regions then, see below --
            // but a wave never writes beyond its part: blocks past it are dropped)
            const bool ev = evc != 0u;
            const uint32_t my = wfill + __builtin_amdgcn_mbcnt_hi((uint32_t)(evm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)evm, 0u));
            // {END state | lane | hits it stands for, end offset in the document, hits of the chunk before it}
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
              const v3u r = {q[0], q[1], q[2]};
#ifndef AHA_KU_LAB_NOFLUSH  // (lab, timing only: what the flush's store costs the trips behind it)
              if (wout + 64u <= wcap) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = r;
#else
              if (wout + 64u == 0xFFFFFFF0u) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = r;
#endif
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
        if (all_left || !wany(rel < lim)) break;
      }
      if (need) pos = pb - 4 + rel;
    }
    {  // the rest of the wave's buffer
      const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + (wbo + __umul24((uint32_t)lane, 12u)));
      const v3u r = {q[0], q[1], q[2]};
      if ((uint32_t)lane < wfill && wout + (uint32_t)lane < wcap) *reinterpret_cast<v3u *>(wreg + (size_t)(wout + lane) * 3) = r;
    }
    if (live) {
      M.ev_cnt[chunk] = seq;
      if (seq > ev_stride) M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
      if (CHARS) M.lead_cnt[chunk] = lead_total;
      M.chunk_hits[chunk] = hits;  // (ku_regroup counts again where chains are longer than the record's field)
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

// The events of a group of 64 chunks, as the traversal's wave wrote them (in the order of its trips, lane tagged in
// bits 22..27 of the first word), go back into the chunks' own regions, each chunk's in order -- and get what
// k2d_count gives an event on the way: the key (or the offset of its flattened output chain) and the chain length
// instead of the END state's base; the hits of every chunk are summed up.
// One workgroup per group.  A block of 1024 records (four per thread, read coalesced) is sorted by chunk in LDS -- a
// stable counting sort: sixteen sub-batches of 64 records in source order; six ballots over the lane tag give every
// record its rank among the sub-batch's records of the same chunk and every lane the sub-batch's count for "its"
// chunk; exclusive sums over the sub-batches and the chunks give the place -- and leaves LDS as 64 runs, each
// appended to its chunk's region: consecutive threads store consecutive records (a store per record into 64 different
// regions costs a memory request per record, here as in the traversal).
constexpr int kRgThreads = 256, kRgPer = 4, kRgBlock = kRgThreads * kRgPer, kRgSubs = kRgBlock / 64;
__global__ __launch_bounds__(kRgThreads) void ku_regroup(DevAut A, V2Args M) {
  __shared__ uint2 s_sorted[kRgBlock];
  __shared__ uint8_t s_lane[kRgBlock];
  __shared__ uint32_t s_cnt[kRgSubs][64], s_part[kRgThreads / 64][64];
  __shared__ uint32_t s_start[64], s_tot[64], s_run[64], s_hits[64];
  if (M.cursor[1]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t n_groups = (M.n_chunks + 63) / 64;
  const uint32_t stride = M.ev_stride;
  const uint32_t bb = M.unit_bb, bmask = (1u << bb) - 1u;  // the image's base width (unit.hpp, BASE WIDTH)
  for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    __syncthreads();
    if (wv == 0) {
      s_run[lane] = 0;
      s_hits[lane] = 0;
    }
    const uint64_t c = g * 64 + lane;
    uint32_t total = c < M.n_chunks ? min(M.ev_cnt[c], stride) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
    const uint32_t *src = M.evg + g * 64 * stride * 3;
    uint2 *dst = M.evd + g * 64 * stride;
    // the next block's records and their end_info entries are requested a block ahead and waited for IN FRONT of this
    // block's stores: vmcnt counts loads and stores in one order, and a load waited for behind the stores waits for their
    // acknowledgement too (profiles/r06_expand_pipeline.txt)
    uint2 nxt[kRgPer];
    uint32_t xn[kRgPer];
#pragma unroll
    for (int q = 0; q < kRgPer; q++) {
      const uint32_t i = q * kRgThreads + threadIdx.x;
      nxt[q] = i < total ? make_uint2(src[(size_t)i * 3], src[(size_t)i * 3 + 1]) : make_uint2(0, 0);
    }
#pragma unroll
    for (int q = 0; q < kRgPer; q++) xn[q] = (uint32_t)(q * kRgThreads) + threadIdx.x < total ? A.end_info[nxt[q].x & bmask] : 0u;
    for (uint32_t i0 = 0; i0 < total; i0 += kRgBlock) {
      uint2 rec[kRgPer];
      uint32_t x[kRgPer], rank[kRgPer], l[kRgPer];
      bool live[kRgPer];
#pragma unroll
      for (int q = 0; q < kRgPer; q++) {
        const uint32_t i = i0 + q * kRgThreads + threadIdx.x;
        live[q] = i < total;
        rec[q] = nxt[q];
        // key id, or (flattened chains) the offset of its chain, | min(chain length, 255) << 24
        x[q] = xn[q];
        nxt[q] = i + kRgBlock < total ? make_uint2(src[(size_t)(i + kRgBlock) * 3], src[(size_t)(i + kRgBlock) * 3 + 1])
                                      : make_uint2(0, 0);
      }
#pragma unroll
      for (int q = 0; q < kRgPer; q++) {
        l[q] = (rec[q].x >> bb) & 63u;
        uint64_t same = __ballot(live[q]), mine = same;  // records of the sub-batch with this record's tag / with tag == lane
#pragma unroll
        for (int b = 0; b < 6; b++) {
          const uint64_t bal = __ballot(live[q] & ((l[q] >> b) & 1u) != 0u);
          same &= ((l[q] >> b) & 1u) ? bal : ~bal;
          mine &= ((lane >> b) & 1) ? bal : ~bal;
        }
        rank[q] = __popcll(same & ((1ull << lane) - 1ull));
        s_cnt[q * (kRgThreads / 64) + wv][lane] = __popcll(mine);  // sub-batch q * 4 + wv holds the records i0 + sb * 64 ..
      }
      __syncthreads();
      {  // exclusive sums down the sub-batches: four rows per thread, then across the four quarters
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < kRgSubs / (kRgThreads / 64); r++) {
          const uint32_t v = s_cnt[wv * (kRgSubs / (kRgThreads / 64)) + r][lane];
          s_cnt[wv * (kRgSubs / (kRgThreads / 64)) + r][lane] = acc;
          acc += v;
        }
        s_part[wv][lane] = acc;
      }
      __syncthreads();
      if (wv == 0) {
        uint32_t acc = 0;
#pragma unroll
        for (int w2 = 0; w2 < kRgThreads / 64; w2++) {
          const uint32_t v = s_part[w2][lane];
          s_part[w2][lane] = acc;
          acc += v;
        }
        s_tot[lane] = acc;
        s_start[lane] = wave_incl_scan(acc) - acc;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kRgPer; q++) {
        if (live[q]) {
          const uint32_t sb = q * (kRgThreads / 64) + wv;
          const uint32_t p = s_start[l[q]] + s_part[sb / (kRgSubs / (kRgThreads / 64))][l[q]] + s_cnt[sb][l[q]] + rank[q];
          uint32_t cnt = x[q] >> 24;
          if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[x[q] & 0xFFFFFFu].y : (x[q] & 0xFFFFFFu)];
          s_sorted[p] = make_uint2(x[q], rec[q].y);
          s_lane[p] = (uint8_t)l[q];
          if (cnt) atomicAdd(&s_hits[l[q]], cnt);
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kRgPer; q++)  // (the next block's records have arrived by now: their entries go out in front of the stores)
        xn[q] = i0 + kRgBlock + (uint32_t)(q * kRgThreads) + threadIdx.x < total ? A.end_info[nxt[q].x & bmask] : 0u;
#pragma unroll
      for (int q = 0; q < kRgPer; q++) asm volatile("" : "+v"(xn[q]));  // (... and are waited for here)
      const uint32_t nb = min(total - i0, (uint32_t)kRgBlock);
#pragma unroll
      for (int q = 0; q < kRgPer; q++) {
        const uint32_t p = q * kRgThreads + threadIdx.x;
        if (p < nb) {
          const uint32_t lp = s_lane[p];
          const uint32_t at = s_run[lp] + (p - s_start[lp]);
          if (at < stride) dst[(uint64_t)lp * stride + at] = s_sorted[p];
        }
      }
      __syncthreads();
      if (wv == 0) s_run[lane] += s_tot[lane];
    }
    __syncthreads();
    if (wv == 0 && c < M.n_chunks) M.chunk_hits[c] = s_hits[lane];
  }
}

// The expansion in one pass over the wave-ordered events (key sets whose output chains hold at most 15 keys -- kUFusedMaxChain --: the
// traversal then knows the hits of every chunk and of every event's predecessors in its chunk -- the record's third
// word -- so the bases are scanned before this kernel and every record knows where its hits go: nothing has to be put
// back into order).  Only the stores want order: a wave's 64 records belong to 64 chunks, and a 12-byte store per
// record is a memory request per record.  So a block of 1024 records stages its hits in LDS chunk by chunk -- the
// place is the chunk's start in the block (an exclusive sum over the chunks' hit counts, LDS atomics) + the hits of
// the chunk before the event - the hits of the chunk before the block -- and consecutive threads write consecutive hits.
// One lookup per event: uend[base of the END state] = its key, the key's length and the offset of its flattened
// output chain (the second and later hits of an event -- rare -- read the chain); the workgroups are persistent (two per
// CU) and keep the entries they have fetched in an LDS cache (below), so most lookups are a ds_read.
// CHARS: uend carries the key's length in characters, the record's second word the character count (see ku_traverse):
// hits are char offsets, Hit(char_of_byte[start], char_of_byte[end - 1] + 1) (matcher.cr:34-39).
// (shape, gather and stores measured apart: profiles/r04_expansion_lab.txt -- 1024 threads x 1 record beat 256 x 4 by 20 %)
constexpr int kXgThreads = 1024, kXgPer = 1, kXgBlock = kXgThreads * kXgPer, kXgStage = 1536;
constexpr int kXgCacheBits = 11;
template <bool CHARS>
__global__ __launch_bounds__(kXgThreads) void ku_expand_groups(const uint2 *uend, DevAut A, V2Args M) {
  __shared__ uint32_t s_hs[kXgStage], s_he[kXgStage], s_hk[kXgStage];  // staged hits: start, end, key
  __shared__ uint8_t s_hl[kXgStage];                                    // ... and the chunk (lane tag) of each
  __shared__ uint32_t s_tot[2][64], s_start[64], s_run[64], s_all;
  __shared__ uint64_t s_base[64];  // the chunk's place in the output
  __shared__ uint32_t s_adj[64];   // CHARS: characters between the start of the document that contains the chunk start and it
  // uend entries this workgroup has seen, direct-mapped by a hash of the base: {base + 1, first word}.  Hits pile up on few
  // END states (cfg 3: 28 M of 30 M events end one of 676 two-letter keys), but those states' bases -- and with them their
  // uend lines -- lie all over the image: 676 lines do not stay in a 32 KiB L1, 676 entries do in 16 KiB of LDS.  Only states
  // that stand for one hit of a key shorter than 256 are kept (the first word is all the expansion needs of them).
  __shared__ unsigned long long s_cache[1 << kXgCacheBits];
  if (M.cursor[1]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t n_groups = (M.n_chunks + 63) / 64;
  const uint32_t stride = M.ev_stride;
  const uint32_t bb = M.unit_bb, bmask = (1u << bb) - 1u;  // the image's base width (unit.hpp, BASE WIDTH)
  for (uint32_t i = threadIdx.x; i < (1u << kXgCacheBits); i += kXgThreads) s_cache[i] = 0ull;
  for (uint64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    __syncthreads();
    const uint64_t c = g * 64 + lane;
    if (wv == 0) {
      if (CHARS && c < M.n_chunks) s_adj[lane] = M.lead_cnt[c];  // (ku_chunk_adj has put the adjustment there)
      s_base[lane] = c < M.n_chunks ? M.hit_base[c] : 0ull;
      s_run[lane] = 0;
      s_tot[0][lane] = 0;
    }
    uint32_t total = c < M.n_chunks ? min(M.ev_cnt[c], stride) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
    const uint32_t *src = M.evg + g * 64 * stride * 3;
    typedef uint32_t v3u __attribute__((ext_vector_type(3)));
    v3u nxt[kXgPer];
#pragma unroll
    for (int q = 0; q < kXgPer; q++) {
      const uint32_t i = q * kXgThreads + threadIdx.x;
      // (non-temporal: the records are read once, they should not push uend lines out of L1 / L2)
      nxt[q] = i < total ? __builtin_nontemporal_load(reinterpret_cast<const v3u *>(src + (size_t)i * 3)) : v3u{0, 0, 0};
    }
    __syncthreads();
    uint32_t par = 0;
    for (uint32_t i0 = 0; i0 < total; i0 += kXgBlock, par ^= 1u) {
      v3u rec[kXgPer];
      uint2 ue[kXgPer];
      bool live[kXgPer], keep[kXgPer];
      uint32_t cs[kXgPer];
#pragma unroll
      for (int q = 0; q < kXgPer; q++) {
        const uint32_t i = i0 + q * kXgThreads + threadIdx.x;
        live[q] = i < total;
        rec[q] = nxt[q];
        nxt[q] = i + kXgBlock < total ? __builtin_nontemporal_load(reinterpret_cast<const v3u *>(src + (size_t)(i + kXgBlock) * 3)) : v3u{0, 0, 0};
        const uint32_t b = rec[q].x & bmask, n = u_rec_n(rec[q].x, rec[q].z, bb);
        cs[q] = (b * 0x9E3779B1u) >> (32 - kXgCacheBits);
        const unsigned long long ce = s_cache[cs[q]];
        const bool cached = live[q] & n == 1u & (uint32_t)ce == b + 1u;
        ue[q] = make_uint2((uint32_t)(ce >> 32), 0u);
        if (live[q] && !cached) ue[q] = uend[b];
        keep[q] = live[q] & !cached & n == 1u;
        if (live[q]) atomicAdd(&s_tot[par][(rec[q].x >> bb) & 63u], n);
      }
      __syncthreads();
      if (wv == 0) {
        const uint32_t t = s_tot[par][lane];
        const uint32_t incl = wave_incl_scan(t);
        s_start[lane] = incl - t;
        if (lane == 63) s_all = incl;
        s_tot[par ^ 1u][lane] = 0;  // the next block's
      }
      __syncthreads();
      const uint32_t T = s_all;
#pragma unroll
      for (int q = 0; q < kXgPer; q++)  // (between two barriers after every lookup of this block, before the next block's)
        if (keep[q] && (ue[q].y >> 24) == 0u)
          atomicExch(&s_cache[cs[q]], (unsigned long long)ue[q].x << 32 | ((rec[q].x & bmask) + 1u));
      for (uint32_t w0 = 0; w0 < T; w0 += kXgStage) {  // one window unless events stand for many hits
#pragma unroll
        for (int q = 0; q < kXgPer; q++) {
          if (live[q]) {
            const uint32_t l = (rec[q].x >> bb) & 63u, n = u_rec_n(rec[q].x, rec[q].z, bb);
            const uint32_t pos = s_start[l] + (u_rec_before(rec[q].z) - s_run[l]);
            const uint32_t end = CHARS ? (rec[q].y >> 1) + ((rec[q].y & 1u) ? 0u : s_adj[l]) : rec[q].y, co = ue[q].y & 0xFFFFFFu;
            // Hit(idx - len + 1, idx + 1, value) ac.cr:271-273: the state's own key, then its output chain (ac.cr:265-278)
            uint32_t len = (ue[q].x >> 24) | (ue[q].y >> 24) << 8, key = ue[q].x & 0xFFFFFFu;
            for (uint32_t k = 0;;) {
              const uint32_t j = pos + k - w0;
              if (j < (uint32_t)kXgStage) {
                s_hs[j] = end - len;
                s_he[j] = end;
                s_hk[j] = key;
                s_hl[j] = (uint8_t)l;
              }
              if (++k >= n) break;
              const uint2 ce = (CHARS ? A.chain_chars : A.chain)[co + k];
              len = ce.x;
              key = ce.y;
            }
          }
        }
        __syncthreads();
        const uint32_t nh = min(T - w0, (uint32_t)kXgStage);
        for (uint32_t j = threadIdx.x; j < nh; j += kXgThreads) {
          const uint32_t l = s_hl[j];
          const uint64_t idx = s_base[l] + s_run[l] + (w0 + j - s_start[l]);
          if (idx < M.cap) {
            aha_hit hit;
            hit.start = (int32_t)s_hs[j];
            hit.end = (int32_t)s_he[j];
            hit.value = (int32_t)s_hk[j];
            M.out[idx] = hit;
          }
        }
        __syncthreads();
      }
      if (wv == 0) s_run[lane] += s_tot[par][lane];
    }
  }
}

// CHARS: the characters between the start of the document that contains a chunk's start and the chunk, for every chunk: three
// dependent loads that a persistent expansion workgroup would wait for at the head of every group.  Written over lead_cnt
// (the scan has consumed it).
__global__ __launch_bounds__(256) void ku_chunk_adj(V2Args M) {
  if (M.cursor[1]) return;
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= M.n_chunks) return;
  const uint32_t d0 = M.chunk_doc0[c];
  const uint64_t dchunk = M.doc_off[d0] / M.S;
  M.lead_cnt[c] = (uint32_t)(M.lead_base[c] - (M.lead_base[dchunk] + M.doc_lead_rank[d0]));
}

// doc_hit_off[d] = hits before the document's first event: the chunk's hit base + the hits of the chunk before the
// document start, which the traversal noted at the boundary
__global__ __launch_bounds__(256) void ku_doc_offsets(V2Args M) {
  // (the call's last kernel: every word the host reads is final -- this kernel changes none of them)
  if (M.publish && blockIdx.x == 0 && threadIdx.x < 5) M.publish[threadIdx.x] = M.cursor[threadIdx.x];
  if (M.clear_next && blockIdx.x == 0 && threadIdx.x < 16) M.clear_next[threadIdx.x] = 0ull;  // (the next call's counters)
  if (M.cursor[1] || !M.doc_hit_off) return;
  const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (d > M.n_docs) return;
  const uint64_t q = M.doc_off[d];
  M.doc_hit_off[d] = q < M.n_bytes ? M.hit_base[q / M.S] + M.doc_hit_rank[d] : M.totals[0];
}

}  // namespace

size_t unit_lds_bytes(uint32_t n_syms) { return u_lds(n_syms); }

int unit_prepare(uint32_t n_syms) {
  const int lds = (int)unit_lds_bytes(n_syms);
  int e = 0;
#define AHA_PREP_KU(C, B, H) \
  if (!e) e = (int)hipFuncSetAttribute((const void *)ku_traverse<C, B, H>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  AHA_PREP_KU(false, 22, false) AHA_PREP_KU(true, 22, false) AHA_PREP_KU(false, 23, false) AHA_PREP_KU(true, 23, false)
  AHA_PREP_KU(false, 22, true) AHA_PREP_KU(true, 22, true) AHA_PREP_KU(false, 23, true) AHA_PREP_KU(true, 23, true)
#undef AHA_PREP_KU
  return e;
}

void unit_launch_traverse(const UnitDev &U, const V2Args &M, uint32_t grid, void *stream) {
  const size_t lds = unit_lds_bytes(U.n_syms);
#define AHA_LAUNCH_KU(C, B, H) hipLaunchKernelGGL((ku_traverse<C, B, H>), dim3(grid), dim3(kV2Threads), lds, (hipStream_t)stream, U, M)
#define AHA_LAUNCH_KU2(C, B) \
  do {                       \
    if (U.hdr_beside)        \
      AHA_LAUNCH_KU(C, B, true);  \
    else                     \
      AHA_LAUNCH_KU(C, B, false); \
  } while (0)
  if (U.base_bits == 23) {
    if (M.chars) AHA_LAUNCH_KU2(true, 23); else AHA_LAUNCH_KU2(false, 23);
  } else {
    if (M.chars) AHA_LAUNCH_KU2(true, 22); else AHA_LAUNCH_KU2(false, 22);
  }
#undef AHA_LAUNCH_KU2
#undef AHA_LAUNCH_KU
}

void unit_launch_regroup(const DevAut &A, const V2Args &M, void *stream) {
  const uint64_t n_groups = (M.n_chunks + 63) / 64;
  hipLaunchKernelGGL(ku_regroup, dim3((uint32_t)std::min<uint64_t>(n_groups, 1u << 16)), dim3(kRgThreads), 0,
                     (hipStream_t)stream, A, M);
}

void unit_launch_expand(const uint2 *uend, const DevAut &A, const V2Args &M, uint32_t workgroups, void *stream) {
  const uint64_t n_groups = (M.n_chunks + 63) / 64;
  // persistent workgroups (two of 1024 threads fit a CU): each keeps its LDS cache of uend entries over its groups
  const dim3 grid((uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_groups, workgroups)));
  if (M.chars) {
    hipLaunchKernelGGL(ku_chunk_adj, dim3((uint32_t)((M.n_chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M);
    hipLaunchKernelGGL(ku_expand_groups<true>, grid, dim3(kXgThreads), 0, (hipStream_t)stream, uend, A, M);
  } else {
    hipLaunchKernelGGL(ku_expand_groups<false>, grid, dim3(kXgThreads), 0, (hipStream_t)stream, uend, A, M);
  }
  if (M.doc_hit_off)
    hipLaunchKernelGGL(ku_doc_offsets, dim3((uint32_t)((M.n_docs + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M);
}

}  // namespace aha
