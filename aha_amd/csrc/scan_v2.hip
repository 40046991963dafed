// scan_v2.hip -- single-traversal match engine for gfx950 (MI355X).
//
// Replaces the same reference path as kernels.hip (src/aha/ac.cr:176-192
// match_, :265-278 fetch, src/aha/cedar.cr:441-447 child, matcher.cr:34-39)
// but walks the corpus ONCE:
//
//   k2_traverse   one persistent 1024-thread workgroup per CU.  The first
//                 lds_slots slots of the BFS-ordered double array (the hot,
//                 shallow states) are copied into LDS once per workgroup;
//                 deeper states are probed in HBM/L2.  Each lane owns one
//                 contiguous super-chunk of S bytes and streams it through a
//                 wave-private LDS window (32 B per lane per round, rows padded
//                 to an odd dword stride).  Lanes advance independently: one
//                 state lookup per loop trip (goto probe or fail header) beside
//                 an always-LDS probe of the root row.  A position whose state
//                 ends a key is NOT expanded here: a lane keeps one pending
//                 16-byte "event" in registers and the wave flushes pending
//                 events with ballot + mbcnt into a slab it reserved with one
//                 atomic -- no key-table loads, no ordering work in the loop.
//   k2_sort       scatters the records into final (position) order using the
//                 exclusive scan of per-chunk event counts, resolves the key
//                 and its output-chain length.
//   k2_expand     walks each event's output chain (ac.cr:265-278) and writes
//                 the Hit triples at their final index; k2_doc_offsets writes
//                 the per-document hit offsets.
//
// Exactness: a lane starts (Lmax-1) bytes before its chunk (clamped to the
// document start) at root, so its state is the sequential automaton's state
// for every position it reports (see kernels.hip header).
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"

namespace aha {

namespace {

constexpr uint32_t kLastFlag = 0x8000u;

template <bool COMPACT>
struct Slot;
template <>
struct Slot<true> {
  using type = uint32_t;
  static __device__ __forceinline__ bool match(type e, uint32_t b) { return (e & 0xFFu) == b; }
  static __device__ __forceinline__ uint32_t base(type e) { return (e >> C_BASE_SHIFT) & C_BASE_MASK; }
  static __device__ __forceinline__ bool end(type e) { return (e & C_END) != 0; }
  static __device__ __forceinline__ uint32_t failroot(type e) { return e & C_FAILROOT; }
  // payload of an event: the new state's base (key resolved later through end_key)
  static __device__ __forceinline__ uint32_t payload(type e) { return base(e); }
};
template <>
struct Slot<false> {
  using type = uint2;
  static __device__ __forceinline__ bool match(type e, uint32_t b) { return (e.y & 0xFFu) == b; }
  static __device__ __forceinline__ uint32_t base(type e) { return e.x & W_BASE_MASK; }
  static __device__ __forceinline__ bool end(type e) { return (e.x & W_END) != 0; }
  static __device__ __forceinline__ uint32_t failroot(type e) { return e.x & W_FAILROOT; }
  static __device__ __forceinline__ uint32_t payload(type e) { return e.y >> 8; }  // key id
};

// ------------------------------------------------------------------ traverse
// The hot loop is VALU-issue bound (measured: its time does not move with the
// number of HBM probes nor with twice the waves), so it is written for few
// instructions per trip: 32-bit positions relative to the staged piece, one
// table lookup per trip (a goto probe OR the fail header of the previous
// miss), conflict-free padded LDS input rows, 64-bit arithmetic only in the
// rare paths (document boundaries, events).
constexpr int kInStride = kV2Piece + 4;           // bytes per lane in the LDS input window (odd dword stride)
constexpr int kWaveIn2 = 64 * kInStride;

// ALL_LDS: the whole image fits the LDS budget (cfg 2): the same trip, every lookup a ds_read (no far path).  (Until round 4
// such automata kept a fail header for every state and ran a trip of their own that read the header beside the probe:
// 1.55 trips per byte on cfg 2; with the shadow fail links of the partial-prefix trip 1.02, 6 % less time even through the
// flat path -- profiles/r04_bench_cfg2_*.)
#ifndef AHA_V2_WAVES_PER_SIMD
#define AHA_V2_WAVES_PER_SIMD 4  // one 1024-thread workgroup per CU; 8 = two (lab: does the occupancy pay?)
#endif
template <bool COMPACT, bool CHARS, bool ALL_LDS>
__global__ __launch_bounds__(kV2Threads, AHA_V2_WAVES_PER_SIMD) void k2_traverse(DevAut A, V2Args M) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // device-resident doc offsets are validated by a small kernel in front of this one (k_check_docs, capi.cpp match_v2): a bad
  // verdict stands in cursor[1] and nothing is indexed with them
  if (M.cursor[1] >= 16ull) return;
  using S_ = Slot<COMPACT>;
  using slot_t = typename S_::type;
  slot_t *lt = reinterpret_cast<slot_t *>(smem);
  const slot_t *gt = reinterpret_cast<const slot_t *>(A.slots);
  const uint32_t T = M.lds_slots;
  uint8_t *in_base = smem + (size_t)T * sizeof(slot_t);
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(gt);
    uint4 *dst = reinterpret_cast<uint4 *>(lt);
    const uint32_t nvec = (uint32_t)((size_t)T * sizeof(slot_t) / 16);
    for (uint32_t i = threadIdx.x; i < nvec; i += kV2Threads) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *inl = in_base + wave * kWaveIn2 + lane * kInStride;
  const uint32_t root = A.root;
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int64_t S = (int64_t)M.S;
  const int rounds = (int)(M.S / kV2Piece);
  const int warm = A.max_len > 1 ? (int)A.max_len - 1 : 0;
  const int R = (warm + kV2Piece - 1) / kV2Piece;  // warm-up rounds before the chunk

  // wave-private slab of event records (values are wave-uniform)
  uint64_t slab_pos = 0;
  uint32_t slab_left = 0, slab_used_n = 0;
  uint64_t slab_id = ~0ull;

  // lazy events: a lane keeps ONE pending record in registers; the wave
  // flushes all pending records (ballot + mbcnt compaction into its slab)
  // only when a lane that already holds one produces another.
  bool pev = false;
  uint32_t p_x = 0, p_y = 0, p_z = 0, p_w = 0, p_aux = 0;
  auto flush_events = [&]() {
    const unsigned long long mask = __ballot(pev);
    if (!mask) return;
    const uint32_t n = (uint32_t)__popcll(mask);
    if (slab_left < n) {
      if (slab_id != ~0ull && lane == 0) M.slab_used[slab_id] = slab_used_n;
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(M.cursor, (unsigned long long)kV2Slab);
      base = __shfl(base, 0, 64);
      slab_pos = base;
      slab_left = kV2Slab;
      slab_used_n = 0;
      slab_id = base / kV2Slab;
      if (base + kV2Slab > M.ev_cap) {  // temp exhausted: the host falls back
        if (lane == 0) M.cursor[1] = 1ull;
        slab_id = ~0ull;
      }
    }
    if (pev && slab_id != ~0ull) {
      const uint32_t my = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
      M.ev[slab_pos + my] = make_uint4(p_x, p_y, p_z, p_w);
      if (CHARS) M.ev_aux[slab_pos + my] = p_aux;
    }
    pev = false;
    slab_pos += n;
    slab_left -= n;
    slab_used_n += n;
  };

  const uint64_t n_tiles = (M.n_chunks + kV2Threads - 1) / kV2Threads;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t chunk = tile * kV2Threads + threadIdx.x;
    p_x = (uint32_t)chunk;
    uint2 *evreg = M.direct ? M.evd + chunk * M.ev_stride : nullptr;  // this chunk's event region (plain mode)
    const uint32_t ev_stride = M.ev_stride;
    const bool live = chunk < M.n_chunks;
    const int64_t a = (int64_t)chunk * S;
    const int64_t e = live ? min(a + S, N) : a;
    uint64_t dn = 0;
    int64_t nb = INT64_MAX, doc_start = a, pos = e;
    uint32_t B = root, fr = 0, seq = 0;
    // shadow fail (A.s2_lo < A.s2_hi): r1 = root-row entry of the last consumed byte (0 = none), s2 = entry of the
    // depth<=2 state of the last two consumed bytes.  A miss in a state whose base lies in [s2_lo, s2_hi) continues
    // in s2's state in the same trip: no header trip, no far header load.
    slot_t r1 = slot_t{}, s2 = slot_t{};
    uint32_t hm = 0xFFu;  // partial-prefix trip: 0xFF = probe trip, 0 = header trip (label 0)
    uint4 half[kV2Piece / 16];    // second half of the current input line (see the staging code)
#pragma unroll
    for (int k = 0; k < kV2Piece / 16; k++) half[k] = make_uint4(0, 0, 0, 0);
    int64_t half_pb = INT64_MIN;  // the piece `half` holds
    uint32_t lc = 0, lc_exact = 0, lead_total = 0;
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

    for (int r = -R; r < rounds; r++) {
      const int64_t pb = a + (int64_t)r * kV2Piece;
      const int64_t pend = min(pb + kV2Piece, e);
      const bool need = live && pos < pend;
      if (!__any(need)) continue;
      uint32_t rel = kV2Piece, lim = 0;  // inactive: rel >= lim
      if (need) {
        // A piece is half a 64-byte line.  The round that starts a line also loads its second half into
        // registers for the next round, so both halves are requested while the line is in flight and every
        // input line is fetched once (it would not survive a whole round in L2 beside the table lookups).
        const bool line_start = (pb & 63) == 0;
        uint4 v[kV2Piece / 16];
        if (!line_start && half_pb == pb) {
#pragma unroll
          for (int k = 0; k < kV2Piece / 16; k++) v[k] = half[k];
        } else {
#pragma unroll
          for (int k = 0; k < kV2Piece / 16; k++) {
            const int64_t g = pb + k * 16;
            v[k] = make_uint4(0, 0, 0, 0);
            if (g >= 0 && g + 16 <= N) {
              v[k] = *reinterpret_cast<const uint4 *>(M.text + g);
            } else if (g >= 0 && g < N) {
              uint32_t w[4] = {0, 0, 0, 0};
              for (int j = 0; j < 16 && g + j < N; j++) w[j >> 2] |= (uint32_t)M.text[g + j] << ((j & 3) * 8);
              v[k] = make_uint4(w[0], w[1], w[2], w[3]);
            }
          }
        }
        half_pb = INT64_MIN;
        if (line_start && pb >= 0 && pb + 2 * kV2Piece <= N && pb + kV2Piece < e) {
#pragma unroll
          for (int k = 0; k < kV2Piece / 16; k++)
            half[k] = *reinterpret_cast<const uint4 *>(M.text + pb + kV2Piece + k * 16);
          half_pb = pb + kV2Piece;
        }
#pragma unroll
        for (int k = 0; k < kV2Piece / 16; k++) {
          uint32_t *dst = reinterpret_cast<uint32_t *>(inl + k * 16);
          dst[0] = v[k].x;
          dst[1] = v[k].y;
          dst[2] = v[k].z;
          dst[3] = v[k].w;
        }
        rel = (uint32_t)(pos - pb);
        lim = (uint32_t)(pend - pb);
      }
      // document boundary inside this piece, as a piece-relative offset
      uint32_t nb_rel = (nb >= pb && nb < pb + kV2Piece) ? (uint32_t)(nb - pb) : ~0u;
      int32_t docrel = (int32_t)(pb - doc_start);  // end offset in the document = docrel + rel + 1
      const bool emit_ok = r >= 0;                 // warm-up rounds report nothing
      const uint32_t chunk_rel0 = (uint32_t)(r * kV2Piece);

      // ---- hot loop: one lookup per trip, lanes advance independently --------
      // Written as predicated straight-line code (selects, not branches): the
      // compiler's nested divergent branches cost ~64 SALU + 63 VALU per trip
      // (rocprofv3 SQ_INSTS_*), which made the loop issue-bound.
      for (;;) {  // outer: resolve document boundaries, then run the hot loop up to the next one
        const bool bnd = rel < lim && rel == nb_rel;
        if (__any(bnd)) {  // rare: a document starts here (ac.cr:177: state is per sequence)
          if (bnd) {
            const int64_t here = pb + rel;
            do {
              M.doc_ev_rank[dn] = seq;
              if (CHARS) M.doc_lead_rank[dn] = lead_total;
              dn++;
              nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            } while (nb == here);
            asm volatile("" : "+v"(nb));  // retire the load inside this block
            nb_rel = (nb < pb + kV2Piece) ? (uint32_t)(nb - pb) : ~0u;
            B = root;
            fr = 0;
            r1 = slot_t{};
            s2 = slot_t{};
            hm = 0xFFu;
            doc_start = here;
            docrel = -(int32_t)rel;
            lc = 0;
            lc_exact = 1;
          }
        }
        const uint32_t lim2 = min(lim, nb_rel);  // lanes park at the next boundary: no boundary test per trip
        uint32_t bcur = 0;
        bcur = inl[min(rel, (uint32_t)kV2Piece)];  // first byte of this run of trips
      for (;;) {
        const bool act = rel < lim2;
        if (!__any(act)) break;
        bool ev = false;
        uint32_t en_keep = 0;
        if (act) {
          // Partial prefix: the trip with few mask operations.  A header trip is a probe with label 0 (the
          // header slot is slot[B ^ 0] and carries label 0), so one compare serves goto and header alike;
          // `hm` is 0xFF in a probe trip and 0 in a header trip.
          // the byte comes from a register: the next one was loaded during the previous trip (nearly every trip
          // consumes), which takes the LDS round trip of the byte out of the dependent chain
          const uint32_t b = bcur;
          const uint32_t bnext = inl[rel + 1];                    // rows are padded: rel + 1 <= piece + 3
          const uint32_t c = b & hm;
          const uint32_t idx = B ^ c;
          const slot_t e0 = lt[root ^ b];                         // root row: always LDS resident
          const slot_t e2 = lt[S_::base(r1) ^ b];                 // depth-1 rows: always LDS resident
          const slot_t sx = B < A.s2_lo ? r1 : s2;                // shadow fail target of B (if B has one)
          const uint32_t i3 = S_::base(sx) ^ b;
          const bool near3 = i3 < T;
          const slot_t e3 = lt[near3 ? i3 : 0u];                  // sx's row (depth <= 2: mostly LDS resident)
          slot_t en;
          if constexpr (ALL_LDS) {
            en = lt[idx];                                         // the whole image is in LDS: no far path
          } else {
            // (one flat load over both apertures; ds_read + exec-masked global_load, sc1 / nt / 8-byte far loads were
            // measured in round 3: profiles/r03_trip_anatomy.txt)
            if (idx < T)
              en = lt[idx];
            else
              en = gt[idx];
          }
          const bool nz = b != 0;
          const bool probe = hm != 0;
          const bool bzp = !nz && probe;                          // NUL contract: state := root, byte consumed
          const bool atroot = B == root || fr != 0 || bzp;        // fails[nid] = root: probe the root row now
          const bool mr = nz && S_::match(e0, b);                 // b has a depth-1 state
          // fails[nid] of a state in [s1_lo, s2_hi) follows from the last bytes: sx = r1 (depth-2 state) or s2.
          // Its row, the row of ITS fail target (r1: e2) and the root row (e0) are probed in this same trip, so
          // the whole rest of the fail chain is resolved here and the byte is consumed (1.19 -> 1.02 trips per
          // byte); only when sx's row lies beyond the LDS prefix the walk continues in sx without consuming.
          const bool m3 = nz && S_::match(e3, b);
          const bool m2 = nz && S_::match(e2, b);
          const slot_t chain = m3 ? e3 : (m2 ? e2 : (mr ? e0 : slot_t{}));   // first goto along sx -> r1 -> root
          // the byte was consumed: new depth<=2 state of the last two bytes, new depth-1 entry
          const slot_t r1n = mr ? e0 : slot_t{};
          const slot_t s2n = m2 ? e2 : r1n;
          const bool shadow = !atroot && (B - A.s1_lo) < (A.s2_hi - A.s1_lo);
          const bool t = S_::match(en, c) && !bzp;                // goto (cedar.cr:441-447), or the header itself
          const bool sgo = !t && shadow;
          const bool sres = sgo && near3;
          const slot_t ex = t ? en : (sgo ? (near3 ? chain : sx) : (mr ? e0 : slot_t{}));
          const bool land = t || atroot || sgo;                   // else: the next trip loads fails[nid] (ac.cr:189)
          const bool consumed = (t && probe) || (!t && atroot) || sres;  // at root a miss consumes (ac.cr:188)
          B = land ? S_::base(ex) : B;
          fr = land ? S_::failroot(ex) : fr;
          hm = land ? 0xFFu : 0u;
          ev = consumed && S_::end(ex) && emit_ok;                // is_end? -> fetch later (ac.cr:183-185)
          s2 = consumed ? s2n : s2;
          r1 = consumed ? r1n : r1;
          if (CHARS) {
            const uint32_t isl = (consumed && emit_ok && (b & 0xC0u) != 0x80u) ? 1u : 0u;
            lc += isl;
            lead_total += isl;
          }
          rel += consumed ? 1u : 0u;
          bcur = consumed ? bnext : bcur;
          en_keep = S_::payload(ex);
        }
        if (__any(ev)) {
          if (M.direct) {
            // plain mode: the chunk's events go, in order, to its own region -- no compaction, no sort
            if (ev) {
              if (seq < ev_stride) {
                typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                const v2u rec = {en_keep, CHARS ? ((lc << 1) | lc_exact) : (uint32_t)(docrel + (int32_t)rel)};
                // a plain 8-byte store: hipcc ignores __builtin_nontemporal_store on gfx950, and a real non-temporal
                // store (buffer store, aux = nt) was worth under 1 % (profiles/r03_trip_anatomy.txt)
                *reinterpret_cast<v2u *>(evreg + seq) = rec;
              } else {
                M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
              }
              seq++;
            }
          } else {
            if (__any(ev && pev)) flush_events();
            if (ev) {
              // rel was already advanced: the hit ended at rel-1
              const uint32_t last = (rel == nb_rel) ? kLastFlag : 0u;
              pev = true;
              p_y = (seq << 16) | last | (chunk_rel0 + rel - 1u);
              p_z = (uint32_t)(docrel + (int32_t)rel);
              p_w = en_keep;
              p_aux = (lc << 1) | lc_exact;
              seq++;
            }
          }
        }
      }
        if (!__any(rel < lim)) break;
      }
      if (need) pos = pb + rel;
    }
    if (__any(pev)) flush_events();
    if (live) {
      M.ev_cnt[chunk] = seq;
      if (CHARS) M.lead_cnt[chunk] = lead_total;
      if (e == N) {  // documents that start at N (empty tail documents, and d = D)
        while (dn <= D) {
          M.doc_ev_rank[dn] = seq;
          if (CHARS) M.doc_lead_rank[dn] = lead_total;
          dn++;
        }
      }
    }
  }
  if (slab_id != ~0ull && lane == 0) M.slab_used[slab_id] = slab_used_n;
}

// ---------------------------------------------------------------- scans
// `abort_flag` (nullable): when set (event temp exhausted) the device-side
// count in *n_ptr may exceed the scratch sizes, so the pass must not run.
__global__ __launch_bounds__(256) void k2_blocksum(const uint32_t *in, const uint64_t *n_ptr, uint64_t n_fixed,
                                                    uint64_t *blk, const unsigned long long *abort_flag) {
  __shared__ uint64_t sm[4];
  if (abort_flag && *abort_flag) return;
  const uint64_t n = n_ptr ? *n_ptr : n_fixed;
  const uint64_t nblk = (n + 255) / 256;
  for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint64_t i = b * 256 + threadIdx.x;
    uint64_t tot;
    block_excl_scan<uint64_t>(i < n ? in[i] : 0u, sm, &tot);
    if (threadIdx.x == 0) blk[b] = tot;
  }
}

// exclusive scan of blk[0..ceil(n/256)) in place by one workgroup; *total = sum
__global__ __launch_bounds__(1024) void k2_scan_blocks(uint64_t *arr, const uint64_t *n_ptr, uint64_t n_fixed,
                                                       uint64_t *total, const unsigned long long *abort_flag) {
  __shared__ uint64_t sm[16];
  if (abort_flag && *abort_flag) return;
  const uint64_t n_items = n_ptr ? *n_ptr : n_fixed;
  const uint64_t n = (n_items + 255) / 256;
  uint64_t carry = 0;
  // 4 consecutive items per thread: 4096 per tile
  for (uint64_t t0 = 0; t0 < n; t0 += 4096) {
    const uint64_t i0 = t0 + (uint64_t)threadIdx.x * 4;
    uint64_t v[4], s4 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      v[j] = (i0 + j < n) ? arr[i0 + j] : 0;
      s4 += v[j];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t inc = wave_incl_scan(s4);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
    for (int k = 0; k < 16; k++) {
      const uint64_t s = sm[k];
      if (k < w) base += s;
      tot += s;
    }
    uint64_t run = carry + base + inc - s4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i0 + j < n) arr[i0 + j] = run;
      run += v[j];
    }
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void k2_scan_apply(const uint32_t *in, uint64_t n, const uint64_t *blk,
                                                      uint64_t *out) {
  __shared__ uint64_t sm[4];
  const uint64_t nblk = (n + 255) / 256;
  for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint64_t i = b * 256 + threadIdx.x;
    const uint64_t ex = block_excl_scan<uint64_t>(i < n ? in[i] : 0u, sm, nullptr);
    if (i < n) out[i] = blk[b] + ex;
  }
}

// The three passes above in one launch for up to 16384 items (a batch of 64 MiB in chunks of 4 KiB -- where three launches
// and their gaps are a tenth of the call): one workgroup, 7 us against ~14 + two gaps; thread t loads the 16 consecutive
// items [16 t, 16 t + 16) (four 16-byte loads in flight), sums them, the block scans the sums, the thread numbers its items.
// (More rounds work -- 65536 items: 24 us -- but the three launches are faster there.)
constexpr uint64_t kScanSmall = 16384;
__global__ __launch_bounds__(1024) void k2_scan_small(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out,
                                                      uint64_t *total, const unsigned long long *abort_flag) {
  __shared__ uint64_t sm[16];
  if (abort_flag && *abort_flag) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t carry = 0;
  for (uint64_t r0 = 0; r0 < n; r0 += 16384) {
    const uint64_t i0 = r0 + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 x = reinterpret_cast<const uint4 *>(in + i0)[q];
        v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = i0 + j < n ? in[i0 + j] : 0u;
    }
    uint64_t mine = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) mine += v[j];
    const uint64_t inc = wave_incl_scan(mine);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t x = sm[j];
      if (j < w) base += x;
      tot += x;
    }
    uint64_t run = carry + base + inc - mine;
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint64_t a = run, b = run + v[2 * q];
        run = b + v[2 * q + 1];
        reinterpret_cast<ulonglong2 *>(out + i0)[q] = make_ulonglong2(a, b);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
      }
    }
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

// exclusive scan of in[0..n) into out, the sum into *total
static void launch_scan(const uint32_t *in, uint64_t n, uint64_t *blk, uint64_t *out, uint64_t *total,
                        const unsigned long long *abortf, hipStream_t s) {
  if (n <= kScanSmall) {
    hipLaunchKernelGGL(k2_scan_small, dim3(1), dim3(1024), 0, s, in, n, out, total, abortf);
    return;
  }
  const uint32_t g = (uint32_t)std::min<uint64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k2_blocksum, dim3(g), dim3(256), 0, s, in, (const uint64_t *)nullptr, n, blk, abortf);
  hipLaunchKernelGGL(k2_scan_blocks, dim3(1), dim3(1024), 0, s, blk, (const uint64_t *)nullptr, n, total, abortf);
  hipLaunchKernelGGL(k2_scan_apply, dim3(g), dim3(256), 0, s, in, n, blk, out);
}

// -------------------------------------------------------- event -> hit chain
// Walks the output chain of `key` for an event at absolute position abs_pos
// (end offset end_b inside its document) and calls f(k, len) for every hit the
// reference yields, in its order (fetch, ac.cr:265-278; with the separator
// tests of match(seq, sep), ac.cr:324-336).
template <class F>
__device__ __forceinline__ void for_each_hit(const DevAut &A, const V2Args &M, uint32_t key, uint64_t abs_pos,
                                             uint32_t end_b, bool last, F &&f) {
  if (M.sep && !last && sep_blocked_bits(M.sep_block, M.text[abs_pos + 1])) return;
  int32_t k = (int32_t)key;
  do {
    const uint2 ln = A.key_ln[k];
    const bool blocked = M.sep && end_b > ln.x && sep_blocked_bits(M.sep_block, M.text[abs_pos - ln.x]);
    if (!blocked) f(k, ln.x);
    k = (int32_t)ln.y;
  } while (k >= 0);
}

// PLAIN (no separator filter, byte offsets): an event needs only {key, end_b}
// downstream, and its chain length is key_cnt[key], so the sorted stream is
// 8 bytes per event instead of 20.
template <bool COMPACT, bool PLAIN>
__global__ __launch_bounds__(256) void k2_sort(DevAut A, V2Args M) {
  if (M.cursor[1]) return;  // temp overflow: results are discarded by the host
  const uint64_t n = min<uint64_t>(M.cursor[0], M.ev_cap);
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    if ((uint32_t)(i % kV2Slab) >= M.slab_used[i / kV2Slab]) continue;
    const uint4 rec = M.ev[i];
    const uint64_t p = M.ev_base[rec.x] + (rec.y >> 16);
    const uint32_t key = COMPACT ? (uint32_t)A.end_key[rec.w] : rec.w;
    if (PLAIN) {
      reinterpret_cast<uint2 *>(M.sorted_ev)[p] = make_uint2(key, rec.z);
      continue;
    }
    uint32_t cnt;
    if (!M.sep) {
      cnt = A.key_cnt[key];
    } else {
      cnt = 0;
      const uint64_t abs_pos = (uint64_t)rec.x * M.S + (rec.y & 0x7FFFu);
      for_each_hit(A, M, key, abs_pos, rec.z, (rec.y & kLastFlag) != 0, [&](int32_t, uint32_t) { cnt++; });
    }
    M.sorted_ev[p] = make_uint4(key, rec.z, rec.x, rec.y);
    M.sorted_cnt[p] = cnt;
    if (M.chars) M.sorted_aux[p] = M.ev_aux[i];
  }
}

// PLAIN: per-256 block sums of key_cnt[key] over the sorted 8-byte stream
__global__ __launch_bounds__(256) void k2_blocksum_keys(DevAut A, V2Args M) {
  __shared__ uint64_t sm[4];
  if (M.cursor[1]) return;
  const uint64_t n = M.totals[2];
  const uint64_t nblk = (n + 255) / 256;
  const uint2 *sv = reinterpret_cast<const uint2 *>(M.sorted_ev);
  for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint64_t i = b * 256 + threadIdx.x;
    uint64_t tot;
    block_excl_scan<uint64_t>(i < n ? A.key_cnt[sv[i].x] : 0u, sm, &tot);
    if (threadIdx.x == 0) M.blk_a[b] = tot;
  }
}

// Hits of a block's 256 events are assembled in LDS (each thread walks its own
// chain) and then streamed out with fully coalesced dword stores; hit-dense
// inputs (cfg 5: ~10 hits per event) are write-bandwidth bound here.
constexpr uint32_t kHitStage = 4096;  // hits staged per block (48 KiB of LDS)

__global__ __launch_bounds__(256) void k2_expand_plain(DevAut A, V2Args M) {
  __shared__ uint64_t sm[4];
  __shared__ uint32_t hbuf[kHitStage * 3];
  if (M.cursor[1]) return;
  const uint64_t n = M.totals[2];
  const uint64_t nblk = (n + 255) / 256;
  const uint2 *sv = reinterpret_cast<const uint2 *>(M.sorted_ev);
  for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint64_t p = b * 256 + threadIdx.x;
    const bool live = p < n;
    uint2 rec = make_uint2(0, 0);
    uint32_t cnt = 0;
    if (live) {
      rec = sv[p];
      cnt = A.key_cnt[rec.x];
    }
    uint64_t tot64;
    const uint64_t off = block_excl_scan<uint64_t>(cnt, sm, &tot64);
    const uint64_t base = M.blk_a[b];
    if (tot64 <= kHitStage) {
      if (live) {
        uint32_t w = (uint32_t)off * 3;
        int32_t k = (int32_t)rec.x;
        do {  // fetch (ac.cr:265-278): own key, then the output chain
          const uint2 ln = A.key_ln[k];
          hbuf[w] = rec.y - ln.x;  // Hit(idx-len+1, idx+1, value) ac.cr:271-273
          hbuf[w + 1] = rec.y;
          hbuf[w + 2] = (uint32_t)k;
          w += 3;
          k = (int32_t)ln.y;
        } while (k >= 0);
      }
      __syncthreads();
      const uint64_t room = base < M.cap ? M.cap - base : 0;
      const uint32_t nd = (uint32_t)(tot64 < room ? tot64 : room) * 3;
      uint32_t *dst = reinterpret_cast<uint32_t *>(M.out + base);
      for (uint32_t i = threadIdx.x; i < nd; i += 256) dst[i] = hbuf[i];
      __syncthreads();
    } else if (live) {
      uint64_t idx = base + off;
      int32_t k = (int32_t)rec.x;
      do {
        const uint2 ln = A.key_ln[k];
        if (idx < M.cap) {
          aha_hit h;
          h.start = (int32_t)rec.y - (int32_t)ln.x;
          h.end = (int32_t)rec.y;
          h.value = k;
          M.out[idx] = h;
        }
        idx++;
        k = (int32_t)ln.y;
      } while (k >= 0);
    }
  }
}

__global__ __launch_bounds__(256) void k2_expand(DevAut A, V2Args M) {
  __shared__ uint64_t sm[4];
  if (M.cursor[1]) return;
  const uint64_t n = M.totals[2];
  const uint64_t nblk = (n + 255) / 256;
  for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const uint64_t p = b * 256 + threadIdx.x;
    const bool live = p < n;
    uint64_t idx = M.blk_a[b] + block_excl_scan<uint64_t>(live ? M.sorted_cnt[p] : 0u, sm, nullptr);
    if (!live) continue;
    const uint4 rec = M.sorted_ev[p];  // {key, end_b, chunk, y}
    const uint64_t abs_pos = (uint64_t)rec.z * M.S + (rec.w & 0x7FFFu);
    int32_t end_c = 0;
    if (M.chars) {
      const uint32_t aux = M.sorted_aux[p];
      end_c = (int32_t)(aux >> 1);
      if (!(aux & 1u)) {
        // the document began before this chunk: add the lead bytes between
        // the document start and the chunk start
        const uint32_t d0 = M.chunk_doc0[rec.z];
        const uint64_t dchunk = M.doc_off[d0] / M.S;
        const uint64_t docg = M.lead_base[dchunk] + M.doc_lead_rank[d0];
        end_c += (int32_t)(M.lead_base[rec.z] - docg);
      }
    }
    for_each_hit(A, M, rec.x, abs_pos, rec.y, (rec.w & kLastFlag) != 0, [&](int32_t k, uint32_t len) {
      if (idx < M.cap) {
        aha_hit h;
        if (M.chars) {  // Hit(char_of_byte[start], char_of_byte[end-1]+1) matcher.cr:37
          h.start = end_c - (int32_t)A.key_kc[k] - 1;
          h.end = end_c;
        } else {  // Hit(idx-len+1, idx+1) ac.cr:271-273
          h.start = (int32_t)rec.y - (int32_t)len;
          h.end = (int32_t)rec.y;
        }
        h.value = k;
        M.out[idx] = h;
      }
      idx++;
    });
  }
}

// doc_hit_off[d] = index of the first hit at or after the document's start: the hits of the 256-event block in front of the
// document's first event (blk_a) + the hits of the events between.  One wave per document: up to 255 events whose
// counts are a gather each -- one thread per document walked them one after the other (42 us for 1024 documents, a
// seventh of cfg 2's step at 64 MiB).
template <bool PLAIN>
__global__ __launch_bounds__(256) void k2_doc_offsets(DevAut A, V2Args M) {
  if (M.cursor[1] || !M.doc_hit_off) return;
  const uint64_t d = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (d > M.n_docs) return;
  const uint64_t q = M.doc_off[d];
  const uint64_t n_ev = M.totals[2], n_hits = M.totals[0];
  uint64_t r = n_hits;
  if (q < M.n_bytes) {
    const uint64_t p = M.ev_base[q / M.S] + M.doc_ev_rank[d];
    if (p < n_ev) {
      const uint64_t b = p / 256;
      uint32_t sum = 0;
      if (PLAIN) {
        const uint2 *sv = reinterpret_cast<const uint2 *>(M.sorted_ev);
        for (uint64_t j = b * 256 + lane; j < p; j += 64) sum += A.key_cnt[sv[j].x];
      } else {
        for (uint64_t j = b * 256 + lane; j < p; j += 64) sum += M.sorted_cnt[j];
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
      r = M.blk_a[b] + sum;
    }
  }
  if (lane == 0) M.doc_hit_off[d] = r;
}

}  // namespace

// ------------------------------------------------ direct pipeline (plain mode)
// One wave per chunk region: hits per chunk (and the key of every event, written back over the
// state base in the compact format so that the expansion needs no second end_key lookup).
template <bool COMPACT>
__global__ __launch_bounds__(256) void k2d_count(DevAut A, V2Args M) {
  if (M.cursor[1]) return;
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * 256) >> 6;
  for (uint64_t c = wave; c < M.n_chunks; c += n_waves) {
    const uint32_t n = M.ev_cnt[c];
    uint2 *reg = M.evd + c * M.ev_stride;
    uint32_t sum = 0;
    for (uint32_t i = lane; i < n; i += 64) {
      // the event record gets the key and (8 bits, 255 = look it up) the chain length: key ids fit 24 bits
      uint32_t x = reg[i].x, cnt;
      if (COMPACT) {
        x = A.end_info[x];  // key id, or (flattened chains) the offset of its chain, | min(chain length, 255) << 24
        cnt = x >> 24;
        if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[x & 0xFFFFFFu].y : (x & 0xFFFFFFu)];
      } else if (A.chain) {
        const uint32_t key = x;
        x = A.key_info[key];
        cnt = x >> 24;
        if (cnt == 255u) cnt = A.key_cnt[key];
      } else {
        cnt = A.key_cnt[x];
        x |= min(cnt, 255u) << 24;
      }
      reg[i].x = x;
      sum += cnt;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_down(sum, d, 64);
    if (lane == 0) M.chunk_hits[c] = sum;
  }
}

// One wave (= one 64-thread workgroup) per chunk: in-wave scan of hits per event, chains
// assembled in LDS and streamed out with coalesced stores.
// kWaveStage hits are staged per batch of 64 events: 512 (6 KiB of LDS) for hit-dense input, 256 (3 KiB: more
// workgroups per CU to hide the table gathers) otherwise; a batch with more hits takes the direct-store path.
// CHARS (String overload, matcher.cr:34-39): the record's second word is the lead-byte count of the position
// (<< 1 | "counted from the document start") instead of the byte offset; hits are char offsets.
template <uint32_t kWaveStage, bool CHARS>
__global__ __launch_bounds__(64) void k2d_expand(DevAut A, V2Args M) {
  __shared__ __attribute__((aligned(16))) uint32_t hbuf[kWaveStage * 3 + 4];
  __shared__ uint32_t s_excl[64], s_co[64], s_end[64];
  __shared__ __attribute__((aligned(16))) uint8_t s_mark[kWaveStage];  // by hit index: 1 where an event's first hit stands
  const bool out16 = (reinterpret_cast<uintptr_t>(M.out) & 15u) == 0;
  if (M.cursor[1]) return;
  const int lane = threadIdx.x;
  for (uint64_t c = blockIdx.x; c < M.n_chunks; c += gridDim.x) {
    const uint32_t n = M.ev_cnt[c];
    if (n == 0) continue;
    const uint2 *reg = M.evd + c * M.ev_stride;
    const uint64_t base = M.hit_base[c];
    uint32_t run = 0;
    int32_t lead_adj = 0;  // CHARS: lead bytes between the start of the document that contains the chunk start and it
    if (CHARS) {
      const uint32_t d0 = M.chunk_doc0[c];
      const uint64_t dchunk = M.doc_off[d0] / M.S;
      lead_adj = (int32_t)(M.lead_base[c] - (M.lead_base[dchunk] + M.doc_lead_rank[d0]));
    }
    uint2 rec_next = (uint32_t)lane < n ? reg[lane] : make_uint2(0, 0);
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool live = i < n;
      uint2 rec = rec_next;
      // the next batch's records are requested now: their HBM round trip runs beside this batch's expansion
      rec_next = i + 64 < n ? reg[i + 64] : make_uint2(0, 0);
      uint32_t cnt = 0;
      if (live) {
        cnt = rec.x >> 24;  // packed by k2d_count
        rec.x &= 0xFFFFFFu;  // key id, or the offset of its flattened chain
        if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[rec.x].y : rec.x];
        if (CHARS) rec.y = (rec.y >> 1) + ((rec.y & 1u) ? 0u : (uint32_t)lead_adj);  // end offset in chars
      }
      const uint32_t incl = wave_incl_scan(cnt);
      const uint32_t tot = __shfl(incl, 63, 64);
      const uint32_t off = incl - cnt;
      const uint32_t n_live = min(n - i0, 64u);
      // Long chains (cfg 5: runs and suffix-closed families, 16 hits per event) or a batch beyond the stage: the
      // flattened chains make every HIT addressable, so the batch is expanded by hit index -- lane h takes hits
      // h, h + 64, ... of a window of kWaveStage hits, finds its event by a binary search over the events' exclusive
      // hit counts (LDS) and reads chain[offset + rank in the event].  A loop per event would run as long as the
      // wave's longest chain with most lanes idle, and a batch beyond the stage would store hit by hit.
      // (the hit-dense instantiation only: with about one hit per event the loop per event is the cheaper one)
      const bool by_hit = kWaveStage > 256 && A.chain && (tot >= 2 * n_live || tot > kWaveStage);
      if (by_hit) {
        s_excl[lane] = live ? off : tot;
        s_co[lane] = rec.x;
        s_end[lane] = rec.y;
      }
      if (by_hit || tot <= kWaveStage) {
        for (uint32_t h0 = 0; h0 < tot; h0 += kWaveStage) {  // one window unless by_hit
          const uint32_t nh = min(tot - h0, kWaveStage);
          // the window is staged at the dword phase of its place in the output (hit index * 3 mod 4), so that LDS and
          // global addresses are 16-byte aligned together and the body goes out as 16 bytes per lane
          const uint64_t first = base + run + h0;
          const uint32_t ph = out16 ? (uint32_t)((first * 3) & 3u) : 0u;
          if (by_hit) {
            __syncthreads();
            // the lane's hits of the window (lane, lane + 64, ...) in three sweeps -- all event lookups, all gathers, all LDS
            // writes -- so that the window costs one LDS depth and one gather latency, not one per hit.
            // Which event a hit belongs to: the events that start inside the window mark their first hit in s_mark; a hit's event
            // is then (events that start before its sweep) + (marks at or below its place in the sweep: a ballot and a count)
            // - 1 -- two dependent LDS reads per hit instead of the seven of a binary search over the events' hit counts.
            constexpr int kPer = (int)(kWaveStage / 64);
            uint32_t at[kPer], en[kPer];
            uint2 ce[kPer];
#pragma unroll
            for (int k = 0; k < kPer / 4; k++) reinterpret_cast<uint32_t *>(s_mark)[lane + 64 * k] = 0u;
            __syncthreads();
            if (live && off - h0 < nh) s_mark[off - h0] = 1;  // (off < h0 wraps around to a large number)
            uint32_t before = (uint32_t)__popcll(__ballot(live && off < h0));
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kPer; k++) {
              const uint32_t j = (uint32_t)lane + 64u * (uint32_t)k;
              const bool mk = j < nh && s_mark[j] != 0;
              const uint64_t marks = __ballot(mk);
              const uint32_t e = min(before + __builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u)) +
                                         (mk ? 1u : 0u) - 1u, 63u);  // (beyond the window: some event, never loaded)
              before += (uint32_t)__popcll(marks);
              at[k] = s_co[e] + (h0 + j - s_excl[e]);
              en[k] = s_end[e];
            }
#pragma unroll
            for (int k = 0; k < kPer; k++) {
              const bool in = (uint32_t)lane + 64u * (uint32_t)k < nh;
              ce[k] = in ? (CHARS ? A.chain_chars : A.chain)[at[k]] : make_uint2(0, 0);
            }
#pragma unroll
            for (int k = 0; k < kPer; k++) {
              const uint32_t j = (uint32_t)lane + 64u * (uint32_t)k;
              if (j < nh) {
                const uint32_t w = ph + j * 3;
                // Hit(idx-len+1, idx+1, value) ac.cr:271-273; chars: Hit(char_of_byte[start], char_of_byte[end-1]+1)
                hbuf[w] = en[k] - ce[k].x;
                hbuf[w + 1] = en[k];
                hbuf[w + 2] = ce[k].y;
              }
            }
          } else if (live && A.chain) {
            // fetch (ac.cr:265-278): own key, then the output chain -- from the flattened copy: consecutive loads
            uint32_t w = ph + off * 3;
            const uint32_t co = rec.x;
            for (uint32_t j = 0; j < cnt; j++) {
              const uint2 e = (CHARS ? A.chain_chars : A.chain)[co + j];
              hbuf[w] = rec.y - e.x;
              hbuf[w + 1] = rec.y;
              hbuf[w + 2] = e.y;
              w += 3;
            }
          } else if (live) {
            uint32_t w = ph + off * 3;
            int32_t k = (int32_t)rec.x;
            do {
              const uint2 ln = A.key_ln[k];
              hbuf[w] = CHARS ? rec.y - A.key_kc[k] - 1u : rec.y - ln.x;
              hbuf[w + 1] = rec.y;
              hbuf[w + 2] = (uint32_t)k;
              w += 3;
              k = (int32_t)ln.y;
            } while (k >= 0);
          }
          __syncthreads();
          const uint64_t room = first < M.cap ? M.cap - first : 0;
          const uint32_t nd = (uint32_t)(nh < room ? nh : room) * 3;
          uint32_t *dst = reinterpret_cast<uint32_t *>(M.out + first);
          if (out16 && nd >= 64) {
            const uint32_t head = min(nd, (4u - ph) & 3u);
            if ((uint32_t)lane < head) dst[lane] = hbuf[ph + lane];
            const uint32_t nq = (nd - head) >> 2;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(hbuf + ph + head);
            uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
            for (uint32_t q = lane; q < nq; q += 64) d4[q] = s4[q];
            const uint32_t done = head + (nq << 2);
            if (done + (uint32_t)lane < nd) dst[done + lane] = hbuf[ph + done + lane];
          } else {
            for (uint32_t j = lane; j < nd; j += 64) dst[j] = hbuf[ph + j];
          }
          __syncthreads();
        }
      } else if (live) {  // a batch beyond the stage (and no expansion by hit index): hit by hit
        uint64_t idx = base + run + off;
        int32_t k = (int32_t)(A.chain ? A.chain[rec.x].y : rec.x);
        do {
          const uint2 ln = A.key_ln[k];
          if (idx < M.cap) {
            aha_hit h;
            h.start = CHARS ? (int32_t)rec.y - (int32_t)A.key_kc[k] - 1 : (int32_t)rec.y - (int32_t)ln.x;
            h.end = (int32_t)rec.y;
            h.value = k;
            M.out[idx] = h;
          }
          idx++;
          k = (int32_t)ln.y;
        } while (k >= 0);
      }
      run += tot;
    }
  }
}

// doc_hit_off[d] = hits before the document's first event: the chunk's hit base + the chain lengths (carried by the
// records since k2d_count) of the chunk's events before it.  One thread per document; a chunk holds few events.
// LANES = 16: sixteen lanes (a DPP row) per document -- the events of the chunk before the document's first are summed sixteen
// at a time (one thread per document walks them one by one: 19 us for cfg 2's 16 384 documents with 70 events per 32 KiB
// chunk); LANES = 1: a thread per document, for batches of many small documents (few events before each).
template <int LANES>
__global__ __launch_bounds__(256) void k2d_doc_offsets(DevAut A, V2Args M) {
  // (the call's last kernel: every word the host reads is final -- this kernel changes none of them)
  if (M.publish && blockIdx.x == 0 && threadIdx.x < 5) M.publish[threadIdx.x] = M.cursor[threadIdx.x];
  if (M.cursor[1] || !M.doc_hit_off) return;
  const uint64_t d = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LANES;
  const uint32_t j = threadIdx.x & (uint32_t)(LANES - 1);
  const bool in = d <= M.n_docs;
  const uint64_t q = in ? M.doc_off[d] : M.n_bytes;
  uint64_t r = M.totals[0];
  uint32_t part = 0;
  bool summed = false;
  if (q < M.n_bytes) {
    const uint64_t c = q / M.S;
    const uint32_t rank = M.doc_ev_rank[d];
    if (rank >= M.ev_cnt[c]) {
      r = M.hit_base[c] + M.chunk_hits[c];
    } else {
      const uint2 *reg = M.evd + c * M.ev_stride;
      for (uint32_t i = j; i < rank; i += LANES) {
        const uint32_t x = reg[i].x;
        const uint32_t cnt = x >> 24;
        part += cnt == 255u ? A.key_cnt[A.chain ? A.chain[x & 0xFFFFFFu].y : (x & 0xFFFFFFu)] : cnt;
      }
      r = M.hit_base[c];
      summed = true;
    }
  }
  if (LANES == 16) {  // the row's sum in its last lane (row_shr 1, 2, 4, 8; lanes without a source add 0)
    part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x111, 0xf, 0xf, false);
    part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x112, 0xf, 0xf, false);
    part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x114, 0xf, 0xf, false);
    part += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)part, 0x118, 0xf, 0xf, false);
  }
  if (in && j == (uint32_t)(LANES - 1)) M.doc_hit_off[d] = r + (summed ? part : 0u);
}

// ---------------------------------------------------------------- launchers
size_t v2_lds_bytes(uint32_t lds_slots, bool compact) {
  return (size_t)lds_slots * (compact ? 4 : 8) + (size_t)(kV2Threads / 64) * kWaveIn2;
}

int v2_prepare(bool compact, size_t lds_bytes) {
  const void *fs[8] = {(const void *)k2_traverse<false, false, false>, (const void *)k2_traverse<false, true, false>,
                       (const void *)k2_traverse<false, false, true>,  (const void *)k2_traverse<false, true, true>,
                       (const void *)k2_traverse<true, false, false>,  (const void *)k2_traverse<true, true, false>,
                       (const void *)k2_traverse<true, false, true>,   (const void *)k2_traverse<true, true, true>};
  int rc = 0;
  for (int i = 0; i < 4; i++) {
    int e = (int)hipFuncSetAttribute(fs[(compact ? 4 : 0) + i], hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
    if (e) rc = e;
  }
  return rc;
}


void v2_launch_traverse(const DevAut &A, const V2Args &M, uint32_t grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = v2_lds_bytes(M.lds_slots, A.compact != 0);
  {
    const bool all = M.lds_slots >= A.n_slots;  // the whole image is in LDS
#define AHA_LAUNCH_K2(C, H, L) \
  hipLaunchKernelGGL((k2_traverse<C, H, L>), dim3(grid), dim3(kV2Threads), lds, s, A, M)
    if (A.compact) {
      if (M.chars) { if (all) AHA_LAUNCH_K2(true, true, true); else AHA_LAUNCH_K2(true, true, false); }
      else         { if (all) AHA_LAUNCH_K2(true, false, true); else AHA_LAUNCH_K2(true, false, false); }
    } else {
      if (M.chars) { if (all) AHA_LAUNCH_K2(false, true, true); else AHA_LAUNCH_K2(false, true, false); }
      else         { if (all) AHA_LAUNCH_K2(false, false, true); else AHA_LAUNCH_K2(false, false, false); }
    }
#undef AHA_LAUNCH_K2
  }
}

static inline uint32_t grid_for(uint64_t n_items, uint32_t per_block, uint32_t max_blocks) {
  uint64_t g = (n_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (uint32_t)(g > max_blocks ? max_blocks : g);
}

void v2_launch_chunk_scan(const V2Args &M, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  launch_scan(M.ev_cnt, M.n_chunks, M.blk_a, M.ev_base, M.totals + 2, nullptr, s);
  if (M.chars) launch_scan(M.lead_cnt, M.n_chunks, M.blk_b, M.lead_base, M.totals + 1, nullptr, s);
}

void v2_launch_sort(const DevAut &A, const V2Args &M, uint64_t n_records_hint, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t g = grid_for(n_records_hint, 256, 8192);
  const bool plain = !M.sep && !M.chars;
  if (A.compact) {
    if (plain)
      hipLaunchKernelGGL((k2_sort<true, true>), dim3(g), dim3(256), 0, s, A, M);
    else
      hipLaunchKernelGGL((k2_sort<true, false>), dim3(g), dim3(256), 0, s, A, M);
  } else {
    if (plain)
      hipLaunchKernelGGL((k2_sort<false, true>), dim3(g), dim3(256), 0, s, A, M);
    else
      hipLaunchKernelGGL((k2_sort<false, false>), dim3(g), dim3(256), 0, s, A, M);
  }
}

void v2_launch_expand(const DevAut &A, const V2Args &M, uint64_t n_events_hint, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t g = grid_for(n_events_hint, 256, 8192);
  const bool plain = !M.sep && !M.chars;
  // hits per event -> block sums -> bases (device-side counts: no host sync)
  if (plain)
    hipLaunchKernelGGL(k2_blocksum_keys, dim3(g), dim3(256), 0, s, A, M);
  else
    hipLaunchKernelGGL(k2_blocksum, dim3(g), dim3(256), 0, s, M.sorted_cnt, (const uint64_t *)(M.totals + 2),
                       (uint64_t)0, M.blk_a, (const unsigned long long *)(M.cursor + 1));
  hipLaunchKernelGGL(k2_scan_blocks, dim3(1), dim3(1024), 0, s, M.blk_a, (const uint64_t *)(M.totals + 2),
                     (uint64_t)0, M.totals + 0, (const unsigned long long *)(M.cursor + 1));
  if (plain)
    hipLaunchKernelGGL(k2_expand_plain, dim3(g), dim3(256), 0, s, A, M);
  else
    hipLaunchKernelGGL(k2_expand, dim3(g), dim3(256), 0, s, A, M);
  if (M.doc_hit_off) {
    const uint64_t nd = M.n_docs + 1;
    if (plain)
      hipLaunchKernelGGL(k2_doc_offsets<true>, dim3((uint32_t)((nd + 3) / 4)), dim3(256), 0, s, A, M);
    else
      hipLaunchKernelGGL(k2_doc_offsets<false>, dim3((uint32_t)((nd + 3) / 4)), dim3(256), 0, s, A, M);
  }
}

void v2_launch_hit_scan(const V2Args &M, void *stream) {
  launch_scan(M.chunk_hits, M.n_chunks, M.blk_a, M.hit_base, M.totals + 0, (const unsigned long long *)(M.cursor + 1),
              (hipStream_t)stream);
}

void v2_launch_lead_scan(const V2Args &M, void *stream) {  // lead_cnt -> lead_base, totals[1] (char offsets)
  launch_scan(M.lead_cnt, M.n_chunks, M.blk_b, M.lead_base, M.totals + 1, (const unsigned long long *)(M.cursor + 1),
              (hipStream_t)stream);
}

void v2_launch_direct_post(const DevAut &A, const V2Args &M, void *stream, void *ev_mid, bool counted) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t gw = grid_for(M.n_chunks, 4, 8192);  // 4 waves (chunks) per 256-thread block
  if (counted) {
    // (the character-level traversal's regroup pass has done it: unit_launch_regroup)
  } else if (A.compact) {
    hipLaunchKernelGGL(k2d_count<true>, dim3(gw), dim3(256), 0, s, A, M);
  } else {
    hipLaunchKernelGGL(k2d_count<false>, dim3(gw), dim3(256), 0, s, A, M);
  }
  const unsigned long long *abortf = (const unsigned long long *)(M.cursor + 1);
  launch_scan(M.chunk_hits, M.n_chunks, M.blk_a, M.hit_base, M.totals + 0, abortf, s);
  if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
  const dim3 ge(grid_for(M.n_chunks, 1, 1u << 20));
  if (M.chars) {
    // lead bytes before every chunk (the traversal counted them per chunk)
    launch_scan(M.lead_cnt, M.n_chunks, M.blk_b, M.lead_base, M.totals + 1, abortf, s);
    if (M.dense_hits)
      hipLaunchKernelGGL((k2d_expand<512, true>), ge, dim3(64), 0, s, A, M);
    else
      hipLaunchKernelGGL((k2d_expand<256, true>), ge, dim3(64), 0, s, A, M);
  } else if (M.dense_hits) {
    hipLaunchKernelGGL((k2d_expand<512, false>), ge, dim3(64), 0, s, A, M);
  } else {
    hipLaunchKernelGGL((k2d_expand<256, false>), ge, dim3(64), 0, s, A, M);
  }
  if (M.doc_hit_off) {
    const uint64_t nd = M.n_docs + 1;
    if (nd <= 4 * M.n_chunks)  // (few documents per chunk: many events can stand before a document's first)
      hipLaunchKernelGGL(k2d_doc_offsets<16>, dim3((uint32_t)((nd * 16 + 255) / 256)), dim3(256), 0, s, A, M);
    else
      hipLaunchKernelGGL(k2d_doc_offsets<1>, dim3((uint32_t)((nd + 255) / 256)), dim3(256), 0, s, A, M);
  }
}

}  // namespace aha
