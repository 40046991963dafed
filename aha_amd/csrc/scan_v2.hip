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
  scan_small_block(in, n, out, total, sm);
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
// doc_hit_off[d] = hits before the document's first event: the chunk's hit base + the chain lengths (carried by the
// records since k2d_count) of the chunk's events before it.  One thread per document; a chunk holds few events.
// LANES = 16: sixteen lanes (a DPP row) per document -- the events of the chunk before the document's first are summed sixteen
// at a time (one thread per document walks them one by one: 19 us for cfg 2's 16 384 documents with 70 events per 32 KiB
// chunk); LANES = 1: a thread per document, for batches of many small documents (few events before each).
// A block of BLOCK threads, the blk-th of the documents' blocks: rides on the expansion's launch (blocks behind the expansion's
// own: a launch of its own costs ~5 us of a 64 MiB call) or is k2d_doc_offsets' (batches the expansion's launch does not serve).
template <int LANES, int BLOCK>
__device__ __forceinline__ void doc_offsets_block(const DevAut &A, const V2Args &M, uint32_t blk) {
  // (part of the call's last launch, and nothing in that launch changes a word the host reads)
  if (M.publish && blk == 0 && threadIdx.x < 5) M.publish[threadIdx.x] = M.cursor[threadIdx.x];
  if (M.clear_next && blk == 0 && threadIdx.x < 16) M.clear_next[threadIdx.x] = 0ull;  // (the next call's counters)
  if (M.cursor[1] || !M.doc_hit_off) return;
  const uint64_t d = ((uint64_t)blk * BLOCK + threadIdx.x) / LANES;
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

template <int LANES>
__global__ __launch_bounds__(256) void k2d_doc_offsets(DevAut A, V2Args M) {
  doc_offsets_block<LANES, 256>(A, M, blockIdx.x);
}

// (the expansion kernels' first lines: blocks from M.doc_from on are the documents')
__device__ __forceinline__ bool expand_block_is_docs(const DevAut &A, const V2Args &M) {
  if (!M.doc_from || blockIdx.x < M.doc_from) return false;
  if (M.doc_lanes16)
    doc_offsets_block<16, 64>(A, M, blockIdx.x - M.doc_from);
  else
    doc_offsets_block<1, 64>(A, M, blockIdx.x - M.doc_from);
  return true;
}

// The block is ONE wave: what its lanes hand one another through LDS needs the order of the wave's own LDS instructions
// (which the hardware keeps) and a compiler fence -- not __syncthreads(), whose s_waitcnt vmcnt(0) would also wait for the
// window's stores to be acknowledged by memory before the next window's first LDS write.
#ifndef AHA_EXPAND_LAB
#define AHA_EXPAND_LAB 0
#endif
__device__ __forceinline__ void expand_sync() {
#ifdef AHA_EXPAND_SYNCTHREADS
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

template <uint32_t kWaveStage, bool CHARS>
__global__ __launch_bounds__(64) void k2d_expand(DevAut A, V2Args M) {
  __shared__ __attribute__((aligned(16))) uint32_t hbuf[kWaveStage * 3 + 4];
  __shared__ uint32_t s_excl[64], s_co[64], s_end[64];
  __shared__ __attribute__((aligned(16))) uint8_t s_mark[kWaveStage];  // by hit index: 1 where an event's first hit stands
  const bool out16 = (reinterpret_cast<uintptr_t>(M.out) & 15u) == 0;
  if (expand_block_is_docs(A, M)) return;
  if (M.cursor[1]) return;
  const int lane = threadIdx.x;
  const uint32_t n_blocks = M.doc_from ? M.doc_from : gridDim.x;
  for (uint64_t c = blockIdx.x; c < M.n_chunks; c += n_blocks) {
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
            expand_sync();
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
            expand_sync();
            if (live && off - h0 < nh) s_mark[off - h0] = 1;  // (off < h0 wraps around to a large number)
            uint32_t before = (uint32_t)__popcll(__ballot(live && off < h0));
            expand_sync();
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
#if AHA_EXPAND_LAB & 2  // (lab: no gathers)
              ce[k] = make_uint2(1u, at[k]);
#else
              ce[k] = in ? (CHARS ? A.chain_chars : A.chain)[at[k]] : make_uint2(0, 0);
#endif
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
          expand_sync();
          const uint64_t room = first < M.cap ? M.cap - first : 0;
          const uint32_t nd = (uint32_t)(nh < room ? nh : room) * 3;
          uint32_t *dst = reinterpret_cast<uint32_t *>(M.out + first);
#if AHA_EXPAND_LAB & 1  // (lab: no stores)
          if (nd == 0xFFFFFFFFu) dst[lane] = hbuf[lane];
          else if (false) {
#else
          if (out16 && nd >= 64) {
#endif
            const uint32_t head = min(nd, (4u - ph) & 3u);
            if ((uint32_t)lane < head) dst[lane] = hbuf[ph + lane];
            const uint32_t nq = (nd - head) >> 2;
            const uint4 *s4 = reinterpret_cast<const uint4 *>(hbuf + ph + head);
            uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
            for (uint32_t q = lane; q < nq; q += 64) d4[q] = s4[q];
            const uint32_t done = head + (nq << 2);
            if (done + (uint32_t)lane < nd) dst[done + lane] = hbuf[ph + done + lane];
          } else {
#if !(AHA_EXPAND_LAB & 1)
            for (uint32_t j = lane; j < nd; j += 64) dst[j] = hbuf[ph + j];
#endif
          }
          expand_sync();
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

// ---- the hit-dense expansion (cfg 5: 3.5 hits per byte, 11 GB of hits per step), software-pipelined.
// k2d_expand<512> runs a window's steps one behind the other, and every wait of the wave for a load -- the chain gathers, the
// next events' records -- is, on gfx950, also a wait for every store issued before it: vmcnt counts loads and stores in one
// order, and the compiler, which cannot count the stores of a loop, waits for 0.  The stores of a window (6 KiB) are
// acknowledged by memory after microseconds; a wave of k2d_expand<512> stands 9 400 of its 14 500 clocks per window in front of
// such a wait (profiles/r06_expand_pipeline.txt).  Here a window's life is cut in three --
//   A   events' marks, the lookups, the gathers issued           (reads s_excl / s_co / s_end, writes s_mark)
//   B1  the gathers' results into the staged window              (writes hbuf)
//   B2  the staged window out, 16 bytes a lane                   (reads hbuf)
// -- the wave runs  A(w0) B1(w0);  { A(w + 1); B2(w); wait; B1(w + 1); }  and every load of the loop (the gathers, the next
// batch's records) and every store of B2 is HAND-ISSUED, so that the one wait can name how many instructions younger than
// the loads -- B2's stores, exactly -- may stay outstanding.  What that takes from the compiler, the code has to guarantee
// itself:
//   * a hand-issued load's register holds nothing until the wait: between the asm statement and the wait the VALUE must not
//     be merged with another definition (hipcc resolves a phi with a copy of the register, then and there): the loads stand
//     unconditionally in the loop body -- a window that does not exist gathers entry 0 and is not filled -- and are handed
//     to ordinary variables only behind the wait;
//   * the count must not be larger than the stores really issued: B2's counted form issues them with every lane in every
//     instruction (no predicate the compiler could turn into a skipped instruction); a window it cannot take (under 64
//     dwords, or beyond the capacity) goes out through ordinary stores and the wait is for 0.
// The next window may be the first of the next batch of events (its scan and LDS tables are set up inside A); a batch takes
// 64 events or, if those hold more hits than a window, as many as fill one.  A batch that is not expanded by hit index (few
// hits per event, or no flattened chains) drains the pipeline and takes k2d_expand's per-event path.
template <bool CHARS>
__global__ __launch_bounds__(64) void k2d_expand_dense(DevAut A, V2Args M) {
  constexpr uint32_t kStage = 512;
  constexpr int kPer = (int)(kStage / 64);
  typedef uint32_t v2u __attribute__((ext_vector_type(2)));
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) uint32_t hbuf[kStage * 3 + 4];
  __shared__ uint4 s_ev[64];  // per event of the batch: {hits of the batch before it, offset of its flattened chain, its end offset, -}
  __shared__ __attribute__((aligned(16))) uint8_t s_mark[kStage];
#ifdef AHA_EXPAND_CLK
  const uint64_t k_start = clock64();
  uint64_t k_chunks = 0;
#endif
  const bool out16 = (reinterpret_cast<uintptr_t>(M.out) & 15u) == 0;
  if (expand_block_is_docs(A, M)) return;
  if (M.cursor[1]) return;
  const int lane = threadIdx.x;
  const uint2 *chain = CHARS ? A.chain_chars : A.chain;
  const uint32_t n_blocks = M.doc_from ? M.doc_from : gridDim.x;
  for (uint64_t c = blockIdx.x; c < M.n_chunks; c += n_blocks) {
    const uint32_t n = M.ev_cnt[c];
    if (n == 0) continue;
    const uint2 *reg = M.evd + c * M.ev_stride;
    const uint64_t base = M.hit_base[c];
    int32_t lead_adj = 0;
    if (CHARS) {
      const uint32_t d0 = M.chunk_doc0[c];
      const uint64_t dchunk = M.doc_off[d0] / M.S;
      lead_adj = (int32_t)(M.lead_base[c] - (M.lead_base[dchunk] + M.doc_lead_rank[d0]));
    }
#ifdef AHA_EXPAND_CLK
    uint64_t c_a = 0, c_b2 = 0, c_wait = 0, c_fill = 0, c_other = 0, n_win = 0, t_last = clock64();
    const uint64_t t_start = t_last, w_start = wall_clock64();
#define CLK_MARK(acc) { const uint64_t t_ = clock64(); acc += t_ - t_last; t_last = t_; }
#else
#define CLK_MARK(acc)
#endif
    // the batch being prepared: events i0 .. i0 + n_ev - 1, `run` hits of the chunk before it, h0 = its next window
    uint32_t i0 = 0, run = 0, h0 = 0, n_ev = 0;
    bool have = false, by_hit = false, live = false;
    uint2 rec = make_uint2(0, 0);
    uint32_t cnt = 0, off = 0, tot = 0;
    // the records of events ready_from .. ready_from + 63 (a lane beyond the chunk's events holds the last one's and ignores it)
    uint2 rec_ready = make_uint2(0, 0);
    uint32_t ready_from = ~0u;

    // (rec_ready holds the records from i0.)  An event whose chain is longer than the record's count field says (255: look
    // it up) needs a load the compiler counts -- and a load it counts anywhere in the pipeline's loop puts a wait for 0 on the
    // loop's common path: inside the loop the batch is loaded WITHOUT the lookup, and one that needs it is left to the caller
    // outside (returns false, nothing changed).
    auto load_batch = [&](auto in_pipeline) -> bool {
      const uint32_t i = i0 + lane;
      const bool lv = i < n;
      if (decltype(in_pipeline)::value && __ballot(lv && (rec_ready.x >> 24) == 255u)) return false;
      live = lv;
      rec = rec_ready;
      cnt = 0;
      if (live) {
        cnt = rec.x >> 24;
        rec.x &= 0xFFFFFFu;
        if (!decltype(in_pipeline)::value) {
          if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[rec.x].y : rec.x];
        }
        if (CHARS) rec.y = (rec.y >> 1) + ((rec.y & 1u) ? 0u : (uint32_t)lead_adj);
      }
      const uint32_t incl = wave_incl_scan(cnt);
      tot = __shfl(incl, 63, 64);
      off = incl - cnt;
      n_ev = min(n - i0, 64u);
      // 64 events of 9 hits are a window and a ninth of one: a batch that would spill takes only the events whose hits fill
      // ONE window (at least 16 of them), the rest open the next batch
      if (A.chain && tot > kStage) {
        const uint32_t m = (uint32_t)__popcll(__ballot(live && incl <= kStage));
        if (m >= 16u) {
          n_ev = m;
          tot = __shfl(incl, (int)m - 1, 64);
          live = (uint32_t)lane < m;
          cnt = live ? cnt : 0u;
        }
      }
      by_hit = A.chain && (tot >= 2 * n_ev || tot > kStage);
      h0 = 0;
      have = true;
      if (by_hit) s_ev[lane] = make_uint4(live ? off : tot, rec.x, rec.y, 0u);
      return true;
    };
    auto batch_done = [&]() {
      run += tot;
      i0 += n_ev;
      have = false;
    };
    // B2, uncounted: the staged window out through ordinary stores (k2d_expand's sequence)
    auto stage_out = [&](uint64_t first, uint32_t nh, uint32_t ph) {
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
      expand_sync();
    };

    for (;;) {
      if (!have) {
        if (i0 >= n) break;
        if (ready_from != i0) {  // (the chunk's first batch, or the one behind a batch that went the per-event way)
          v2u r;
          const uint2 *ptr = reg + min(i0 + (uint32_t)lane, n - 1u);
          asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(ptr) : "memory");
          rec_ready = make_uint2(r.x, r.y);
          ready_from = i0;
        }
        load_batch(std::false_type{});
      }
      if (!by_hit) {
        // ---- not by hit index: one window filled event by event, or -- beyond the stage without flattened chains -- hit by hit
        if (tot <= kStage) {
          const uint64_t first = base + run;
          const uint32_t ph = out16 ? (uint32_t)((first * 3) & 3u) : 0u;
          if (live && A.chain) {
            uint32_t w = ph + off * 3;
            for (uint32_t j = 0; j < cnt; j++) {
              const uint2 e = chain[rec.x + j];
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
          expand_sync();
          if (tot) stage_out(first, tot, ph);
        } else if (live) {
          uint64_t idx = base + run + off;
          int32_t k = (int32_t)rec.x;  // (no flattened chains here: by_hit would have taken the batch)
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
        batch_done();
        continue;
      }
      // ---- by hit index, pipelined
      uint32_t at[kPer], en[kPer];
      uint64_t w_first = 0;
      uint32_t w_nh = 0, w_ph = 0;
      // A without the gathers: window h0 of the loaded batch -> at / en, its place and size -> w_*; moves on to the next window
      auto lookups = [&]() {
        const uint32_t nh = min(tot - h0, kStage);
        w_first = base + run + h0;
        w_nh = nh;
        w_ph = out16 ? (uint32_t)((w_first * 3) & 3u) : 0u;
        expand_sync();
#pragma unroll
        for (int k = 0; k < kPer / 4; k++) reinterpret_cast<uint32_t *>(s_mark)[lane + 64 * k] = 0u;
        expand_sync();
        if (live && off - h0 < nh) s_mark[off - h0] = 1;  // (off < h0 wraps around to a large number)
        uint32_t before = (uint32_t)__popcll(__ballot(live && off < h0));
        expand_sync();
        // (no predicate on an LDS read: a read under a condition is a branch and a wait of its own, eight in a row -- the
        // marks beyond the window are 0, every event index is one of the 64)
        uint32_t mv[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) mv[k] = s_mark[(uint32_t)lane + 64u * (uint32_t)k];
        uint32_t ev[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          const uint32_t j = (uint32_t)lane + 64u * (uint32_t)k;
          const bool mk = j < nh && mv[k] != 0;
          const uint64_t marks = __ballot(mk);
          ev[k] = min(before + __builtin_amdgcn_mbcnt_hi((uint32_t)(marks >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)marks, 0u)) +
                          (mk ? 1u : 0u) - 1u, 63u);
          before += (uint32_t)__popcll(marks);
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          const uint32_t j = (uint32_t)lane + 64u * (uint32_t)k;
          uint4 e = s_ev[ev[k]];
          asm volatile("" : "+v"(e.x), "+v"(e.y));  // (read by every lane: hipcc would sink the read into a branch on j < nh)
#ifdef AHA_EXPAND_NOGATHER  // (lab: every gather reads entry 0 -- wrong hits, the time without the gathers' misses)
          at[k] = (j < nh && e.y == 0xFFFFFFFFu) ? e.y + (h0 + j - e.x) : 0u;
#else
          at[k] = j < nh ? e.y + (h0 + j - e.x) : 0u;  // (beyond the window: entry 0, gathered and ignored)
#endif
          en[k] = e.z;
        }
        h0 += kStage;
        if (h0 >= tot) batch_done();
      };
      // B1: gathered entries into the staged window
      auto fill = [&](const uint2 (&ce)[kPer], uint32_t nh, uint32_t ph) {
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          // (no predicate: a place beyond the window's hits still lies inside hbuf, and nothing reads it)
          const uint32_t w = ph + ((uint32_t)lane + 64u * (uint32_t)k) * 3;
          hbuf[w] = en[k] - ce[k].x;  // Hit(idx-len+1, idx+1, value) ac.cr:271-273
          hbuf[w + 1] = en[k];
          hbuf[w + 2] = ce[k].y;
        }
        (void)nh;
        expand_sync();
      };
      CLK_MARK(c_other)
      bool prev = false;  // a staged window waits to go out (not in the pipeline's first round)
      uint64_t p_first = 0;
      uint32_t p_nh = 0, p_ph = 0;
      for (;;) {
        // A of the next window, if it is one of this pipeline's (in the first round: the window of the batch just loaded)
        if (!have && i0 < n && ready_from == i0) (void)load_batch(std::true_type{});
        const bool nx = have && by_hit;
        if (nx) {
          lookups();
        } else {
#pragma unroll
          for (int k = 0; k < kPer; k++) at[k] = 0u;
        }
        // ... its gathers and the records of the batch behind the loaded one (or, while none is loaded, of the next): in flight
        // until the wait below, their registers not to be looked at
        const uint32_t want_from = have ? i0 + n_ev : i0;
        v2u ce_raw[kPer], rec_raw;
        {
          const uint2 *ptr = reg + min(want_from + (uint32_t)lane, n - 1u);
          asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rec_raw) : "v"(ptr) : "memory");
        }
#pragma unroll
        for (int k = 0; k < kPer; k++) {
          const uint32_t byte_off = at[k] << 3;  // (the table's base in scalar registers: no 64-bit address per lane)
          asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(ce_raw[k]) : "v"(byte_off), "s"(chain) : "memory");
        }
        CLK_MARK(c_a)
        // B2.  A window of 64 dwords or more that fits the output goes out as a COUNTED sequence of stores, every lane in every
        // one of them: dwords 0 .. 63, the 16-byte quads from the first 128-byte line on in six rounds of 64 (a lane beyond the last
        // quad stores the last quad again), dwords nd - 64 .. nd - 1.  What lanes store twice they store with the same value.
        uint32_t counted = 0;  // stores issued that way (wave-uniform); 0: ordinary stores, uncounted
        {
          const uint64_t room = p_first < M.cap ? M.cap - p_first : 0;
          const uint32_t nd = p_nh * 3;
          if (!prev) {
            // (nothing staged yet)
          } else if (__builtin_amdgcn_readfirstlane((out16 && nd >= 64 && p_nh <= room) ? 1 : 0)) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(M.out + p_first);
            const uint32_t head = (uint32_t)(((128u - ((uint32_t)reinterpret_cast<uintptr_t>(dst) & 127u)) & 127u) >> 2);  // < 32, = (4 - ph) mod 4
            const uint32_t nq = (nd - head) >> 2;  // >= 8, <= 384
            // (the window's address is the wave's: scalar registers, a 32-bit offset per lane)
            const uint64_t dsta = reinterpret_cast<uint64_t>(dst);
            const uint64_t sdst = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dsta >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dsta);
            {
              const uint32_t o0 = (uint32_t)lane * 4u;
              const uint32_t d0 = hbuf[p_ph + lane];
              // (s_nop: v_readfirstlane has just written the scalar pair, and a VMEM instruction may read an SGPR a VALU
              // instruction wrote only five wait states later -- a hazard hipcc pads for its own instructions, not inside asm)
              asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" ::"v"(o0), "v"(d0), "s"(sdst) : "memory");
            }
            // (always the six rounds of a full window -- a window of the cut batches is nearly one --, the six LDS reads in
            // front of the six stores)
            const uint4 *s4 = reinterpret_cast<const uint4 *>(hbuf + p_ph + head);
            // (three at a time: six quads in registers beside the gathers in flight would cost the fifth wave of a SIMD)
#pragma unroll
            for (int g = 0; g < 2; g++) {
              uint4 tq[3];
              uint32_t oq[3];
#pragma unroll
              for (int r = 0; r < 3; r++) {
                const uint32_t q = min((uint32_t)(g * 3 + r) * 64u + (uint32_t)lane, nq - 1u);
                tq[r] = s4[q];
                oq[r] = head * 4u + q * 16u;
              }
#pragma unroll
              for (int r = 0; r < 3; r++) {
                const v4u dv = {tq[r].x, tq[r].y, tq[r].z, tq[r].w};
                // (s_nop: the instruction behind a store of more than 8 bytes must not write the store's data registers --
                // one wait state, which hipcc keeps for its own stores and cannot see to keep here)
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(oq[r]), "v"(dv), "s"(sdst) : "memory");
              }
            }
            {
              const uint32_t o1 = ((nd - 64u) + (uint32_t)lane) * 4u;
              const uint32_t d1 = hbuf[p_ph + (nd - 64u) + lane];
              asm volatile("global_store_dword %0, %1, %2" ::"v"(o1), "v"(d1), "s"(sdst) : "memory");
            }
            counted = 2u + (uint32_t)(kPer - 2);
            expand_sync();
          } else {
            stage_out(p_first, p_nh, p_ph);
          }
        }
        CLK_MARK(c_b2)
        if (counted == 8u)
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CLK_MARK(c_wait)
        // (behind the wait: the loads' registers are values now)
        rec_ready = make_uint2(rec_raw.x, rec_raw.y);
        ready_from = want_from;
        if (!nx) break;
        {
          uint2 ce[kPer];
#pragma unroll
          for (int k = 0; k < kPer; k++) ce[k] = make_uint2(ce_raw[k].x, ce_raw[k].y);
          fill(ce, w_nh, w_ph);
        }
        p_first = w_first;
        p_nh = w_nh;
        p_ph = w_ph;
        prev = true;
        CLK_MARK(c_fill)
#ifdef AHA_EXPAND_CLK
        n_win++;
#endif
      }
    }
#ifdef AHA_EXPAND_CLK
    if (lane == 0 && (c & 63) == 0) {  // (a sample: one chunk in 64)
      c_other += clock64() - t_last;
      atomicAdd(&M.cursor[8], (unsigned long long)c_a);
      atomicAdd(&M.cursor[9], (unsigned long long)c_b2);
      atomicAdd(&M.cursor[10], (unsigned long long)c_wait);
      atomicAdd(&M.cursor[11], (unsigned long long)c_fill);
      atomicAdd(&M.cursor[12], (unsigned long long)(wall_clock64() - w_start));  // (100 MHz)
      atomicAdd(&M.cursor[13], (unsigned long long)n_win);
      atomicAdd(&M.cursor[14], (unsigned long long)(clock64() - t_start));
      atomicAdd(&M.cursor[15], 1ull);
    }
    k_chunks++;
#endif
  }
#ifdef AHA_EXPAND_CLK
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (what the wave's end waits for anyway)
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
    atomicAdd(&M.cursor[5], (unsigned long long)(clock64() - k_start));
    atomicAdd(&M.cursor[6], (unsigned long long)k_chunks);
    atomicAdd(&M.cursor[7], 1ull);
  }
#endif
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

void v2_launch_direct_post(const DevAut &A, const V2Args &M0, void *stream, void *ev_mid, bool counted) {
  V2Args M = M0;
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
  // the dense expansion: a wave's first loads and the acknowledgement of its last stores (a wave ends when they are in) cost
  // ~35 000 clocks beside the ~90 000 of a chunk's windows -- its blocks take a few chunks each (AHA_EXPAND_BLOCKS, lab: 81 920 blocks 3.27 ms, one per chunk 3.48, 20 480 3.38, 5 120 3.78)
  uint32_t dense_blocks = 256u * 20u * 16u;
  if (const char *e = getenv("AHA_EXPAND_BLOCKS")) dense_blocks = (uint32_t)std::max(1, atoi(e));  // (lab)
  dim3 gd((uint32_t)std::min<uint64_t>(M.n_chunks ? M.n_chunks : 1, dense_blocks));
  dim3 gx = ge;
  // the documents' hit offsets ride on the expansion's launch: blocks of 64 threads behind the expansion's own
  const uint64_t nd = M.n_docs + 1;
  const bool lanes16 = nd <= 4 * M.n_chunks;  // (few documents per chunk: many events can stand before a document's first)
  const uint64_t doc_blocks = lanes16 ? (nd * 16 + 63) / 64 : (nd + 63) / 64;
#ifdef AHA_EXPAND_OLD
  const bool ride = false;
#else
  const bool ride = M.doc_hit_off && doc_blocks + gx.x < (1u << 22) && !getenv("AHA_DOC_OFFSETS_LAUNCH");  // (the variable: lab, a launch of their own)
#endif
  if (ride) {
    M.doc_lanes16 = lanes16 ? 1 : 0;
    M.doc_from = M.dense_hits ? gd.x : gx.x;
    gd.x += (uint32_t)doc_blocks;
    gx.x += (uint32_t)doc_blocks;
  }
  if (M.chars) {
    // lead bytes before every chunk (the traversal counted them per chunk)
    launch_scan(M.lead_cnt, M.n_chunks, M.blk_b, M.lead_base, M.totals + 1, abortf, s);
    if (M.dense_hits)
#ifdef AHA_EXPAND_OLD
      hipLaunchKernelGGL((k2d_expand<512, true>), ge, dim3(64), 0, s, A, M);
#else
      hipLaunchKernelGGL((k2d_expand_dense<true>), gd, dim3(64), 0, s, A, M);
#endif
    else
      hipLaunchKernelGGL((k2d_expand<256, true>), gx, dim3(64), 0, s, A, M);
  } else if (M.dense_hits) {
#ifdef AHA_EXPAND_OLD
    hipLaunchKernelGGL((k2d_expand<512, false>), ge, dim3(64), 0, s, A, M);
#else
    hipLaunchKernelGGL((k2d_expand_dense<false>), gd, dim3(64), 0, s, A, M);
#endif
  } else {
    hipLaunchKernelGGL((k2d_expand<256, false>), gx, dim3(64), 0, s, A, M);
  }
  if (M.doc_hit_off && !ride) {
    if (lanes16)
      hipLaunchKernelGGL(k2d_doc_offsets<16>, dim3((uint32_t)((nd * 16 + 255) / 256)), dim3(256), 0, s, A, M);
    else
      hipLaunchKernelGGL(k2d_doc_offsets<1>, dim3((uint32_t)((nd + 255) / 256)), dim3(256), 0, s, A, M);
  }
}

}  // namespace aha
