// scan_pair.hip -- the pair engine for gfx950 (engine 7): src/aha/ac.cr:176-192 with the state resolved WITHOUT a walk wherever
// it has at most two units -- which is almost everywhere.
//
// Why (DESIGN.md sections 4.4, 4.7, 4.8): a per-lane state machine costs ~410 issue cycles per unit (ku_traverse, 2.24 ms per
// GiB of cfg 3); a stateless look at every unit costs a fifth (ks_mark, 0.67 ms); and a walk that is started only at marked
// positions pays a scattered text request per mark (ks_traverse, 3.75 ms).  But a mark needs no walk at all: the reference's
// state after a unit is the longest suffix of the text that is a trie path, so where no path of THREE units ends, the state is
// the two-unit state of the last two units (if they spell a path) -- a function of two units the stateless pass has in its
// registers.  With no one-unit key (unit.hpp MARKS) an END two-unit state is an event and nothing else is, so:
//
//   kp_pairs   the marking pass (scan_skip.hip ks_mark) that also ANSWERS its marks: a pair that passes the filter is looked
//              up in a perfect hash table keyed by the two units' raw bytes (unit.hpp PAIR TABLE; one 16-byte load, requested
//              in one iteration and consumed in the next) -- key check, event payload, child filter.  An END pair is an event,
//              written into the tile's region in position order (the byte-level engine's records: {payload, end offset in the
//              document}).  A pair whose child filter lets the THIRD unit through is a deep candidate: the walk leaves three
//              empty records behind the pair's event and notes the position.
//   kp_walk    a lane per candidate: the goto walk from the root at the candidate (the unit image's probes, no fail links: the
//              walk of ONE start) -- its reach and its END states of three units or more.
//   kp_void    the events a deep walk covers -- every record that ends behind its pair and at or before its reach: a longer
//              suffix is alive there, the two-unit state is not the reference's -- become empty records.
//   kp_fill    a walk's own END events go into the three records behind its pair -- those that end behind the reach of every
//              EARLIER walk (the earliest start alive owns a position).
// and the byte-level engine's scan, k2d_expand and k2d_doc_offsets finish the call.  tests/pairsim.py is the CPU twin.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"
#include "unit.hpp"

namespace aha {

namespace {

#ifndef AHA_PP_LAB
#define AHA_PP_LAB 0  // (lab switches: timing-only variants of kp_pairs)
#endif
constexpr int kPpPiece = 32;                  // bytes of text per lane of kp_pairs
constexpr int kPpRow = 44;                    // its LDS row: the piece + 12 bytes (a pair's second and third unit behind a unit at byte 31)
constexpr int kPpTile = 64 * kPpPiece;        // a wave's tile = a chunk of the event regions (V2Args.S)
constexpr int kPpPool = 320;                  // event payloads a wave keeps for its tile (LDS; a lane's are chained); more events in
                                              // one tile of 2 KiB -- one per six bytes -- and the call goes to engine 4
constexpr uint32_t kPpPlace = 3;              // empty records behind a deep candidate's pair (unit.hpp kPairMaxDeepEnds)
constexpr uint32_t kPpNull = 0x00FFFFFFu;     // a record that stands for no hit (count 0)

__host__ __device__ inline size_t pp_lds(uint32_t bloom_log2, uint32_t groups) {
  return ((size_t)4 << bloom_log2) + ((groups + 15u) & ~15u) + (size_t)kV2Threads * kPpRow + (size_t)(kV2Threads / 64) * (kPpPool * 6 + 16) + 16;
}

typedef uint32_t pp_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t pp_v3u __attribute__((ext_vector_type(3)));

__device__ __forceinline__ uint64_t wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool wany(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// a deep candidate as kp_pairs leaves it, and what kp_walk makes of it
struct PpCand {
  uint32_t tile;
  uint32_t start_end;  // start - tile start | (pair end - tile start) << 12
  uint32_t slot;       // index, in the tile's region, of the first empty record behind the pair's event
  uint32_t dstart_lo;  // start of the candidate's document, relative to the tile start, + 2^31 (it lies at or before the candidate)
  uint32_t dend_lo;    // its end likewise (clamped to 2^20 behind the tile start: a walk is at most 255 bytes long)
  uint32_t next_doc;   // index of that boundary
};
struct PpWalk {
  uint32_t reach;      // first byte behind the walk's last unit - start of the candidate (keys are at most 255 bytes long);
                       // <= pair end - start: the third unit did not continue the pair after all
  uint32_t next_doc;   // index of the first document boundary behind the candidate's start
  uint32_t ends;       // number of END states of three units or more | (end of the END unit - start) << 8, 16, 24
  uint32_t end_base[kPpPlace];  // their bases
};

// ---- first boundary at or behind every tile's start (kp_pairs reads one word per tile instead of searching)
__global__ __launch_bounds__(256) void kp_tile_doc(V2Args M, uint32_t *tile_dn) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= M.n_chunks || M.cursor[1] >= 16ull) return;
  tile_dn[t] = (uint32_t)first_boundary(M.doc_off, M.n_docs, t * (uint64_t)kPpTile);
}

// ---- the pair pass.  LDS: filter, displacement bytes, a 44-byte row per lane, a pool of payloads per wave.
__global__ __launch_bounds__(kV2Threads) void kp_pairs(PairDev P, DevAut A, V2Args M, const uint32_t *__restrict__ tile_dn,
                                                       PpCand *__restrict__ cand, uint32_t seg_cap, uint32_t *__restrict__ seg_cnt) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;
  const uint32_t words = 1u << P.bloom_log2;
  uint32_t *bl = reinterpret_cast<uint32_t *>(smem);
  uint8_t *dsp = reinterpret_cast<uint8_t *>(bl + words);
  uint8_t *rows = dsp + ((P.groups + 15u) & ~15u);
  uint8_t *pools = rows + (size_t)kV2Threads * kPpRow;
  for (uint32_t i = threadIdx.x; i < words; i += kV2Threads) bl[i] = P.bloom[i];
  for (uint32_t i = threadIdx.x; i < P.groups; i += kV2Threads) dsp[i] = P.disp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lb = (uint32_t)(rows - smem) + (uint32_t)(wave * 64 + lane) * kPpRow;
  uint32_t *px = reinterpret_cast<uint32_t *>(pools + (size_t)wave * (kPpPool * 6 + 16));  // the wave's pool: payloads,
  uint16_t *pn = reinterpret_cast<uint16_t *>(px + kPpPool);                                // the lane's next entry,
  uint32_t *pcnt = reinterpret_cast<uint32_t *>(pn + kPpPool);                              // entries taken
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const uint64_t n_tiles = M.n_chunks;
  const uint64_t wave_id = (uint64_t)blockIdx.x * (kV2Threads / 64) + wave, n_waves = (uint64_t)gridDim.x * (kV2Threads / 64);
  const uint32_t wshift = 32u - P.bloom_log2, tmask = (1u << P.log2) - 1u, tshift = 32u - P.log2;
  const uint32_t stride = M.ev_stride;
  // the deep candidates go into the WAVE's own segment of the list (seg_cap entries; one counter of all tiles' candidates --
  // an atomic per tile on one word -- cost 5.7 ms per GiB: half a million additions to one address take their turns in L2)
  uint32_t cused = 0;
  for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    const int64_t t0 = (int64_t)tile * kPpTile;
    const int64_t g0 = t0 + (int64_t)lane * kPpPiece;
    {  // the piece and 12 bytes behind it (the corpus is 16-byte aligned: capi.cpp; bytes beyond the text read as 0)
      pp_v4u v0 = {0, 0, 0, 0}, v1 = v0;
      pp_v3u v2 = {0, 0, 0};
      if (g0 + kPpRow <= N) {
        v0 = *reinterpret_cast<const pp_v4u *>(M.text + g0);
        v1 = *reinterpret_cast<const pp_v4u *>(M.text + g0 + 16);
        v2 = *reinterpret_cast<const pp_v3u *>(M.text + g0 + 32);
      } else if (g0 < N) {
        uint32_t w[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
        for (int j = 0; j < kPpRow && g0 + j < N; j++) {
          const uint32_t b = (uint32_t)M.text[g0 + j] << ((j & 3) * 8);
#pragma unroll
          for (int k = 0; k < 11; k++) w[k] |= (j >> 2) == k ? b : 0u;
        }
        v0 = pp_v4u{w[0], w[1], w[2], w[3]};
        v1 = pp_v4u{w[4], w[5], w[6], w[7]};
        v2 = pp_v3u{w[8], w[9], w[10]};
      }
      uint32_t *d = reinterpret_cast<uint32_t *>(smem + lb);
      d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
      d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
      d[8] = v2.x; d[9] = v2.y; d[10] = v2.z;
    }
    // the document boundaries around the piece: dn = first boundary at or behind g0, nbr its offset from g0 (rare: a
    // boundary inside the tile -- the lanes look their own up; documents of a few bytes make the call another engine's)
    bool give_up = false;
    uint64_t dn = tile_dn[tile];
    int64_t nb = (int64_t)M.doc_off[dn];
    if (wany(nb < t0 + kPpTile + 16)) {
      int guard = 0;
      while (nb < g0 && guard < 64) {
        dn++;
        nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
        guard++;
      }
      if (nb < g0) give_up = true;  // (more than 64 documents in front of a piece of its tile)
    }
    int64_t dstart = dn > 0 ? (int64_t)M.doc_off[dn - 1] : 0;  // the document that holds the byte before nb
    asm volatile("" : "+v"(nb), "+v"(dstart));
    uint32_t nbr = (uint32_t)min<int64_t>(nb - g0, 1 << 20);      // offset of the next boundary from g0
    const uint32_t nr = (uint32_t)min<int64_t>(max<int64_t>(N - g0, 0), 1 << 20);
    uint32_t bpos = ~0u;            // the boundary inside the lane's range that its events have seen (offset), none yet
    uint64_t dn_at_bpos = 0;        // ... its index
    const int64_t dstart0 = dstart;  // ... the document before it
    uint32_t o = 0, po1 = kPpPiece, gp = 0, c1 = 0;
    uint64_t evmask = 0, deepmask = 0;  // end offsets (from g0) of the lane's events / of the pairs with empty records behind
    uint32_t startmask = 0;             // ... and where those pairs start (the k-th set bit belongs to deepmask's k-th)
    uint32_t head = 0xFFFFu, tail = 0xFFFFu;  // the lane's chain in the pool
    if (lane == 0) *pcnt = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t hsum = 0;
    // A loop trip takes eight units: eight steps that only ASK -- unit, hash, filter word, and every lane's request for a table
    // entry (the lanes without a pair ask for slot 0: a load under a condition is waited for where the branches meet) -- and,
    // behind them, eight looks at what came back.  A request takes ~2 us to come back (the slowest of a wave's scattered lanes
    // decides, and requests retire in order): looking at an entry one or two steps after asking for it left the kernel waiting
    // in every step (6.4 ms per GiB); asked for eight deep it is 1.4 ms (profiles/r06_pair_engine.txt).  No request in flight
    // crosses the loop's back edge (hipcc copies such registers at the loop head before the data is there).
    constexpr int kSteps = 8;
    bool first = true;
    // the unit at offset oo: its bits (8, 16, 24) and raw bytes; a lead byte without its continuation bytes, or a unit that
    // does not lie inside its document, is a one-byte unit
    auto unit_at = [&](uint32_t oo, uint32_t &s, uint32_t &c) {
      const uint32_t at = min(oo, (uint32_t)(kPpRow - 5));
      const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + lb + (at & ~3u));
      const uint32_t x = __builtin_amdgcn_alignbyte(q[1], q[0], at & 3u);
      const uint32_t b0 = x & 0xFFu;
      s = (b0 & 0xE0u) == 0xC0u ? 16u : ((b0 & 0xF0u) == 0xE0u ? 24u : 8u);
      const uint32_t cm = s == 24u ? 0xC0C000u : (s == 16u ? 0xC000u : 0u);
      s = ((x & cm) == (cm & 0x808080u) && oo + (s >> 3) <= min(nbr, nr)) ? s : 8u;
      c = __builtin_amdgcn_ubfe(x, 0u, s);
    };
    // a document starts at the unit at o: no pair across it
    auto boundary = [&](bool bnd) {
      if (wany(bnd)) {
        if (bnd) {
          if (bpos != ~0u) give_up = true;  // (a second boundary in one piece: documents of a few bytes)
          bpos = o;
          dn_at_bpos = dn;
          int guard = 0;
          do {
            dn++;
            nb = dn <= D ? (int64_t)M.doc_off[dn] : INT64_MAX;
            guard++;
          } while (nb == g0 + (int64_t)o && guard < 64);
          if (nb == g0 + (int64_t)o) give_up = true;
          asm volatile("" : "+v"(nb));
          nbr = (uint32_t)min<int64_t>(nb - g0, 1 << 20);
        }
      }
    };
    for (;;) {
      if (!wany(first | po1 < (uint32_t)kPpPiece)) break;
      first = false;
      uint32_t uc[kSteps + 1];  // the units' raw bytes
      uint32_t um[kSteps + 1];  // where a unit starts | "a document starts there" << 8 | (the pair that ends with it) "passed the filter" << 9
      uint4 ent[kSteps];
      const uint32_t carry_c = c1, carry_start = po1;
#pragma unroll
      for (int j = 0; j < kSteps; j++) {
        const bool bnd = o == nbr;
        boundary(bnd);
        uint32_t s, c;
        unit_at(o, s, c);
        if (bnd) po1 = kPpPiece;
        uint32_t h = __umul24(c, P.k1) + gp;
        const uint32_t g = __umul24(c, kSkipKB);
        gp = __builtin_amdgcn_alignbit(g, g, 11);
        h ^= h >> 16;
        const uint32_t w = bl[h >> wshift];
        const uint32_t m = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
        const bool pend = po1 < (uint32_t)kPpPiece & (w & m) == m;
        uc[j] = c;
        um[j] = o | (bnd ? 256u : 0u) | (pend ? 512u : 0u);
        {
          const uint32_t dd = dsp[(h >> 7) & (P.groups - 1u)];
          const uint32_t t = h * kPairMix;
          if (AHA_PP_LAB & 1) ent[j] = make_uint4(t, dd, 0, 0);
          else if (AHA_PP_LAB & 32) ent[j] = P.tab[pend ? (((t >> tshift) + dd * ((t << 1) | 1u)) & tmask) : (uint32_t)lane];
          else if (AHA_PP_LAB & 256) {
            const uint32_t sl = pend ? (((t >> tshift) + dd * ((t << 1) | 1u)) & tmask) : 0u;
            const uint2 e2 = *reinterpret_cast<const uint2 *>(P.tab + sl);
            ent[j] = make_uint4(e2.x, e2.y, sl, 0u);
          } else if (AHA_PP_LAB & 128) {
            const uint2 *ep = reinterpret_cast<const uint2 *>(P.tab + (pend ? (((t >> tshift) + dd * ((t << 1) | 1u)) & tmask) : 0u));
            const uint2 e2 = ep[0], e3 = ep[1];
            ent[j] = make_uint4(e2.x, e2.y, e3.x, e3.y);
          } else if (AHA_PP_LAB & 64) {
            const uint2 e2 = *reinterpret_cast<const uint2 *>(P.tab + (pend ? (((t >> tshift) + dd * ((t << 1) | 1u)) & tmask) : 0u));
            ent[j] = make_uint4(e2.x, e2.y, 0u, 0u);
          } else ent[j] = P.tab[pend ? (((t >> tshift) + dd * ((t << 1) | 1u)) & tmask) : 0u];
        }
        po1 = o;
        c1 = c;
        o += s >> 3;
      }
      {  // a peek at the unit behind the trip's last one: the third unit of its last pair (the next trip takes it)
        uint32_t s, c;
        unit_at(o, s, c);
        uc[kSteps] = c;
        um[kSteps] = o | (o == nbr ? 256u : 0u);
      }
#pragma unroll
      for (int j = 0; j < kSteps; j++) {  // the pair (unit j - 1, unit j): it ends where unit j + 1 starts
        const uint32_t c0 = j ? uc[j - 1] : carry_c, start = j ? (um[j - 1] & 255u) : carry_start;
        const uint32_t end = um[j + 1] & 255u;
        if ((um[j] & 512u) && (ent[j].x & 0xFFFFFFu) == c0 && ent[j].y == uc[j]) {
          if (AHA_PP_LAB & 256) {  // (lab: the payload and the child filter in a second load, for the pairs that matched)
            const uint2 zw = reinterpret_cast<const uint2 *>(P.tab + ent[j].z)[1];
            ent[j].z = zw.x;
            ent[j].w = zw.y;
          }
          if (ent[j].z != 0u) {  // an END state: an event
            // (the hits the event stands for: the payload's top byte -- a handle with an output chain of 255 keys or more does
            // not get this engine, and a lookup here would put a `vmcnt(0)` behind the branch)
            hsum += ent[j].z >> 24;
            const uint32_t slot = atomicAdd(pcnt, 1u);
            if (slot < (uint32_t)kPpPool) {
              px[slot] = ent[j].z;
              if (tail != 0xFFFFu) pn[tail] = (uint16_t)slot;
              else head = slot;
              tail = slot;
              evmask |= 1ull << end;
            } else {
              give_up = true;
            }
          }
          const uint32_t cls = pt_cls(uc[j + 1]);
          if (!(um[j + 1] & 256u) && (ent[j].w & cls) == cls) {  // the third unit may continue the pair: a deep candidate
            deepmask |= 1ull << end;
            startmask |= 1u << start;
          }
        }
      }
    }
    // ---- the tile's records, in position order: lane by lane, and in a lane by end offset -- an event, then the three
    // empty records of a deep candidate at the same offset
    const uint32_t n_ev = (uint32_t)__popcll(evmask), n_dc = (uint32_t)__popcll(deepmask);
    const uint32_t mine = n_ev + kPpPlace * n_dc;
    const uint32_t incl = wave_incl_scan(mine);
    const uint32_t total = wave_last(incl);
    const uint32_t dincl = wave_incl_scan(n_dc);
    const uint32_t dtotal = wave_last(dincl);
    const uint32_t cbase = (uint32_t)wave_id * seg_cap + cused;  // the tile's place in the list of candidates
    if (cused + dtotal > seg_cap) give_up = true;
    else cused += dtotal;
    if (lane == 0) {
      M.ev_cnt[tile] = total;
      M.lead_cnt[tile] = dtotal ? cbase : 0u;       // (byte offsets: the lead-byte arrays are free -- the tile's candidates: first,
      M.chunk_doc0[tile] = dtotal;                  //  number)
      if (total > stride) M.cursor[1] = 2ull;       // region full: the host repeats the call with larger regions
    }
    hsum = wave_last(wave_incl_scan(hsum));
    if (lane == 0) M.chunk_hits[tile] = hsum;
    if (wany(give_up)) {
      if (give_up) M.cursor[1] = 3ull;  // not this engine's batch: the character-level traversal takes the call
      continue;
    }
    if (AHA_PP_LAB & 4) continue;
    uint2 *reg = M.evd + tile * (uint64_t)stride;
    uint32_t at_rec = incl - mine, at_c = cbase + dincl - n_dc;
    uint64_t both = evmask | deepmask;
    const int32_t drel0 = (int32_t)(g0 - dstart0);  // end offset in the document = drel0 + offset (in front of the boundary)
    while (wany(both != 0ull)) {
      if (both) {
        const uint32_t e = (uint32_t)__builtin_ctzll(both);
        both &= both - 1;
        const uint32_t y = (bpos != ~0u && e > bpos) ? e - bpos : (uint32_t)(drel0 + (int32_t)e);
        if ((evmask >> e) & 1ull) {
          if (at_rec < stride) reg[at_rec] = make_uint2(px[head], y);
          at_rec++;
          head = pn[head];
        }
        if ((deepmask >> e) & 1ull) {
          // the candidate starts two units in front of e: kp_walk finds the units again; here: where, and which records
          {
            const int64_t ds = (bpos != ~0u && e > bpos) ? g0 + (int64_t)bpos : dstart0;
            const uint32_t st = (uint32_t)__builtin_ctz(startmask);
            // (the candidate's document ends at the boundary the lane last looked up -- or, for a candidate in front of the
            // boundary inside the piece, at that boundary)
            const bool before = bpos != ~0u && e <= bpos;
            const int64_t de = before ? g0 + (int64_t)bpos : nb;
            const uint64_t dnx = before ? dn_at_bpos : dn;
            cand[at_c] = PpCand{(uint32_t)tile, ((uint32_t)(g0 - t0) + st) | ((uint32_t)(g0 - t0) + e) << 12, at_rec,
                                (uint32_t)((ds - t0) + (1ll << 31)), (uint32_t)min<int64_t>(de - t0, 1 << 20), (uint32_t)dnx};
          }
          startmask &= startmask - 1u;
          at_c++;
#pragma unroll
          for (uint32_t j = 0; j < kPpPlace; j++) {
            if (at_rec < stride) reg[at_rec] = make_uint2(kPpNull, y);
            at_rec++;
          }
        }
      }
    }
    // documents that start inside the piece: the records in front of them (k2d_doc_offsets)
    if (wany(bpos != ~0u && bpos < (uint32_t)kPpPiece)) {
      if (bpos != ~0u && bpos < (uint32_t)kPpPiece) {
        const uint64_t below = (2ull << bpos) - 1ull;  // end offsets <= bpos
        const uint32_t rank = incl - mine + (uint32_t)__popcll(evmask & below) + kPpPlace * (uint32_t)__popcll(deepmask & below);
        const int64_t b = g0 + (int64_t)bpos;
        // (dn has moved behind the boundary: the documents that start at b are the ones in front of dn that start there)
        for (uint64_t d = dn; d-- > 0 && (int64_t)M.doc_off[d] == b;) M.doc_ev_rank[d] = rank;
      }
    }
  }
  if (lane == 0) seg_cnt[wave_id] = cused;
}

// the unit at byte `at` (< 8) of the sixteen bytes w: scan_unit.hip's decode over registers
struct PpTabs {
  const uint4 *t0a;
  const uint2 *t0b;
  const uint8_t *tabb;
};
__device__ __forceinline__ void pp_decode(const PpTabs &T, const pp_v4u &w, uint32_t at, int32_t dend, uint32_t &o_code, uint32_t &o_L) {
  const uint32_t lo = at < 4u ? w.x : (at < 8u ? w.y : w.z), hi = at < 4u ? w.y : (at < 8u ? w.z : w.w);
  const uint32_t w4 = __builtin_amdgcn_alignbyte(hi, lo, at & 3u);
  const uint32_t b0 = w4 & 0xFFu;
  const uint4 q0 = T.t0a[b0];
  const uint2 q1 = T.t0b[b0];
  const uint32_t s1 = *reinterpret_cast<const uint32_t *>(T.tabb + q0.x + ((w4 >> 6) & 0x3FCu));
  const uint32_t s2 = *reinterpret_cast<const uint32_t *>(T.tabb + q0.y + ((w4 >> 14) & 0x3FCu));
  const uint32_t sum = q0.z + s1 + s2;
  const uint32_t want = q1.y;
  const bool in_doc = (int32_t)(at + want) <= dend;
  const bool whole = in_doc & sum < kUPoison;
  const bool good = whole & (sum - q0.w) < q1.x;
  o_L = whole ? want : 1u;
  o_code = good ? sum - kUBias : 0u;
}

// 16 text bytes from any byte address; bytes beyond the text read as 0
__device__ __forceinline__ pp_v4u pp_text16(const uint8_t *__restrict__ text, int64_t g, int64_t N) {
  pp_v4u v = {0u, 0u, 0u, 0u};
  if (g + 16 <= N) {
    __builtin_memcpy(&v, text + g, 16);
  } else if (g < N) {
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
    v = pp_v4u{w0, w1, w2, w3};
  }
  return v;
}

// ---- the deep walks: a lane per candidate, the goto walk of its start over the unit image (22-bit bases)
__global__ __launch_bounds__(kV2Threads) void kp_walk(UnitDev U, V2Args M, const PpCand *__restrict__ cand, PpWalk *__restrict__ wres,
                                                      uint32_t seg_log2, uint32_t n_seg, const uint32_t *__restrict__ seg_cnt) {
  const uint32_t seg_cap = 1u << seg_log2;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1]) return;
  constexpr int BB = 22;
  uint32_t *tabw = reinterpret_cast<uint32_t *>(smem);
  uint32_t *rlw = tabw + kUTabWords;
  const uint32_t n_root = (U.n_syms + 3u) & ~3u;
  for (uint32_t i = threadIdx.x; i < kUTabWords; i += kV2Threads) tabw[i] = U.tables[i];
  for (uint32_t i = threadIdx.x; i < n_root; i += kV2Threads) rlw[i] = i < U.n_syms ? U.root[i] : 0u;
  __syncthreads();
  const PpTabs T{reinterpret_cast<const uint4 *>(tabw + kUT0a), reinterpret_cast<const uint2 *>(tabw + kUT0b), reinterpret_cast<const uint8_t *>(tabw)};
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t n_slots = (uint64_t)n_seg * seg_cap;
  for (uint64_t i = (uint64_t)blockIdx.x * kV2Threads + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * kV2Threads) {
    if ((uint32_t)(i & (seg_cap - 1u)) >= seg_cnt[i >> seg_log2]) continue;  // (a wave's 64 slots lie in one segment: all or none, mostly)
    const PpCand c = cand[i];
    const int64_t t0 = (int64_t)c.tile * kPpTile;
    const int64_t pend = t0 + (int64_t)(c.start_end >> 12);
    const int64_t start = t0 + (int64_t)(c.start_end & 0xFFFu);
    const uint64_t dn = c.next_doc;                      // the document's end: kp_pairs has looked it up
    const int64_t dend = t0 + (int64_t)c.dend_lo;
    PpWalk r;
    r.reach = (uint32_t)(pend - start);
    r.ends = 0;
    r.next_doc = (uint32_t)dn;
#pragma unroll
    for (uint32_t j = 0; j < kPpPlace; j++) r.end_base[j] = 0;
    uint32_t n_ends = 0;
    int64_t p = start;
    uint32_t E = 0, depth = 0;
    bool alive = true, over = false;
    while (alive) {
      const pp_v4u w = pp_text16(M.text, p, N);
      uint32_t off = 0;
      // the units of this window (at most three: offsets up to 6 decode from sixteen bytes)
      for (int u = 0; u < 3 && alive; u++) {
        uint32_t code, L;
        pp_decode(T, w, off, (int32_t)min<int64_t>(dend - p, 1 << 20), code, L);
        if (code == 0u) {
          alive = false;
          break;
        }
        uint32_t nE = 0;
        if (E == 0u) {
          nE = rlw[code];
        } else {
          const uint32_t Bq = u_child(E, BB);
          if (Bq >= U.big_lo && code >= U.n_low) {  // a big state's group record, then the child's own slot
            const uint2 gr = U.slots[Bq + U.g0 + (code >> 5)];
            if ((gr.x >> (code & 31u)) & 1u) {
              const uint2 en = U.slots[gr.y + __builtin_popcount(gr.x & ~(~0u << (code & 31u)))];
              nE = en.x;
            }
          } else if ((((E | 0x20000000u) >> ((uint32_t)BB + (code & 7u))) & 1u) != 0u) {
            const uint2 en = U.slots[Bq ^ code];
            if (u_sym(en.y) == code) nE = en.x;
          }
        }
        if (nE == 0u) {
          alive = false;
          break;
        }
        E = nE;
        off += L;
        depth++;
        if (depth >= 3u) {
          r.reach = (uint32_t)(p + (int64_t)off - start);
          if (u_end(E)) {
            if (n_ends < kPpPlace) {
              r.ends |= (r.reach & 255u) << (8u * (n_ends + 1u));
              const uint32_t bs = u_child(E, BB);  // (selects, not an indexed store: the struct stays in registers)
              r.end_base[0] = n_ends == 0u ? bs : r.end_base[0];
              r.end_base[1] = n_ends == 1u ? bs : r.end_base[1];
              r.end_base[2] = n_ends == 2u ? bs : r.end_base[2];
            } else {
              over = true;
            }
            n_ends++;
          }
        }
      }
      p += (int64_t)off;
      if (p >= dend) alive = false;
    }
    if (over) M.cursor[1] = 3ull;  // (more END states on one path than a candidate has records: compile rules this out)
    r.ends |= min(n_ends, kPpPlace);
    wres[i] = r;
  }
}

// ---- the records a deep walk covers become empty: everything that ends behind its pair and at or before its reach
__global__ __launch_bounds__(256) void kp_void(DevAut A, V2Args M, const PpCand *__restrict__ cand, const PpWalk *__restrict__ wres,
                                               uint32_t seg_log2, uint32_t n_seg, const uint32_t *__restrict__ seg_cnt) {
  const uint32_t seg_cap = 1u << seg_log2;
  if (M.cursor[1]) return;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)n_seg * seg_cap || (uint32_t)(i & (seg_cap - 1u)) >= seg_cnt[i >> seg_log2]) return;
  const PpCand c = cand[i];
  const PpWalk r = wres[i];
  const int64_t t0 = (int64_t)c.tile * kPpTile;
  const int64_t pend = t0 + (int64_t)(c.start_end >> 12);
  const int64_t start = t0 + (int64_t)(c.start_end & 0xFFFu);
  const int64_t reach = start + (int64_t)r.reach;
  if (reach <= pend) return;  // the third unit did not continue the pair after all
  const int64_t dstart = t0 + ((int64_t)c.dstart_lo - (1ll << 31));
  const uint32_t y_lo = (uint32_t)(pend - dstart), y_hi = (uint32_t)(reach - dstart);  // end offsets in (y_lo, y_hi]
  const int64_t dend = (int64_t)M.doc_off[r.next_doc];
  const uint32_t stride = M.ev_stride;
  for (uint64_t tile = c.tile; tile < M.n_chunks && (int64_t)tile * kPpTile < reach; tile++) {
    // the records of this tile that belong to the walk's document: in front of the next document's first record, if it starts here
    uint32_t n = min(M.ev_cnt[tile], stride);
    if (dend < (int64_t)(tile + 1) * kPpTile && dend >= (int64_t)tile * kPpTile && r.next_doc <= M.n_docs && dend < (int64_t)M.n_bytes)
      n = min(n, M.doc_ev_rank[r.next_doc]);
    uint2 *reg = M.evd + tile * (uint64_t)stride;
    uint32_t removed = 0;
    for (uint32_t j = tile == c.tile ? c.slot : 0u; j < n; j++) {
      const uint32_t y = reg[j].y;
      if (y > y_hi) break;
      if (y <= y_lo) continue;
      const uint32_t old = atomicExch(&reg[j].x, kPpNull);
      uint32_t cnt = old >> 24;
      if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[old & 0xFFFFFFu].y : (old & 0xFFFFFFu)];
      removed += cnt;
    }
    if (removed) atomicSub(&M.chunk_hits[tile], removed);
  }
}

// ---- a walk's own END events, those behind the reach of every earlier walk, into the records behind its pair
__global__ __launch_bounds__(256) void kp_fill(DevAut A, V2Args M, const PpCand *__restrict__ cand, const PpWalk *__restrict__ wres,
                                               uint32_t max_len, uint32_t seg_log2, uint32_t n_seg, const uint32_t *__restrict__ seg_cnt) {
  const uint32_t seg_cap = 1u << seg_log2;
  if (M.cursor[1]) return;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (uint64_t)n_seg * seg_cap || (uint32_t)(i & (seg_cap - 1u)) >= seg_cnt[i >> seg_log2]) return;
  const PpCand c = cand[i];
  const PpWalk r = wres[i];
  const int64_t t0 = (int64_t)c.tile * kPpTile;
  const int64_t pend = t0 + (int64_t)(c.start_end >> 12);
  const int64_t start = t0 + (int64_t)(c.start_end & 0xFFFu);
  const uint32_t n_ends = r.ends & 255u;
  if (start + (int64_t)r.reach <= pend || n_ends == 0) return;
  const int64_t dstart = t0 + ((int64_t)c.dstart_lo - (1ll << 31));
  // the furthest reach of the walks that started before this one (they start at most max_len bytes in front of it): the
  // tile's own candidates in front of i, then the tiles before it
  int64_t R = 0;
  {
    const uint32_t first = M.lead_cnt[c.tile];
    for (uint64_t j = i; j-- > first;) {
      const PpCand cj = cand[j];
      if (t0 + (int64_t)(cj.start_end & 0xFFFu) < start - (int64_t)max_len) break;
      const int64_t sj = t0 + (int64_t)(cj.start_end & 0xFFFu), rj = sj + (int64_t)wres[j].reach;
      R = max(R, rj > t0 + (int64_t)(cj.start_end >> 12) ? rj : 0);
    }
    for (uint64_t tile = c.tile; tile-- > 0 && (int64_t)(tile + 1) * kPpTile > start - (int64_t)max_len;) {
      const uint32_t f = M.lead_cnt[tile], nn = M.chunk_doc0[tile];
      const int64_t tt = (int64_t)tile * kPpTile;
      for (uint32_t j = nn; j-- > 0;) {
        const PpCand cj = cand[(uint64_t)f + j];
        if (tt + (int64_t)(cj.start_end & 0xFFFu) < start - (int64_t)max_len) break;
        const int64_t rj = tt + (int64_t)(cj.start_end & 0xFFFu) + (int64_t)wres[(uint64_t)f + j].reach;
        R = max(R, rj > tt + (int64_t)(cj.start_end >> 12) ? rj : 0);
      }
    }
  }
  uint2 *reg = M.evd + (uint64_t)c.tile * M.ev_stride;
  uint32_t k = 0, added = 0;
#pragma unroll
  for (uint32_t j = 0; j < kPpPlace; j++) {
    if (j >= n_ends) break;
    const int64_t e = start + (int64_t)((r.ends >> (8u * (j + 1u))) & 255u);
    if (e <= R) continue;
    const uint32_t x = A.end_info[r.end_base[j]];
    uint32_t cnt = x >> 24;
    if (cnt == 255u) cnt = A.key_cnt[A.chain ? A.chain[x & 0xFFFFFFu].y : (x & 0xFFFFFFu)];
    if (c.slot + k < M.ev_stride) reg[c.slot + k] = make_uint2(x, (uint32_t)(e - dstart));
    k++;
    added += cnt;
  }
  if (added) atomicAdd(&M.chunk_hits[c.tile], added);
}

}  // namespace

size_t pair_cand_bytes(uint64_t cap) { return cap * sizeof(PpCand); }
size_t pair_walk_bytes(uint64_t cap) { return cap * sizeof(PpWalk) + (1u << 20); }  // (+ the segments' fill counts)
uint32_t pair_tile_bytes() { return kPpTile; }

int pair_prepare(uint32_t n_syms, uint32_t bloom_log2, uint32_t groups) {
  int e = (int)hipFuncSetAttribute((const void *)kp_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pp_lds(bloom_log2, groups));
  const size_t wl = (size_t)(kUTabWords + ((n_syms + 3u) & ~3u)) * 4 + 16;
  if (!e) e = (int)hipFuncSetAttribute((const void *)kp_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
  return e;
}

void pair_launch(const PairDev &P, const UnitDev &U, const DevAut &A, const V2Args &M, void *tile_dn, void *cand, void *wres,
                 uint64_t cand_cap, uint32_t grid, void *mid_event, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  // the candidates' list: a segment per wave of kp_pairs, the segments' fill counts in front of the walks' results
  const uint32_t n_seg = grid * (kV2Threads / 64);
  uint32_t seg_log2 = 8;  // (a power of two: the kernels find a slot's segment with a shift)
  while (seg_log2 < 20 && ((uint64_t)2 << seg_log2) * n_seg <= cand_cap) seg_log2++;
  const uint32_t seg_cap = 1u << seg_log2;
  uint32_t *seg_cnt = reinterpret_cast<uint32_t *>(wres);
  PpWalk *wr = reinterpret_cast<PpWalk *>(reinterpret_cast<uint8_t *>(wres) + (((size_t)n_seg * 4 + 255) & ~(size_t)255));
  hipLaunchKernelGGL(kp_tile_doc, dim3((uint32_t)((M.n_chunks + 255) / 256)), dim3(256), 0, s, M, (uint32_t *)tile_dn);
  hipLaunchKernelGGL(kp_pairs, dim3(grid), dim3(kV2Threads), pp_lds(P.bloom_log2, P.groups), s, P, A, M, (const uint32_t *)tile_dn,
                     (PpCand *)cand, seg_cap, seg_cnt);
  if (mid_event) (void)hipEventRecord((hipEvent_t)mid_event, s);
  const size_t wl = (size_t)(kUTabWords + ((U.n_syms + 3u) & ~3u)) * 4 + 16;
  hipLaunchKernelGGL(kp_walk, dim3(grid), dim3(kV2Threads), wl, s, U, M, (const PpCand *)cand, wr, seg_log2, n_seg, (const uint32_t *)seg_cnt);
  const uint32_t gb = (uint32_t)(((uint64_t)n_seg * seg_cap + 255) / 256);
  hipLaunchKernelGGL(kp_void, dim3(gb), dim3(256), 0, s, A, M, (const PpCand *)cand, (const PpWalk *)wr, seg_log2, n_seg, (const uint32_t *)seg_cnt);
  hipLaunchKernelGGL(kp_fill, dim3(gb), dim3(256), 0, s, A, M, (const PpCand *)cand, (const PpWalk *)wr, U.max_len, seg_log2, n_seg,
                     (const uint32_t *)seg_cnt);
}

}  // namespace aha
