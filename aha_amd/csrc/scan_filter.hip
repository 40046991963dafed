// scan_filter.hip -- the PREFIX-FILTER engine for gfx950: byte-level Aho-Corasick for key sets and text where few positions
// can start a key at all (keyword lists over logs: BASELINE config 2).  Replaces src/aha/ac.cr:176-192 (match_) for calls
// with byte offsets and no separator filter on handles whose keys are 3 .. 64 bytes long (capi.cpp aha_ac_compile_packed).
//
// The reference's state after byte j is the LONGEST suffix of the text that is a trie path (ac.cr:176-192), i.e. the goto
// walk of the EARLIEST start that is still alive at j.  So instead of carrying a state through every byte:
//   kf_filter   every byte position asks a blocked Bloom filter in LDS (4 .. 64 KiB) whether its next D bytes (D = min(4, shortest
//               key)) are the first D bytes of some key: one bit per position.  Stateless, position-parallel, coalesced.
//   kf_walk     a wave takes a chunk of 4 .. 32 KiB: the candidate bits of the chunk (and of the kfWarm bytes before it: walks that
//               reach into the chunk) become batches of 64 candidates in position order, a lane walks ONE candidate's goto
//               path through the byte-level image (no fail links, no state between candidates); an END step is the
//               reference's event exactly when no earlier start's walk is still alive at its end -- the exclusive prefix
//               maximum of the walks' reaches -- and then, in (start, step) order, the events are already in position order.
//               A start the filter rejects dies within D - 1 bytes: it ends nothing (keys are at least D bytes long) and
//               outlives no later start's END (which lies at least D bytes behind that start).
// The events leave in the byte-level engine's region format ({END state, end offset in the document} at
// evd[chunk * ev_stride + seq], ev_cnt, doc_ev_rank), so count / scan / expansion / document offsets are scan_v2.hip's
// (v2_launch_direct_post).  Candidates are what the text makes of the key set: a chunk with more than one per ~20
// bytes (or a walk through more than kfMaxEnds nested keys) makes kf_walk give the whole call to the other engines
// (cursor[1] = 3; capi.cpp repeats it there), and capi.cpp builds no filter that would be more than a quarter full.
// (No counter of the batch's candidates: thousands of atomics on one address cost kf_filter 40..85 us a launch.)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"

namespace aha {

namespace {

constexpr int kfWarm = 64, kfAhead = 64;
constexpr int kfMaxWords = 8;      // bitmap words a lane of kf_walk takes: chunks of up to 32 KiB
constexpr int kfListPer4K = 256;   // candidates a chunk's list holds per 4 KiB
constexpr int kfDensePer4K = 208;  // more than this (one per ~20 bytes) and the batch is not this engine's: the byte-level engine
                                   // costs what filter + walks cost at about 5.5 % (tools/lab/f4.sh: 30 000 keys over letters, 6.5 %,
                                   // tie; 10 000 keys, 2.2 %, 1.7 x faster here)
constexpr int kfMaxEnds = 4;  // END steps a walk keeps; a walk with more hands the call to the other engines

// LDS of a wave of kf_walk with chunks of W * 4 KiB: its candidate list (2 bytes an entry), the document boundaries near
// its chunk, the END steps of a batch's walks
// (+ for calls with char offsets the chunk's continuation bytes: a 64-bit mask and a running count per 64 text bytes)
__host__ __device__ constexpr uint32_t kf_wave_lds(uint32_t W, bool chars = false) {
  return kfListPer4K * W * 2 + 64 * 4 + 64 * kfMaxEnds * 8 + (chars ? W * 64 * (8 + 2) : 0);
}

// per chunk: the first d with doc_off[d] >= chunk start and the boundary before it (kf_walk reads one record per chunk, a
// chunk ahead, instead of searching).  Written by the blocks of kf_filter's launch that stand behind the filter's own.
struct KfChunk {
  uint64_t dn;
  int64_t b_prev;  // doc_off[dn - 1] < chunk start (the start of the document that holds the byte before the chunk), or 0
};
__device__ __forceinline__ void kf_chunk_doc(const V2Args &M, KfChunk *rec, uint64_t c) {
  if (c >= M.n_chunks || M.cursor[1] >= 16ull) return;
  const uint64_t dn = first_boundary(M.doc_off, M.n_docs, c * (uint64_t)M.S);
  rec[c] = KfChunk{dn, dn > 0 ? (int64_t)M.doc_off[dn - 1] : 0};
}

// ---- the filter: bit p of the bitmap <=> text[p .. p + D) may start a key.  An entry of the filter: the product
// w * 0x9E3779B1 of the D bytes (little endian, D < 4: the upper bytes masked off) selects a word with its top log2 bits and
// two bits of that word with the ten bits below (capi.cpp sets them: filter_entry) -- no fold of the product: its upper half is
// where a multiplicative hash has mixed all of w's bytes, and the word index carries the discrimination.
template <bool D4>
__global__ __launch_bounds__(1024) void kf_filter(FilterDev F, const uint8_t *__restrict__ text, uint64_t n_bytes,
                                                   uint16_t *__restrict__ bitmap, unsigned long long *non_ascii, V2Args M,
                                                   KfChunk *chunk_rec, uint32_t filter_blocks) {
  __shared__ uint32_t bl[1 << kFilterLog2];
  if (blockIdx.x >= filter_blocks) {  // (the chunk records ride on this launch: a kernel of their own costs 5 us of a 64 MiB call)
    kf_chunk_doc(M, chunk_rec, (uint64_t)(blockIdx.x - filter_blocks) * 1024 + threadIdx.x);
    return;
  }
  for (uint32_t i = threadIdx.x; i < (1u << F.log2); i += 1024) bl[i] = F.bloom[i];
  const uint32_t hs = 32u - F.log2;
  __syncthreads();
  const uint32_t dmask = F.d >= 4 ? 0xFFFFFFFFu : ((1u << (8 * F.d)) - 1u);
  const uint64_t n_pieces = ((n_bytes + 63) / 64) * 4;  // whole 64-bit words of the bitmap (pieces beyond the text: no bit)
  uint32_t seen = 0;  // OR of the bytes this lane looked at (a call with char offsets: is the batch plain ASCII?)
  for (uint64_t p = (uint64_t)blockIdx.x * 1024 + threadIdx.x; p < n_pieces; p += (uint64_t)filter_blocks * 1024) {
    const uint64_t g = p * 16;
    uint32_t d[5] = {0, 0, 0, 0, 0};  // 16 bytes + 3 of look-ahead (bytes beyond the text read as 0: no key holds a NUL)
    if (g >= n_bytes) {
      bitmap[p] = 0;
      continue;
    }
    if (g + 20 <= n_bytes) {
      const uint4 v = *reinterpret_cast<const uint4 *>(text + g);
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
      d[4] = *reinterpret_cast<const uint32_t *>(text + g + 16);
    } else {
      for (int j = 0; j < 20 && g + j < n_bytes; j++) d[j >> 2] |= (uint32_t)text[g + j] << ((j & 3) * 8);
    }
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      uint32_t w = (k & 3) ? __builtin_amdgcn_alignbyte(d[(k >> 2) + 1], d[k >> 2], (uint32_t)(k & 3)) : d[k >> 2];
      if (!D4) w &= dmask;
      const uint32_t h = w * kFilterMul;
      const uint32_t word = bl[h >> hs];
      bits |= ((word >> ((h >> (hs - 5u)) & 31u)) & (word >> ((h >> (hs - 10u)) & 31u)) & 1u) << k;  // (the masks fold into the shifts)
    }
    bitmap[p] = (uint16_t)bits;
    seen |= d[0] | d[1] | d[2] | d[3];
  }
  // (a plain store, not an atomic: every wave that has something to say says the same)
  if (non_ascii && (seen & 0x80808080u)) *non_ascii = 1ull;
}

typedef uint32_t kf_v4u __attribute__((ext_vector_type(4)));

// 16 text bytes from any byte address (gfx950 global loads take unaligned addresses); bytes beyond the text read as 0
__device__ __forceinline__ kf_v4u kf_text16(const uint8_t *__restrict__ text, int64_t g, int64_t N) {
  kf_v4u v = {0u, 0u, 0u, 0u};
  if (g + 16 <= N) {
    __builtin_memcpy(&v, text + g, 16);
  } else {  // (the last bytes of the batch)
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
    v = kf_v4u{w0, w1, w2, w3};
  }
  return v;
}

// ---- the walks.  A wave takes a chunk of M.S bytes (a multiple of 4 KiB: kfWords * 64 words of the bitmap, lane l the
// kfWords consecutive words l * W ..): offsets below are relative to chunk start - kfWarm.  IMG: the whole image in LDS
// (a keyword list's automaton is a few tens of KiB), the block's 16 waves share it; otherwise the slots come through L1/L2.
// CHARS (String overload, matcher.cr:34-39): the events carry the lead-byte count of their end (<< 1 | "counted from the
// document's start") in place of the byte offset, the chunk leaves lead_cnt / chunk_doc0 / doc_lead_rank -- what k2_traverse
// leaves for k2d_expand<.., true>.  A character = a byte that is no continuation byte (10xxxxxx), as there.  While the batch
// is plain ASCII (kf_filter has looked at every byte: *non_ascii) the count IS the offset; otherwise the wave reads its chunk
// once more, coalesced, and keeps the continuation bytes' mask and running count per 64 bytes in LDS.
template <bool IMG, bool CHARS>
__global__ __launch_bounds__(IMG ? 1024 : 256) void kf_walk(DevAut A, V2Args M, const unsigned long long *__restrict__ bitmap,
                                                             const KfChunk *__restrict__ chunk_rec,
                                                             const unsigned long long *non_ascii) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // the doc offsets are not what the call says (k_check_docs ran in front): index nothing -- and say nothing: every store to
  // cursor[1] below is a plain one, and the verdict must reach the host (block-uniform, like the hand-back under it)
  if (M.cursor[1] >= 16ull) return;
  const bool asc = !CHARS || !*non_ascii;  // (block-uniform)
  constexpr int WPB = IMG ? 16 : 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t W = M.S / 4096u;  // bitmap words per lane (1, 2, 4 or 8)
  const uint32_t list_cap = kfDensePer4K * W;
  uint8_t *mine = smem + (size_t)wave * kf_wave_lds(W, CHARS);
  uint16_t *list = reinterpret_cast<uint16_t *>(mine);
  uint32_t *bnd = reinterpret_cast<uint32_t *>(mine + kfListPer4K * W * 2);
  uint2 *ends = reinterpret_cast<uint2 *>(mine + kfListPer4K * W * 2 + 256);
  unsigned long long *cmask = reinterpret_cast<unsigned long long *>(mine + kf_wave_lds(W, false));  // CHARS: W * 64 words
  uint16_t *cpre = reinterpret_cast<uint16_t *>(mine + kf_wave_lds(W, false) + W * 64 * 8);          // continuation bytes before each
  uint32_t *lslots = reinterpret_cast<uint32_t *>(smem + (size_t)WPB * kf_wave_lds(W, CHARS));
  // characters that start in [chunk start, chunk start + o), o <= M.S
  auto lead = [&](uint32_t o) -> uint32_t {
    if (asc) return o;
    const uint32_t w = min(o >> 6, W * 64u - 1u), bit = o - w * 64u;
    const unsigned long long low = bit >= 64u ? ~0ull : ((1ull << bit) - 1ull);
    return o - ((uint32_t)cpre[w] + (uint32_t)__popcll(cmask[w] & low));
  };
  const uint32_t *gslots = reinterpret_cast<const uint32_t *>(A.slots);
  if (IMG) {
    for (uint32_t i = threadIdx.x; i < A.n_slots; i += 1024) lslots[i] = gslots[i];
    __syncthreads();
  }
  // (cursor[1] holds 2 or 3 at most from here on: "repeat with larger regions" / "not this engine's batch", written with plain
  // stores by whichever waves find out -- either value makes capi.cpp repeat the call, and 3 wins there when both were
  // written, whatever their order)
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int warm = A.max_len > 1 ? (int)min(A.max_len - 1u, (uint32_t)(kfWarm - 1)) : 0;
  const uint64_t wid = (uint64_t)blockIdx.x * WPB + wave, nw = (uint64_t)gridDim.x * WPB;
  // a chunk's inputs: its candidate bits (lane 0 also those of the `warm` bytes before the chunk: starts whose walks can
  // reach into it) and its document record.  Loaded a chunk ahead: a wave has nothing else to hide their latency behind.
  unsigned long long m[kfMaxWords], mw = 0, nm[kfMaxWords], nmw = 0;
  KfChunk rec{}, nrec{};
  auto fetch = [&](uint64_t c, unsigned long long (&bm)[kfMaxWords], unsigned long long &bw, KfChunk &r) {
    const uint64_t p0 = c * (uint64_t)(M.S / 64u);
#pragma unroll
    for (int k = 0; k < kfMaxWords; k++) {
      const uint64_t w = p0 + (uint64_t)lane * W + (uint32_t)k;
      bm[k] = ((uint32_t)k < W && (int64_t)(w * 64) < N) ? bitmap[w] : 0ull;
    }
    bw = (lane == 0 && p0 > 0 && warm > 0) ? bitmap[p0 - 1] & (~0ull << (64 - warm)) : 0ull;
    r = chunk_rec[c];
  };
  if (wid < M.n_chunks) fetch(wid, nm, nmw, nrec);
  for (uint64_t chunk = wid; chunk < M.n_chunks; chunk += nw) {
#pragma unroll
    for (int k = 0; k < kfMaxWords; k++) m[k] = nm[k];
    mw = nmw;
    rec = nrec;
    if (chunk + nw < M.n_chunks) fetch(chunk + nw, nm, nmw, nrec);
    const int64_t a = (int64_t)chunk * M.S;
    const int64_t e = min(a + (int64_t)M.S, N);
    const int64_t g0 = a - kfWarm;  // text position of offset 0
    if (CHARS && !asc) {
      // piece P = 16 bytes: its continuation bytes as 16 bits (bit 7 set, bit 6 clear; four of them gathered by a multiply)
      uint16_t *cm16 = reinterpret_cast<uint16_t *>(cmask);
      for (uint32_t j = 0; j < M.S / 1024u; j++) {
        const uint32_t P = j * 64u + (uint32_t)lane;
        const int64_t g = a + (int64_t)P * 16;
        const kf_v4u v = g < N ? kf_text16(M.text, g, N) : kf_v4u{0u, 0u, 0u, 0u};
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t c = ((v[k] & (~v[k] << 1)) >> 7) & 0x01010101u;
          bits |= ((c * 0x01020408u) >> 24 & 0xFu) << (4 * k);
        }
        cm16[P] = (uint16_t)bits;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      uint32_t mine_cnt = 0;
      for (uint32_t k = 0; k < W; k++) mine_cnt += (uint32_t)__popcll(cmask[(uint32_t)lane * W + k]);
      uint32_t run = wave_incl_scan(mine_cnt) - mine_cnt;
      for (uint32_t k = 0; k < W; k++) {
        cpre[(uint32_t)lane * W + k] = (uint16_t)run;
        run += (uint32_t)__popcll(cmask[(uint32_t)lane * W + k]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- documents: dn = first boundary at or behind the chunk start.  Lane i looks at boundary dn + i: those below
    // e + kfAhead are the ones a walk of this chunk can meet (usually none or one); their offsets go to LDS, a candidate
    // counts the ones at or before it.  64 of them or more (documents of a few bytes): the general way, from memory.
    const uint64_t dn = rec.dn;
    const int64_t b_prev = rec.b_prev;
    const int64_t bv = dn + (uint64_t)lane <= D ? (int64_t)M.doc_off[dn + (uint64_t)lane] : INT64_MAX;
    // ---- candidate bits -> the chunk's candidates, in position order, as offsets in LDS
    uint32_t cnt = (uint32_t)__popcll(mw);
#pragma unroll
    for (int k = 0; k < kfMaxWords; k++) cnt += (uint32_t)__popcll(m[k]);
    const uint32_t incl = wave_incl_scan(cnt);
    const uint32_t total = wave_last(incl);
    if (total > list_cap) {  // more than a candidate per ~20 bytes: not this engine's text -- the other engines take the call
      if (lane == 0) M.cursor[1] = 3ull;
      break;
    }
    {
      uint32_t rank = incl - cnt;
      while (mw) {
        list[rank++] = (uint16_t)__builtin_ctzll(mw);
        mw &= mw - 1;
      }
#pragma unroll
      for (int k = 0; k < kfMaxWords; k++) {
        const uint32_t o = (uint32_t)kfWarm + ((uint32_t)lane * W + (uint32_t)k) * 64u;
        while (m[k]) {
          list[rank++] = (uint16_t)(o + (uint32_t)__builtin_ctzll(m[k]));
          m[k] &= m[k] - 1;
        }
      }
    }
    // (list[] and bnd[] go from lane to lane through LDS inside one wave: the stores must be out before another lane reads)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t nb = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bv < e + kfAhead));
    const uint32_t n_in = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(bv < e));  // documents that start inside the chunk
    const bool many = nb >= 64u;
    if ((uint32_t)lane < nb) bnd[lane] = (uint32_t)(bv - g0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t before_doc = 0;  // lane i < n_in: events of the chunk that end at or before the first byte of document dn + i
    if (many) {  // their events-before counts start at 0
      for (uint64_t d = dn + (uint64_t)lane; d <= D && (int64_t)M.doc_off[d] < e; d += 64) {
        M.doc_ev_rank[d] = 0;
        if (CHARS) M.doc_lead_rank[d] = lead((uint32_t)((int64_t)M.doc_off[d] - a));
      }
    }
    uint32_t carry = 0;  // furthest offset (exclusive) an earlier start's walk is alive at
    uint32_t seq = 0;    // events of the chunk so far (wave-uniform)
    uint32_t hsum = 0;   // hits of this lane's events
    uint2 *reg = M.evd + chunk * (uint64_t)M.ev_stride;
    bool give_up = false;
    // (the text of a batch's candidates, 16 bytes each straight from memory -- they are too few to stage the chunk for
    // them -- is asked for a batch ahead)
    uint32_t qn = (uint32_t)lane < total ? list[lane] : 0u;
    kf_v4u tn = (uint32_t)lane < total ? kf_text16(M.text, g0 + (int64_t)qn, N) : kf_v4u{0u, 0u, 0u, 0u};
    for (uint32_t first = 0; first < total; first += 64) {
      const bool have = first + (uint32_t)lane < total;
      const uint32_t q = qn;
      const kf_v4u t0 = tn;
      if (first + 64 < total) {
        const bool hn = first + 64 + (uint32_t)lane < total;
        qn = hn ? list[first + 64 + lane] : 0u;
        tn = hn ? kf_text16(M.text, g0 + (int64_t)qn, N) : kf_v4u{0u, 0u, 0u, 0u};
      }
      // ---- the candidate's document: [ds, de) as offsets (a start in a document that ends before the chunk is dead)
      int32_t ds = 0;  // the document's first byte (negative: it starts before offset 0)
      uint32_t de = 0;
      bool ok = have;
      if (!many) {
        uint32_t idx = 0;
        for (uint32_t i = 0; i < nb; i++) idx += bnd[i] <= q ? 1u : 0u;
        // (a warm-up start in front of b_prev lies in a document that ends before the chunk: dead)
        ok = have && (idx > 0 || g0 + (int64_t)q >= b_prev);
        ds = idx ? (int32_t)bnd[idx - 1] : (int32_t)(b_prev - g0);
        de = idx < nb ? bnd[min(idx, 63u)] : (uint32_t)min<int64_t>(N - g0, (int64_t)M.S + kfWarm + kfAhead);
      } else if (have) {
        const int64_t qa = g0 + (int64_t)q;
        uint64_t d = dn > 0 ? dn - 1 : 0;
        if (qa < (int64_t)M.doc_off[d]) {
          ok = false;  // (in the warm-up, in a document before the one that reaches the chunk)
        } else {
          while (d + 1 <= D && (int64_t)M.doc_off[d + 1] <= qa) d++;
          ds = (int32_t)((int64_t)M.doc_off[d] - g0);
          de = (uint32_t)min<int64_t>((d + 1 <= D ? (int64_t)M.doc_off[d + 1] : N) - g0, (int64_t)M.S + kfWarm + kfAhead);
        }
      }
      // ---- the goto walk from q (byte-level image: slot[base ^ byte] belongs to the state iff its label is the byte)
      // Branch-free steps: a lane that is out keeps probing its last state and ignores what comes back.  The END steps of a
      // walk (few) leave {state, offset} in LDS, entry k of lane l at ends[k * 64 + l]; they come back after the walk.
      uint32_t ne = 0;
      bool over = false;
      uint32_t B = A.root, p = q;
      bool alive = ok;
      for (uint32_t blk = 0; blk * 16 < A.max_len; blk++) {
        if (!__builtin_amdgcn_ballot_w64(alive & p < de)) break;
        kf_v4u t = t0;
        if (blk) t = alive ? kf_text16(M.text, g0 + (int64_t)q + (int64_t)blk * 16, N) : kf_v4u{0u, 0u, 0u, 0u};
        bool run = true;  // (wave-uniform; looked at again every four steps)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if ((i & 3) == 0 && i) run = run && __builtin_amdgcn_ballot_w64(alive & p < de) != 0ull;
          if (!run) continue;
          const uint32_t byte = (t[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
          const uint32_t en = IMG ? lslots[B ^ byte] : gslots[B ^ byte];
          // (label 0 is a fail header's: NUL follows no goto)
          const bool hit = alive & p < de & byte != 0u & (en & 0xFFu) == byte;
          alive = hit;
          B = hit ? (en >> C_BASE_SHIFT) & C_BASE_MASK : B;
          p += hit ? 1u : 0u;
          const bool end = hit & (en & C_END) != 0u;
          if (end & ne < (uint32_t)kfMaxEnds) ends[ne * 64u + (uint32_t)lane] = make_uint2(B, p);
          over |= end & ne >= (uint32_t)kfMaxEnds;
          ne += end ? 1u : 0u;
        }
      }
      const uint32_t reach = p > q ? p : 0u;  // offset (exclusive) of the last byte the walk is alive at; 0: not even one byte
      uint32_t ej[kfMaxEnds], eb[kfMaxEnds];
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) {
        const uint2 r = ends[(uint32_t)k * 64u + (uint32_t)lane];
        eb[k] = r.x;
        ej[k] = r.y;
      }
      if (__builtin_amdgcn_ballot_w64(over)) {  // more END steps on one walk than a lane keeps: the other engines take the call
        if (lane == 0) M.cursor[1] = 3ull;
        give_up = true;
        break;
      }
      // ---- an END step at offset j is the reference's event when no earlier start is alive there
      uint32_t pm = wave_incl_scan_max(reach);
      const uint32_t last = wave_last(pm);
      const uint32_t before = max(wave_shr1(pm, 0u), carry);
      carry = max(carry, last);
      uint32_t nv = 0;
      bool val[kfMaxEnds];
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) {
        // this chunk reports the ends in [a, e): offsets (kfWarm, kfWarm + (e - a)]
        val[k] = (uint32_t)k < ne && ej[k] > before && ej[k] > (uint32_t)kfWarm && ej[k] <= (uint32_t)(kfWarm + (e - a));
        nv += val[k] ? 1u : 0u;
      }
      if (!__builtin_amdgcn_ballot_w64(nv != 0u)) continue;
      const uint32_t vincl = wave_incl_scan(nv);
      const uint32_t vtot = wave_last(vincl);
      uint32_t at = seq + vincl - nv;
      // {key id or offset of its flattened chain | chain length << 24, end offset in the document}: what k2d_count makes of
      // {END state, ..} -- one gather per event here saves its launch (4.4 us + the gap of a 64 MiB call) and its pass.
      // All gathers (and the next batch's text, requested at the top) are waited for IN FRONT of the batch's stores: vmcnt
      // counts loads and stores in one order, a load waited for behind a store waits for the store's acknowledgement too.
      uint32_t xk[kfMaxEnds], ck[kfMaxEnds];
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) xk[k] = val[k] ? A.end_info[eb[k]] : 0u;
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) {
        ck[k] = xk[k] >> 24;
        if (val[k] && ck[k] == 255u) ck[k] = A.key_cnt[A.chain ? A.chain[xk[k] & 0xFFFFFFu].y : (xk[k] & 0xFFFFFFu)];
      }
      asm volatile("" : "+v"(tn));
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++)
        if (val[k]) {
          const uint32_t x = xk[k];
          hsum += ck[k];
          uint32_t y = (uint32_t)((int32_t)ej[k] - ds);
          if (CHARS) {  // the document starts inside the chunk: counted from there, exactly; else from the chunk start
            const bool exact = ds >= kfWarm;
            y = ((lead(ej[k] - (uint32_t)kfWarm) - (exact ? lead((uint32_t)(ds - kfWarm)) : 0u)) << 1) | (exact ? 1u : 0u);
          }
          if (at < M.ev_stride) reg[at] = make_uint2(x, y);
          at++;
        }
      // documents that start inside the chunk: the events that end at or before their first byte
      if (!many) {
        for (uint32_t i = 0; i < n_in; i++) {
          const uint32_t bo = bnd[i];
          uint32_t c = 0;
#pragma unroll
          for (int k = 0; k < kfMaxEnds; k++) c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(val[k] && ej[k] <= bo));
          before_doc += (uint32_t)lane == i ? c : 0u;
        }
      } else {
        for (uint64_t d = dn; d <= D && (int64_t)M.doc_off[d] < e; d++) {
          const uint32_t bo = (uint32_t)((int64_t)M.doc_off[d] - g0);
          uint32_t c = 0;
#pragma unroll
          for (int k = 0; k < kfMaxEnds; k++) c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(val[k] && ej[k] <= bo));
          if ((uint64_t)lane == ((d - dn) & 63) && c) M.doc_ev_rank[d] += c;  // (the lane that zeroed it)
        }
      }
      seq += vtot;
    }
    if (give_up) break;
    if (!many && (uint32_t)lane < n_in) {
      M.doc_ev_rank[dn + (uint64_t)lane] = before_doc;
      if (CHARS) M.doc_lead_rank[dn + (uint64_t)lane] = lead(bnd[lane] - (uint32_t)kfWarm);
    }
    hsum = wave_last(wave_incl_scan(hsum));
    if (lane == 0) {
      M.ev_cnt[chunk] = seq;
      M.chunk_hits[chunk] = hsum;
      if (CHARS) {
        M.lead_cnt[chunk] = lead((uint32_t)(e - a));
        M.chunk_doc0[chunk] = (uint32_t)(bv == a ? dn : dn - 1);  // the document that holds the chunk's first byte
      }
      if (seq > M.ev_stride) M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
    }
  }
  // (The scan of the chunks' hit counts does NOT ride on this launch: the last block to finish would do it behind a
  // device-scope fence per block -- on this chip an L2 write-back per XCD -- and the walks' 20 us became 89, profiles/r06_cfg2_fixed_costs.txt.)
}

}  // namespace

size_t filter_chunk_rec_bytes() { return sizeof(KfChunk); }

// LDS of a CU that a block of 16 waves may take (nothing static in kf_walk)
constexpr size_t kfLdsBudget = 160u << 10;
bool filter_image_in_lds(uint32_t n_slots, uint32_t chunk_bytes, bool chars) {
  return (size_t)n_slots * 4 + 16 * (size_t)kf_wave_lds(chunk_bytes / 4096u, chars) <= kfLdsBudget;
}

static size_t walk_lds(bool img, uint32_t n_slots, uint32_t W, bool chars) {
  return (size_t)(img ? 16 : 4) * kf_wave_lds(W, chars) + (img ? (size_t)n_slots * 4 : 0);
}

void filter_launch_filter(const FilterDev &F, const V2Args &M, void *bitmap, void *chunk_rec, unsigned long long *non_ascii,
                          uint32_t cus, void *stream) {
  // two blocks per CU (2 x 64 KiB of LDS): the loop is VALU work, eight waves per SIMD hide its loads; behind them the
  // blocks that write the chunk records (1024 chunks each)
  const uint32_t fb = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((M.n_bytes + 16383) / 16384, (uint64_t)cus * 2));
  const uint32_t grid = fb + (uint32_t)((M.n_chunks + 1023) / 1024);
  if (F.d >= 4)
    hipLaunchKernelGGL(kf_filter<true>, dim3(grid), dim3(1024), 0, (hipStream_t)stream, F, M.text, M.n_bytes, (uint16_t *)bitmap,
                       non_ascii, M, (KfChunk *)chunk_rec, fb);
  else
    hipLaunchKernelGGL(kf_filter<false>, dim3(grid), dim3(1024), 0, (hipStream_t)stream, F, M.text, M.n_bytes, (uint16_t *)bitmap,
                       non_ascii, M, (KfChunk *)chunk_rec, fb);
}

// M.S: the chunk, 4096 << {0, 1, 2, 3}; cus: the device's compute units (a block per CU when the image sits in LDS)
void filter_launch_walk(const DevAut &A, const V2Args &M, const void *bitmap, const void *chunk_rec, const unsigned long long *non_ascii,
                        uint32_t cus, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const bool chars = M.chars != 0;
  const bool img = filter_image_in_lds(A.n_slots, M.S, chars);
  const uint32_t W = M.S / 4096u;
  const auto *bm = (const unsigned long long *)bitmap;
  const auto *cr = (const KfChunk *)chunk_rec;
  if (img) {
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((M.n_chunks + 15) / 16, cus));
    const size_t lds = walk_lds(true, A.n_slots, W, chars);
    if (chars)
      hipLaunchKernelGGL((kf_walk<true, true>), dim3(grid), dim3(1024), lds, s, A, M, bm, cr, non_ascii);
    else
      hipLaunchKernelGGL((kf_walk<true, false>), dim3(grid), dim3(1024), lds, s, A, M, bm, cr, non_ascii);
  } else {
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((M.n_chunks + 3) / 4, (uint64_t)cus * 5));
    const size_t lds = walk_lds(false, 0, W, chars);
    if (chars)
      hipLaunchKernelGGL((kf_walk<false, true>), dim3(grid), dim3(256), lds, s, A, M, bm, cr, non_ascii);
    else
      hipLaunchKernelGGL((kf_walk<false, false>), dim3(grid), dim3(256), lds, s, A, M, bm, cr, non_ascii);
  }
}

int filter_prepare() {
  const void *fs[2] = {reinterpret_cast<const void *>(&kf_walk<true, false>), reinterpret_cast<const void *>(&kf_walk<true, true>)};
  for (const void *f : fs)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kfLdsBudget) != hipSuccess) {
      (void)hipGetLastError();  // (not left for the next call's check to find)
      return -1;
    }
  return 0;
}

}  // namespace aha
