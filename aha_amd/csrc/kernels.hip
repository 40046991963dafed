// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for Aha::AC#match.
//
// Path replaced (reference, file:line):
//   ACX#match_(Bytes)   src/aha/ac.cr:176-192   goto/fail traversal
//   CedarX#child        src/aha/cedar.cr:441-447 base/check probe
//   ACX#fetch           src/aha/ac.cr:265-278   output-chain emission
//   MatchString#match   src/aha/matcher.cr:34-39 byte -> char offset remap
//   ACX#match(seq,sep)  src/aha/ac.cr:321-340   separator-filtered emission
//
// Parallelisation: the corpus is cut into fixed-size chunks at absolute byte
// positions.  A chunk's traversal starts at root (Lmax-1) bytes before the
// chunk (clamped to the start of the enclosing document); after that warm-up
// its state equals the sequential automaton's state exactly (the AC state is
// the longest suffix that is a trie path, depth <= Lmax), so the reference's
// non-textbook emission rule is reproduced bit for bit.  Hits are produced in
// the reference's order by count -> scan -> ordered write.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"

namespace aha {

// ------------------------------------------------------------- count kernel
// Pass 1: per-chunk number of hits (and UTF-8 lead bytes in chars mode).
template <bool COMPACT>
__global__ __launch_bounds__(kBlock) void k_count(DevAut A, MatchArgs M) {
  __shared__ uint64_t sm[kBlock / 64];
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  uint32_t hits = 0, leads = 0;
  if (c < M.n_chunks) {
    const uint64_t N = M.n_bytes, D = M.n_docs;
    const uint64_t a = c * M.chunk;
    const uint64_t e = min(a + M.chunk, N);
    uint64_t dn = first_boundary(M.doc_off, D, a);
    uint64_t nb = M.doc_off[dn];
    uint64_t doc_start = a;
    uint32_t B = A.root, key = 0;
    if (nb != a) {
      doc_start = M.doc_off[dn - 1];
      uint64_t back = min<uint64_t>(a - doc_start, (uint64_t)(A.max_len ? A.max_len - 1 : 0));
      for (uint64_t p = a - back; p < a; p++) aut_step<COMPACT>(A, B, M.text + p, key);
    }
    for (uint64_t p = a; p < e; p++) {
      if (p == nb) {
        do {
          dn++;
          nb = dn <= D ? M.doc_off[dn] : ~0ull;
        } while (nb == p);
        B = A.root;
        doc_start = p;
      }
      const uint32_t b = M.text[p];
      if (M.chars) leads += (b & 0xC0u) != 0x80u;
      if (aut_step<COMPACT>(A, B, M.text + p, key)) {
        if (!M.sep) {
          hits += A.key_cnt[key];
        } else {
          // match(seq, sep): right neighbour of the end position (ac.cr:324-329)
          if (p + 1 < nb && sep_blocked(M, M.text[p + 1])) continue;
          int32_t k = (int32_t)key;
          do {
            uint2 ln = A.key_ln[k];
            uint64_t s = p + 1 - ln.x;  // absolute start
            // left neighbour of the hit (ac.cr:331-336)
            if (!(s > doc_start && sep_blocked(M, M.text[s - 1]))) hits++;
            k = (int32_t)ln.y;
          } while (k >= 0);
        }
      }
    }
    M.counts[c] = hits;
    if (M.chars) M.leads[c] = leads;
  }
  // block sums for the scan
  uint64_t tot;
  block_excl_scan<uint64_t>(hits, sm, &tot);
  if (threadIdx.x == 0) M.blk_hits[blockIdx.x] = tot;
  if (M.chars) {
    block_excl_scan<uint64_t>(leads, sm, &tot);
    if (threadIdx.x == 0) M.blk_leads[blockIdx.x] = tot;
  }
}

// -------------------------------------------------------------- scan kernel
// Exclusive scan of the per-block sums, in place; one workgroup walks the
// array tile by tile with a running carry.  totals[0] = hits, totals[1] = leads.
__global__ __launch_bounds__(1024) void k_scan_blocks(uint64_t *blk_hits, uint64_t *blk_leads,
                                                      uint64_t n, uint64_t *totals) {
  __shared__ uint64_t sm[16];
  uint64_t *arrs[2] = {blk_hits, blk_leads};
  for (int which = 0; which < 2; which++) {
    uint64_t *arr = arrs[which];
    if (!arr) continue;
    uint64_t carry = 0;
    for (uint64_t t0 = 0; t0 < n; t0 += 1024) {
      uint64_t i = t0 + threadIdx.x;
      uint64_t v = i < n ? arr[i] : 0;
      // block scan over 1024 threads = 16 waves
      const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
      uint64_t inc = wave_incl_scan(v);
      if (lane == 63) sm[w] = inc;
      __syncthreads();
      uint64_t base = 0, tot = 0;
      for (int k = 0; k < 16; k++) {
        uint64_t s = sm[k];
        if (k < w) base += s;
        tot += s;
      }
      if (i < n) arr[i] = carry + base + inc - v;
      __syncthreads();  // sm is rewritten by the next tile
      carry += tot;
    }
    if (threadIdx.x == 0) totals[which] = carry;
  }
}

// -------------------------------------------------------------- docg kernel
// chars mode: absolute lead-byte count at the start of every document, so the
// write pass can turn absolute counts into per-document char offsets
// (char_map, matcher.cr:14-22, without the 4-bytes-per-input-byte array).
__global__ void k_docg(MatchArgs M) {
  const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d > M.n_docs) return;
  const uint64_t q = M.doc_off[d];
  if (q >= M.n_bytes) {
    M.docg[d] = M.totals[1];
    return;
  }
  const uint64_t c = q / M.chunk;
  uint64_t g = M.blk_leads[c / kBlock];
  for (uint64_t cc = (c / kBlock) * kBlock; cc < c; cc++) g += M.leads[cc];
  for (uint64_t p = c * M.chunk; p < q; p++) g += (M.text[p] & 0xC0u) != 0x80u;
  M.docg[d] = g;
}

// -------------------------------------------------------------- write kernel
// Pass 2: same traversal, hits written at their final position.
template <bool COMPACT>
__global__ __launch_bounds__(kBlock) void k_write(DevAut A, MatchArgs M) {
  __shared__ uint64_t sm[kBlock / 64];
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = c < M.n_chunks;
  uint64_t idx = M.blk_hits[blockIdx.x] +
                 block_excl_scan<uint64_t>(live ? M.counts[c] : 0u, sm, nullptr);
  uint64_t lead_abs = 0;
  if (M.chars)
    lead_abs = M.blk_leads[blockIdx.x] +
               block_excl_scan<uint64_t>(live ? M.leads[c] : 0u, sm, nullptr);
  if (!live) return;

  const uint64_t N = M.n_bytes, D = M.n_docs;
  const uint64_t a = c * M.chunk;
  const uint64_t e = min(a + M.chunk, N);
  uint64_t dn = first_boundary(M.doc_off, D, a);
  uint64_t nb = M.doc_off[dn];
  uint64_t doc_start = a;
  uint64_t doc_lead0 = 0;
  uint32_t B = A.root, key = 0;
  if (nb != a) {
    doc_start = M.doc_off[dn - 1];
    if (M.chars) doc_lead0 = M.docg[dn - 1];
    uint64_t back = min<uint64_t>(a - doc_start, (uint64_t)(A.max_len ? A.max_len - 1 : 0));
    for (uint64_t p = a - back; p < a; p++) aut_step<COMPACT>(A, B, M.text + p, key);
  }
  for (uint64_t p = a; p < e; p++) {
    if (p == nb) {
      do {
        if (M.doc_hit_off) M.doc_hit_off[dn] = idx;
        dn++;
        nb = dn <= D ? M.doc_off[dn] : ~0ull;
      } while (nb == p);
      B = A.root;
      doc_start = p;
      doc_lead0 = lead_abs;
    }
    const uint32_t b = M.text[p];
    if (M.chars) lead_abs += (b & 0xC0u) != 0x80u;
    if (aut_step<COMPACT>(A, B, M.text + p, key)) {
      if (M.sep && p + 1 < nb && sep_blocked(M, M.text[p + 1])) continue;
      const int32_t end_b = (int32_t)(p - doc_start) + 1;
      int32_t k = (int32_t)key;
      do {
        uint2 ln = A.key_ln[k];
        const int32_t start_b = end_b - (int32_t)ln.x;
        if (!(M.sep && start_b > 0 && sep_blocked(M, M.text[doc_start + start_b - 1]))) {
          if (idx < M.cap) {
            aha_hit h;
            if (M.chars) {
              // end_char = #lead bytes in doc[0,end); start_char = end_char - kc - 1
              const int32_t end_c = (int32_t)(lead_abs - doc_lead0);
              h.start = end_c - (int32_t)A.key_kc[k] - 1;
              h.end = end_c;
            } else {
              h.start = start_b;
              h.end = end_b;
            }
            h.value = k;
            M.out[idx] = h;
          }
          idx++;
        }
        k = (int32_t)ln.y;
      } while (k >= 0);
    }
  }
  if (e == N && M.doc_hit_off) {
    // documents that start at N (empty tail documents) and the final total
    while (dn <= D) {
      M.doc_hit_off[dn] = idx;
      dn++;
    }
  }
}

// ------------------------------------------------------------ match_longest
// ACX#match_longest (src/aha/ac.cr:118-143 match_longest_, :249-263 fetch_one, :297-303): the traversal keeps the
// last position whose state ends a key (prev_i, prev_nid); when a byte finds no goto from the current state that
// pending end is yielded (its own key: fetch_one stops at the first live chain entry), cleared, and -- unless
// `intersectable` -- the state is reset to the root, where the byte is NOT tried again (ac.cr:131-136).  Whatever is
// pending at the end of the sequence is yielded last.
// is_end? (cedar.cr:657-660) is true for a state that ends a key AND for a state whose Cedar node kept the END flag of
// an evicted node (a slot reused inside resolve, cedar.cr:642-648): such a stale end replaces the pending end by one
// for which fetch_one yields nothing, and still resets the state when it is "yielded".  Which states are stale follows
// from Cedar's slot history, replayed on the host (cedar_replay.cpp): A.stale_bits, one bit per slot at the state's base.
// A NUL byte (keys hold none, cedar.cr:235) is not simply a miss here: a Cedar node that holds a value AND has children
// keeps the value in a child with label 0, which `child` finds like any other (cedar.cr:441-447).  That node has no
// children of its own, so is_end? holds for it (cedar.cr:657-660) -- it REPLACES the pending end by one that yields
// nothing (fetch_one finds no output) -- and compile's BFS skips it (cedar.cr:450-463), so its fail link is unset: the
// byte after it is consumed at the root without a goto (defined for intersectable = false; for true the reference
// reads array[-1] there and the oracle pins the same).  A.term_bits marks the states with such a child (they end a
// key and have children), kValueNode is the state "in it".
constexpr uint32_t kNoKey = 0xFFFFFFFFu;  // a pending end that yields nothing
constexpr uint32_t kValueNode = 0xFFFFFFFEu;
struct Pending {
  int64_t p = -1;      // absolute position of the pending end (-1: none)
  uint32_t key = 0;
  uint32_t endc = 0;   // chars mode: lead bytes of the document up to and including p
};

// One byte of match_longest_.  Returns true when a pending end was yielded (into *out).
template <bool COMPACT>
__device__ __forceinline__ bool longest_step(const DevAut &A, uint32_t &B, const uint8_t *tp, int64_t p, uint32_t lc,
                                             bool intersectable, Pending &pend, Pending *out) {
  const uint32_t b = *tp;
  bool yielded = false;
  if (B == kValueNode) {  // no goto from it, no fail link: the pending (empty) end is yielded, the byte is gone
    *out = pend;
    pend.p = -1;
    B = A.root;
    return true;
  }
  for (;;) {
    uint32_t key = 0;
    if (b == 0 && A.term_bits && ((A.term_bits[B >> 5] >> (B & 31)) & 1u)) {
      B = kValueNode;
      pend.p = p;
      pend.key = kNoKey;
      pend.endc = lc;
      break;
    }
    const int r = b ? Probe<COMPACT>::go(A, B, b, key) : 0;  // a NUL byte has no other goto (keys hold none)
    if (r) {
      if (r == 2) {
        pend.p = p;
        pend.key = key;
        pend.endc = lc;
      } else if (A.stale_bits && ((A.stale_bits[B >> 5] >> (B & 31)) & 1u)) {
        pend.p = p;
        pend.key = kNoKey;
        pend.endc = lc;
      }
      break;
    }
    if (pend.p >= 0) {
      *out = pend;
      yielded = true;
      pend.p = -1;
      if (!intersectable) B = A.root;
    }
    if (B == A.root) break;
    B = fail_of<COMPACT>(A, B, tp);
  }
  return yielded;
}

__device__ __forceinline__ void longest_emit(const DevAut &A, const MatchArgs &M, const Pending &h, uint64_t doc_start,
                                             uint64_t &idx) {
  if (idx < M.cap) {
    aha_hit o;
    if (M.chars) {  // Hit(char_of_byte[start], char_of_byte[end - 1] + 1) ac.cr:305-310
      o.end = (int32_t)h.endc;
      o.start = (int32_t)h.endc - (int32_t)A.key_kc[h.key] - 1;
    } else {        // Hit(idx - len + 1, idx + 1) ac.cr:255-257
      o.end = (int32_t)(h.p - (int64_t)doc_start) + 1;
      o.start = o.end - (int32_t)A.key_ln[h.key].x;
    }
    o.value = (int32_t)h.key;
    M.out[idx] = o;
  }
  idx++;
}

// One thread per DOCUMENT, the whole sequence in order: exact for both forms; the only form for intersectable =
// false, whose state depends on every earlier yield of the document.  WRITE = false counts, true writes.
template <bool COMPACT, bool WRITE>
__global__ __launch_bounds__(kBlock) void k_longest_docs(DevAut A, MatchArgs M, int intersectable) {
  __shared__ uint64_t sm[kBlock / 64];
  const uint64_t d = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = d < M.n_docs;
  uint64_t idx = 0;
  if (WRITE) idx = M.blk_hits[blockIdx.x] + block_excl_scan<uint64_t>(live ? M.counts[d] : 0u, sm, nullptr);
  uint32_t hits = 0;
  if (live) {
    const uint64_t a = M.doc_off[d], e = M.doc_off[d + 1];
    if (WRITE && M.doc_hit_off) M.doc_hit_off[d] = idx;
    uint32_t B = A.root, lc = 0;
    Pending pend, out;
    for (uint64_t p = a; p < e; p++) {
      if (M.chars) lc += (M.text[p] & 0xC0u) != 0x80u;
      if (longest_step<COMPACT>(A, B, M.text + p, (int64_t)p, lc, intersectable != 0, pend, &out) &&
          out.key != kNoKey) {
        if (WRITE) longest_emit(A, M, out, a, idx);
        hits++;
      }
    }
    if (pend.p >= 0 && pend.key != kNoKey) {
      if (WRITE) longest_emit(A, M, pend, a, idx);
      hits++;
    }
    if (!WRITE) M.counts[d] = hits;
  }
  if (!WRITE) {
    uint64_t tot;
    block_excl_scan<uint64_t>(hits, sm, &tot);
    if (threadIdx.x == 0) M.blk_hits[blockIdx.x] = tot;
  } else if (d == M.n_docs && M.doc_hit_off) {
    M.doc_hit_off[d] = M.totals[0];
  }
}

// intersectable = true keeps the plain automaton state, so chunks are independent given a warm-up: a thread owns
// the yields whose pending end lies in its chunk.  A run of direct gotos is shorter than Lmax, so the last miss in
// front of the chunk is at most Lmax back and the state there needs Lmax more bytes to be exact: the thread starts
// 2 * Lmax bytes early (clamped to the document) and walks on past its chunk until its last pending end is yielded
// or replaced by one that belongs to the next chunk.
template <bool COMPACT, bool WRITE>
__global__ __launch_bounds__(kBlock) void k_longest_chunks(DevAut A, MatchArgs M) {
  __shared__ uint64_t sm[kBlock / 64];
  const uint64_t c = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool live = c < M.n_chunks;
  uint64_t idx = 0;
  if (WRITE) idx = M.blk_hits[blockIdx.x] + block_excl_scan<uint64_t>(live ? M.counts[c] : 0u, sm, nullptr);
  uint32_t hits = 0;
  if (live) {
    const uint64_t N = M.n_bytes, D = M.n_docs;
    const uint64_t a = c * M.chunk, e = min(a + M.chunk, N);
    uint64_t dn = first_boundary(M.doc_off, D, a);  // first document that starts at or after a
    uint64_t nb = M.doc_off[dn];
    uint64_t doc_start = a;
    uint64_t p = a;
    if (nb != a) {
      doc_start = M.doc_off[dn - 1];
      const uint64_t W = 2ull * A.max_len;
      p = a - min<uint64_t>(a - doc_start, W);
      if (M.has_nul) {
        // A NUL makes the byte behind it vanish or not depending on the state it meets (kValueNode), so the state behind
        // a NUL is exact only if the state AT the NUL was: the warm-up must reach 2 * Lmax NUL-free bytes in front of the
        // earliest NUL it crosses (or the start of the document, or two NULs in a row: whatever the first one did, the
        // state behind the second is the root).  Text with a NUL every few bytes (UTF-16) would walk back for ever: past
        // kNulBack bytes the chunk gives up and the host runs the batch document by document (totals[1] = 2).
        constexpr uint64_t kNulBack = 4096;
        uint64_t hi = a;  // [p, hi) has not been searched yet
        for (;;) {
          uint64_t z = ~0ull;
          for (uint64_t q = p; q < hi; q++)
            if (M.text[q] == 0) {
              z = q;
              break;
            }
          if (z == ~0ull || p == doc_start) break;  // a clean warm-up, or the walk starts with the document: exact
          if (z > doc_start && M.text[z - 1] == 0) {  // two in a row: the root behind them
            p = z + 1;
            break;
          }
          if (a - z > kNulBack) {
            M.totals[1] = 2ull;
            break;
          }
          hi = z;
          p = z - min<uint64_t>(z - doc_start, W);
        }
      }
    }
    uint32_t B = A.root;
    const uint32_t lc = 0;  // byte offsets only: char offsets take the per-document kernel (launch_longest)
    Pending pend, out;
    auto mine = [&](const Pending &h) { return (uint64_t)h.p >= a && (uint64_t)h.p < e; };
    for (;;) {
      if (p == nb) {  // end of a document: its pending end is yielded, the next one starts at the root
        if (pend.p >= 0 && mine(pend) && pend.key != kNoKey) {
          if (WRITE) longest_emit(A, M, pend, doc_start, idx);
          hits++;
        }
        pend.p = -1;
        if (p >= e) break;
        do {
          if (WRITE && M.doc_hit_off && p >= a) M.doc_hit_off[dn] = idx;
          dn++;
          nb = dn <= D ? M.doc_off[dn] : ~0ull;
        } while (nb == p);
        B = A.root;
        doc_start = p;
        if (p >= N) break;
      }
      if (p >= e && (pend.p < 0 || !mine(pend))) break;  // nothing of this chunk is pending any more
      if (longest_step<COMPACT>(A, B, M.text + p, (int64_t)p, lc, true, pend, &out) && mine(out) &&
          out.key != kNoKey) {
        if (WRITE) longest_emit(A, M, out, doc_start, idx);
        hits++;
      }
      p++;
    }
    if (!WRITE) M.counts[c] = hits;
    if (WRITE && e == N && M.doc_hit_off) {  // documents that start at N (empty tail documents) and the final total
      const uint64_t tot = M.totals[0];
      for (uint64_t q = first_boundary(M.doc_off, D, N); q <= D; q++) M.doc_hit_off[q] = tot;
    }
  }
  if (!WRITE) {
    uint64_t tot;
    block_excl_scan<uint64_t>(hits, sm, &tot);
    if (threadIdx.x == 0) M.blk_hits[blockIdx.x] = tot;
  }
}

// Is there a NUL byte in the text?  The chunked form of match_longest (a thread per chunk with a warm-up) relies on the
// state being a function of the last Lmax bytes; a NUL makes the byte after it vanish or not depending on the state it
// met (kValueNode), which a warm-up cannot know -- such a batch takes the per-document kernel.
__global__ __launch_bounds__(256) void k_has_nul(const uint8_t *text, uint64_t n, uint64_t *flag) {
  bool z = false;
  const uint64_t n16 = n / 16;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4 *>(text)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; k++) z |= ((w[k] - 0x01010101u) & ~w[k] & 0x80808080u) != 0u;  // some byte of the word is 0
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) z |= text[n16 * 16 + threadIdx.x] == 0;
  if (__any(z) && (threadIdx.x & 63) == 0) *flag = 1;
}

void launch_has_nul(const uint8_t *text, uint64_t n, uint64_t *flag, void *stream) {
  const uint32_t g = (uint32_t)std::min<uint64_t>((n / 16 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(k_has_nul, dim3(g), dim3(256), 0, (hipStream_t)stream, text, n, flag);
}

void launch_longest(const DevAut &A, const MatchArgs &M, int mode, bool write, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode != 2) {  // 1: intersectable = false, 3: true -- one thread per document (+1 for the closing offset)
    const uint32_t g = (uint32_t)((M.n_docs + 1 + kBlock - 1) / kBlock);
    const int isect = mode == 3 ? 1 : 0;
#define AHA_LD(C, W) hipLaunchKernelGGL((k_longest_docs<C, W>), dim3(g), dim3(kBlock), 0, s, A, M, isect)
    if (A.compact) { if (write) AHA_LD(true, true); else AHA_LD(true, false); }
    else           { if (write) AHA_LD(false, true); else AHA_LD(false, false); }
#undef AHA_LD
  } else {
    const uint32_t g = (uint32_t)((M.n_chunks + kBlock - 1) / kBlock);
#define AHA_LC(C, W) hipLaunchKernelGGL((k_longest_chunks<C, W>), dim3(g), dim3(kBlock), 0, s, A, M)
    if (A.compact) { if (write) AHA_LC(true, true); else AHA_LC(true, false); }
    else           { if (write) AHA_LC(false, true); else AHA_LC(false, false); }
#undef AHA_LC
  }
}

// ------------------------------------------------ exchange format (multi-GPU)
// A hit's start is its end minus the key's length (ac.cr:270-272; in char
// offsets: minus the key's UTF-8 lead bytes), so ranks exchange {end, value}
// pairs -- 8 instead of 12 bytes per hit on the xGMI links -- and rebuild the
// triples on arrival.  One thread per 32-bit word on both sides: fully
// coalesced loads and stores.
__global__ __launch_bounds__(256) void k_hits_pack(const int32_t *hits, uint64_t n_words, int32_t *pairs) {
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_words; j += (uint64_t)gridDim.x * 256) {
    const uint64_t i = j >> 1;
    pairs[j] = hits[i * 3 + 1 + (j & 1)];
  }
}

__global__ __launch_bounds__(256) void k_hits_unpack(const int32_t *pairs, uint64_t n_words, const uint2 *key_ln,
                                                      const uint32_t *key_kc, int chars, int32_t *hits) {
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n_words; j += (uint64_t)gridDim.x * 256) {
    const uint64_t i = j / 3;
    const uint32_t f = (uint32_t)(j - i * 3);
    int32_t v = pairs[i * 2 + (f == 2 ? 1 : 0)];  // end for start/end, value for value
    if (f == 0) {
      const int32_t key = pairs[i * 2 + 1];
      v -= chars ? (int32_t)key_kc[key] + 1 : (int32_t)key_ln[key].x;
    }
    hits[j] = v;
  }
}

void launch_hits_pack(const int32_t *hits, uint64_t n, int32_t *pairs, void *stream) {
  if (!n) return;
  const uint64_t w = n * 2;
  const uint32_t g = (uint32_t)std::min<uint64_t>((w + 255) / 256, 1u << 16);
  hipLaunchKernelGGL(k_hits_pack, dim3(g), dim3(256), 0, (hipStream_t)stream, hits, w, pairs);
}

void launch_hits_unpack(const DevAut &A, const int32_t *pairs, uint64_t n, int chars, int32_t *hits, void *stream) {
  if (!n) return;
  const uint64_t w = n * 3;
  const uint32_t g = (uint32_t)std::min<uint64_t>((w + 255) / 256, 1u << 16);
  hipLaunchKernelGGL(k_hits_unpack, dim3(g), dim3(256), 0, (hipStream_t)stream, pairs, w, A.key_ln, A.key_kc, chars,
                     hits);
}

// -------------------------------------------------- 4-byte exchange stream
// Hits leave the match in per-document order with ascending `end`, so the stream of `end` values is a sequence of
// small non-negative steps with a reset per document.  The compressed exchange format of n hits:
//   words[n]   value << (sb + lb) | len << sb | step; step = end - previous end when that is below 2^sb - 1, else
//              2^sb - 1 = exception; len = end - start (lb bits; lb = 0: not carried)
//   blk[nb]    nb = ceil(n / 1024): index into exc[] of the block's first exception (every block starts with one)
//   exc[]      the absolute `end` of every exception, in hit order
// i.e. 4 bytes per hit + 8 bytes per 1024 hits + 4 bytes per document change or long gap.  The widths (StreamFmt) follow
// from the automaton, which every rank holds: with the key's length in the word the receiver needs no table lookup per
// hit to rebuild Hit#start (ac.cr:270-272) -- its seven rebuilds per step were bound by that gather (1.55 -> 0.75 ms) --;
// where key ids and lengths leave fewer than 6 bits for the step, the length is looked up as in the 8-byte form.
constexpr uint32_t kPk4Block = 1024;

__device__ __forceinline__ bool pack4_is_exc(const int32_t *hits, uint64_t i, uint32_t t, int32_t &e, uint32_t &step,
                                             uint32_t exc_code) {
  e = hits[i * 3 + 1];
  if (t == 0) {
    step = exc_code;
    return true;
  }
  const int64_t d = (int64_t)e - (int64_t)hits[(i - 1) * 3 + 1];
  const bool x = d < 0 || d >= (int64_t)exc_code;
  step = x ? exc_code : (uint32_t)d;
  return x;
}

__global__ __launch_bounds__(kPk4Block) void k_pack4_flags(const int32_t *hits, uint64_t n, uint32_t *blk, StreamFmt F) {
  const uint64_t i = (uint64_t)blockIdx.x * kPk4Block + threadIdx.x;
  int32_t e;
  uint32_t step;
  const int x = i < n && pack4_is_exc(hits, i, threadIdx.x, e, step, (1u << F.step_bits) - 1u);
  const int c = __syncthreads_count(x);
  if (threadIdx.x == 0) blk[blockIdx.x] = (uint32_t)c;
}

// exclusive scan of blk[0..nb) in place by one workgroup; n_words = n + nb + number of exceptions
__global__ __launch_bounds__(1024) void k_pack4_scan(uint32_t *blk, uint64_t nb, uint64_t n, unsigned long long *n_words) {
  __shared__ uint32_t wsum[16];
  __shared__ unsigned long long run;
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) run = 0;
  __syncthreads();
  for (uint64_t base = 0; base < nb; base += 1024) {
    const uint64_t i = base + t;
    const uint32_t v = i < nb ? blk[i] : 0;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t u = 0; u < w; u++) before += wsum[u];
    const unsigned long long r = run;
    if (i < nb) blk[i] = (uint32_t)(r + before + inc - v);
    __syncthreads();
    if (t == 1023) run = r + before + inc;
    __syncthreads();
  }
  if (t == 0) *n_words = n + nb + run;
}

__global__ __launch_bounds__(kPk4Block) void k_pack4_write(const int32_t *hits, uint64_t n, uint32_t *words,
                                                           const uint32_t *blk, int32_t *exc, StreamFmt F) {
  __shared__ uint32_t wcnt[16];
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * kPk4Block + t;
  int32_t e = 0;
  uint32_t step = 0;
  const bool x = i < n && pack4_is_exc(hits, i, t, e, step, (1u << F.step_bits) - 1u);
  const unsigned long long m = __ballot(x);
  if (lane == 0) wcnt[w] = (uint32_t)__popcll(m);
  __syncthreads();
  if (i >= n) return;
  uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (uint32_t u = 0; u < w; u++) rank += wcnt[u];
  const uint32_t len = F.len_bits ? (uint32_t)(e - hits[i * 3]) : 0u;  // Hit#end - Hit#start, in the batch's offsets
  words[i] = ((uint32_t)hits[i * 3 + 2] << (F.step_bits + F.len_bits)) | (len << F.step_bits) | step;
  if (x) exc[(uint64_t)blk[blockIdx.x] + rank] = e;
}

// Rebuilds the Hit triples of up to kMaxSegs streams in ONE launch (an 8-GPU step receives seven): the workgroups
// [blk0[s], blk0[s+1]) belong to stream s, which starts at word woff[s] of `land`, holds nh[s] hits and is written
// to hits[ooff[s]..].
struct SegTab {
  uint32_t n;
  uint32_t blk0[kMaxSegs + 1];
  uint64_t woff[kMaxSegs], nh[kMaxSegs], ooff[kMaxSegs];
};

// 256 threads rebuild a block of 1024 hits, four consecutive hits per thread: one 16-byte load of the words, the
// exception ranks from four ballots, a segmented scan over (restart, sum) pairs -- inside the thread, then across the
// wave, then across the four waves --, and the 48 bytes of a thread's triples go through LDS so that every store
// instruction writes 1 KiB of consecutive bytes (16 bytes per lane).
constexpr uint32_t kUnpThreads = 256;

__global__ __launch_bounds__(kUnpThreads) void k_unpack4(const uint32_t *land, SegTab S, const uint2 *key_ln,
                                                         const uint32_t *key_kc, int chars, int32_t *hits_all,
                                                         StreamFmt F) {
  const uint32_t exc_code = (1u << F.step_bits) - 1u, vshift = F.step_bits + F.len_bits;
  __shared__ uint32_t wcnt[4];
  __shared__ int32_t wagg[4];
  __shared__ uint32_t wflag[4];
  __shared__ uint4 stage[kPk4Block * 3 / 4];
  uint32_t seg = 0;
  while (seg + 1 < S.n && blockIdx.x >= S.blk0[seg + 1]) seg++;  // wave-uniform: a handful of scalar compares
  const uint32_t bid = blockIdx.x - S.blk0[seg];
  const uint64_t n = S.nh[seg];
  const uint64_t nb = (n + kPk4Block - 1) / kPk4Block;
  const uint32_t *words = land + S.woff[seg];
  const uint32_t *blk = words + n;
  const int32_t *exc = reinterpret_cast<const int32_t *>(words + n + nb);
  int32_t *hits = hits_all + S.ooff[seg] * 3;
  const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
  const uint64_t i0 = (uint64_t)bid * kPk4Block + 4u * t;
  const uint32_t left = i0 < n ? (uint32_t)min<uint64_t>(4, n - i0) : 0u;  // valid hits of this thread
  uint32_t wd[4] = {0, 0, 0, 0};
  const uint32_t *src = words + i0;
  if (left == 4 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    const uint4 q = *reinterpret_cast<const uint4 *>(src);
    wd[0] = q.x; wd[1] = q.y; wd[2] = q.z; wd[3] = q.w;
  } else {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
      if (k < left) wd[k] = src[k];
  }
  bool f[4];
  uint32_t before = 0, mine = 0;  // exceptions of the wave before this thread; of this thread
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    f[k] = k < left && (wd[k] & exc_code) == exc_code;
    const unsigned long long m = __ballot(f[k]);
    before += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    mine += f[k] ? 1u : 0u;
  }
  {
    // exceptions of the whole wave = those before the last lane + the last lane's own
    const uint32_t tot = __shfl(before + mine, 63, 64);
    if (lane == 0) wcnt[w] = tot;
  }
  __syncthreads();
  uint32_t rank = before;
  for (uint32_t u = 0; u < w; u++) rank += wcnt[u];
  const uint64_t e0 = (uint64_t)blk[bid] + rank;
  // inside the thread: pre[k] = end of hit k if nothing came before the thread, seen[k] = an exception at or before k
  int32_t pre[4];
  bool seen[4];
  int32_t v = 0;
  bool any = false;
  uint32_t taken = 0;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const int32_t x = f[k] ? exc[e0 + taken] : (int32_t)(wd[k] & exc_code);
    taken += f[k] ? 1u : 0u;
    v = f[k] ? x : v + x;
    any = any || f[k];
    pre[k] = v;
    seen[k] = any;
  }
  // across the wave: inclusive segmented scan of the threads' (restart, sum) pairs
  int32_t x = v;
  uint32_t fl = any ? 1u : 0u;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t ux = __shfl_up(x, o);
    const uint32_t uf = __shfl_up(fl, o);
    if ((int)lane >= o) {
      if (!fl) x += ux;
      fl |= uf;
    }
  }
  if (lane == 63) {
    wagg[w] = x;
    wflag[w] = fl;
  }
  int32_t cx = __shfl_up(x, 1);  // what the threads before this one carry in
  uint32_t cf = __shfl_up(fl, 1);
  if (lane == 0) {
    cx = 0;
    cf = 0;
  }
  __syncthreads();
  if (!cf) {  // nothing restarted the sum in this wave before the thread: add what the earlier waves hold
    int32_t c = 0;
    for (uint32_t u = 0; u < w; u++) c = wflag[u] ? wagg[u] : c + wagg[u];
    cx += c;
  }
  int32_t o12[12];
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint32_t value = wd[k] >> vshift;
    const int32_t end = seen[k] ? pre[k] : pre[k] + cx;
    int32_t len = 0;
    if (F.len_bits)
      len = (int32_t)((wd[k] >> F.step_bits) & ((1u << F.len_bits) - 1u));
    else if (k < left)
      len = chars ? (int32_t)key_kc[value] + 1 : (int32_t)key_ln[value].x;
    o12[3 * k + 0] = end - len;
    o12[3 * k + 1] = end;
    o12[3 * k + 2] = (int32_t)value;
  }
  int32_t *dst = hits + (uint64_t)bid * kPk4Block * 3;
  const uint64_t rest = n - (uint64_t)bid * kPk4Block;  // hits of this block and after
  if (rest >= kPk4Block && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {  // block-uniform
#pragma unroll
    for (uint32_t j = 0; j < 3; j++)
      stage[3 * t + j] = make_uint4((uint32_t)o12[4 * j], (uint32_t)o12[4 * j + 1], (uint32_t)o12[4 * j + 2],
                                    (uint32_t)o12[4 * j + 3]);
    __syncthreads();
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) d4[j * kUnpThreads + t] = stage[j * kUnpThreads + t];
  } else {  // the last block of a stream, or an output that is not 16-byte aligned: dword stores through the same staging
    int32_t *st = reinterpret_cast<int32_t *>(stage);
#pragma unroll
    for (uint32_t q = 0; q < 12; q++) st[12 * t + q] = o12[q];
    __syncthreads();
    const uint32_t total = (uint32_t)min<uint64_t>(rest, kPk4Block) * 3;
    for (uint32_t j = t; j < total; j += kUnpThreads) dst[j] = st[j];
  }
}

void launch_hits_pack4(const int32_t *hits, uint64_t n, uint32_t *stream_words, unsigned long long *n_words,
                       StreamFmt F, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint64_t nb = (n + kPk4Block - 1) / kPk4Block;
  uint32_t *words = stream_words, *blk = stream_words + n;
  int32_t *exc = reinterpret_cast<int32_t *>(stream_words + n + nb);
  if (nb) hipLaunchKernelGGL(k_pack4_flags, dim3((uint32_t)nb), dim3(kPk4Block), 0, s, hits, n, blk, F);
  hipLaunchKernelGGL(k_pack4_scan, dim3(1), dim3(1024), 0, s, blk, nb, n, n_words);
  if (nb) hipLaunchKernelGGL(k_pack4_write, dim3((uint32_t)nb), dim3(kPk4Block), 0, s, hits, n, words, blk, exc, F);
}

void launch_hits_unpack4_segs(const DevAut &A, const uint32_t *land, const uint64_t *word_off, const uint64_t *n_hits,
                              const uint64_t *out_off, uint32_t n_segs, int chars, int32_t *hits, StreamFmt F,
                              void *stream) {
  SegTab S{};
  uint32_t blocks = 0;
  for (uint32_t k = 0; k < n_segs && S.n < kMaxSegs; k++) {
    if (!n_hits[k]) continue;
    S.blk0[S.n] = blocks;
    S.woff[S.n] = word_off[k];
    S.nh[S.n] = n_hits[k];
    S.ooff[S.n] = out_off[k];
    blocks += (uint32_t)((n_hits[k] + kPk4Block - 1) / kPk4Block);
    S.n++;
  }
  S.blk0[S.n] = blocks;
  if (!blocks) return;
  hipLaunchKernelGGL(k_unpack4, dim3(blocks), dim3(kUnpThreads), 0, (hipStream_t)stream, land, S, A.key_ln, A.key_kc,
                     chars, hits, F);
}

void launch_hits_unpack4(const DevAut &A, const uint32_t *stream_words, uint64_t n, int chars, int32_t *hits,
                         StreamFmt F, void *stream) {
  const uint64_t zero = 0;
  launch_hits_unpack4_segs(A, stream_words, &zero, &n, &zero, 1, chars, hits, F, stream);
}

// -------------------------------------------------- device-resident doc offsets
// The device entry point cannot read its doc offsets on the host: flag[0] |= 1 when they are not the offsets of
// n_docs documents over exactly n_bytes (doc_off[0] = 0, ascending, doc_off[D] = n_bytes), |= 2 when a document
// is 2^31 bytes or longer (Int32 offsets, src/aha/matcher.cr:3-5).  abort_word (the single-traversal pipelines' cursor[1]): the
// same verdict | 16 where the traversal and every post pass look before they index anything with the offsets -- the
// call then needs no round trip to the host in front of the traversal (the host reads the word with the totals).
__global__ __launch_bounds__(256) void k_check_docs(const uint64_t *doc_off, uint64_t D, uint64_t N, uint32_t *flag,
                                                    unsigned long long *abort_word) {
  uint32_t bad = 0;
  for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d <= D; d += (uint64_t)gridDim.x * 256) {
    const uint64_t q = doc_off[d];
    if (d == 0 && q != 0) bad |= 1u;
    if (d == D && q != N) bad |= 1u;
    if (d < D) {
      const uint64_t q1 = doc_off[d + 1];
      if (q1 < q) bad |= 1u;
      else if (q1 - q >= 0x7FFFFFFFull) bad |= 2u;
    }
  }
  if (bad && flag) atomicOr(flag, bad);
  if (bad && abort_word) atomicOr(abort_word, (unsigned long long)(16u | bad));
}

void launch_check_docs(const uint64_t *doc_off, uint64_t n_docs, uint64_t n_bytes, uint32_t *flag, unsigned long long *abort_word,
                       void *stream) {
  const uint32_t g = (uint32_t)std::min<uint64_t>((n_docs + 256) / 256, 1024);
  hipLaunchKernelGGL(k_check_docs, dim3(g), dim3(256), 0, (hipStream_t)stream, doc_off, n_docs, n_bytes, flag, abort_word);
}

// The verdict and the totals of a call (five words) into the caller's pinned host words, by a store from the device: the
// 40-byte hipMemcpyAsync it replaces is a blit kernel of 5 (64 MiB batch) to 17 us (1 GiB) on this stack.
__global__ void k_publish_words(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ host_dst, int n) {
  if ((int)threadIdx.x < n) host_dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
}
void launch_publish_words(const unsigned long long *src, unsigned long long *host_dst, int n, void *stream) {
  hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, (hipStream_t)stream, src, host_dst, n);
}

// ---------------------------------------------------------------- launchers
static inline uint32_t blocks_for(uint64_t n_chunks) {
  return (uint32_t)((n_chunks + kBlock - 1) / kBlock);
}

void launch_count(const DevAut &A, const MatchArgs &M, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  uint32_t g = blocks_for(M.n_chunks);
  if (A.compact)
    hipLaunchKernelGGL(k_count<true>, dim3(g), dim3(kBlock), 0, s, A, M);
  else
    hipLaunchKernelGGL(k_count<false>, dim3(g), dim3(kBlock), 0, s, A, M);
}

void launch_scan_blocks(const MatchArgs &M, uint64_t n_blocks, void *stream) {
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, (hipStream_t)stream, M.blk_hits,
                     M.chars ? M.blk_leads : nullptr, n_blocks, M.totals);
}

void launch_docg(const MatchArgs &M, void *stream) {
  uint64_t n = M.n_docs + 1;
  hipLaunchKernelGGL(k_docg, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M);
}

void launch_write(const DevAut &A, const MatchArgs &M, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  uint32_t g = blocks_for(M.n_chunks);
  if (A.compact)
    hipLaunchKernelGGL(k_write<true>, dim3(g), dim3(kBlock), 0, s, A, M);
  else
    hipLaunchKernelGGL(k_write<false>, dim3(g), dim3(kBlock), 0, s, A, M);
}

}  // namespace aha
