// pp.hpp -- the position-parallel engine (scan_pp.hip): filter definition shared by the host
// builder (automaton.cpp) and the HIP kernels.
//
// Exactness argument (reference semantics: src/aha/ac.cr:176-192 match_, :265-278 fetch).
// The AC state after byte i is the longest suffix of text[doc_start..i] that is a trie path.  Let
// W(j) be the trie walk from the root along text[j..] (inside j's document), L(j) its length.  The
// state after byte i is the node of depth i-j*+1 on W(j*), j* = the smallest j with j + L(j) > i.
// So position i reports (is_end?, ac.cr:183-185) exactly when some start j has an END node at depth
// d = i-j+1 on W(j) and NO earlier start j' < j of the same document has j' + L(j') > i.  Walks of
// different starts are independent: every start position can be examined in parallel, and only the
// "no earlier start is still alive" test couples them (a prefix maximum of j' + L(j')).
//
// Pass 1 (k_pp_filter) proves most starts BORING with two kinds of LDS lookups:
//   T2      direct table on the two bytes at j: bit0 = the pair is a trie path with children
//           ("deep"), bit1 = the pair is a key (END at depth 2).  Exact.
//   Bloom   blocked Bloom filter (one bit in each byte of a 32-bit word, no false negatives) over
//           E3 = {3-byte keys}, E4 = {4-byte keys}, P5 = {trie paths of depth 5}, probed with the
//           3 / 4 / 5 bytes at j.
// A start that is not deep, or deep with all three probes negative and no END at depth 2, has no END
// node on its walk and L(j) <= 4: it can never report, and it can only suppress a report that ends
// at most 3 bytes after it.  Every other start becomes an ITEM.
// Pass 2 (k_pp_resolve) walks the items exactly in the device image (goto probes only, cedar.cr:441-447),
// takes the prefix maximum of their reaches, re-checks the at most three boring starts in front of a
// candidate exactly, and writes the surviving events in position order into the chunk's event
// region -- the same records the single-traversal engine produces, so the chain expansion (fetch)
// and the per-document offsets are shared with it.
//
// Preconditions (else the handle uses the single-traversal engine): no 1-byte key, longest key
// <= kPpMaxKeyLen bytes, compact slot format, and a Bloom load that keeps false positives rare.
#pragma once
#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace aha {

constexpr uint32_t kPpChunk = 4096;          // bytes per chunk (item list / event region)
constexpr uint32_t kPpItemCap = kPpChunk / 8;  // items per chunk
constexpr uint32_t kPpEvStride = kPpChunk / 8; // events per chunk region
constexpr uint32_t kPpHalo = 256;             // a start reaches at most this far: kPpMaxKeyLen < kPpHalo
constexpr uint32_t kPpDeepCap = 64;           // starts whose walk is alive at depth kPpGuard, per chunk
constexpr uint32_t kPpLongCap = 64;           // starts with walks of kPpGuard bytes and more, per chunk
constexpr uint32_t kPpMaxKeyLen = 240;       // reaches are bytes; the halo in front of a chunk is 256 bytes
constexpr uint32_t kPpGuard = 5;             // P-probe depth; boring starts have L < kPpGuard
constexpr uint32_t kPpT2Words = 4096;        // 64 Ki entries x 2 bit
constexpr uint32_t kPpBloomWords = 28672;   // 112 KiB of the 160 KiB LDS (T2 16 KiB, rings and item buffers 32 KiB)
constexpr uint32_t kPpK1 = 0x9E3779u, kPpK2 = 0x85EBCBu, kPpK3 = 0xC2B2AFu;

// item word: bits 0..11 position in the chunk, 12 = E3 positive, 13 = E4 positive, 14 = P5 positive,
// 15 = END at depth 2
constexpr uint32_t kPpItemE3 = 0x1000u, kPpItemE4 = 0x2000u, kPpItemP5 = 0x4000u, kPpItemEnd2 = 0x8000u;

// T2 entry of the pair (b0, b1): word b0 | (b1 & 15) << 8, 2-bit field (b1 >> 4)
AHA_HD inline uint32_t pp_t2_word(uint32_t b0, uint32_t b1) { return b0 | ((b1 & 15u) << 8); }
AHA_HD inline uint32_t pp_t2_shift(uint32_t b1) { return (b1 >> 4) * 2u; }

// hashes of the window at a start: lo = bytes 0..3 (little endian), b4 = byte 4
struct PpHash {
  uint32_t m1, h3, h4, h5;
};
AHA_HD inline PpHash pp_hash(uint32_t lo, uint32_t b4) {
  PpHash h;
  h.m1 = (lo & 0xFFFFFFu) * kPpK1;                 // bytes 0..2
  h.h3 = ((lo >> 8) & 0xFFFFu) * kPpK2 + h.m1;     // bytes 0..2
  h.h4 = (lo >> 8) * kPpK2 + h.m1;                 // bytes 0..3
  h.h5 = (b4 & 0xFFu) * kPpK3 + h.h4;              // bytes 0..4
  return h;
}
// word index in [0, words), words < 65536: ((h >> 8) * words) >> 24 -- one v_mul_hi_u32_u24 on the device
AHA_HD inline uint32_t pp_word(uint32_t h, uint32_t words) {
  return (uint32_t)(((uint64_t)(h >> 8) * (uint64_t)((words << 8) & 0xFFFFFFu)) >> 32);
}
// one bit in each byte of the word, from 12 bits of a hash of the first three bytes (shared by the three probes)
AHA_HD inline uint32_t pp_mask(uint32_t m1) {
  const uint32_t g = m1 ^ (m1 >> 11);
  return (1u << (g & 7u)) | (0x100u << ((g >> 8) & 7u)) | (0x10000u << ((g >> 16) & 7u)) |
         (0x1000000u << ((g >> 24) & 7u));
}

struct PpTables {
  bool ok = false;              // the automaton meets the preconditions
  const char *why = "";         // reason when !ok
  std::vector<uint32_t> t2;     // [kPpT2Words]
  std::vector<uint32_t> bloom;  // [words], words < 65536
  uint64_t n_entries = 0;       // E3 + E4 + P5
  double fill = 0.0;            // fraction of Bloom bits set
};
// words = 0: as many as fit the LDS budget of the filter kernel (kPpBloomWords)
void build_pp(const Automaton &a, bool compact, uint32_t words, PpTables &t);

}  // namespace aha
