// unit.hpp -- the character-level ("unit") image of the same automaton (scan_unit.hip walks it).
//
// Why (profiles/r03_trip_anatomy.txt): the byte-level trip is bound by VALU issue -- 63 vector instructions per input
// byte whatever the byte is.  UTF-8 text spends two of three trips inside a character, where nothing can be reported
// and nothing can fail interestingly.  This image has one state per CHARACTER boundary and takes one step per character.
//
// A UNIT is what a position of the text offers as one symbol:
//   b0 in 0x01..0x7F                                         the byte itself
//   b0 in 0xC0..0xDF followed by one byte in 0x80..0xBF      a two-byte unit,   payload (b0 & 31) << 6 | b1 & 63
//   b0 in 0xE0..0xEF followed by two bytes in 0x80..0xBF     a three-byte unit, payload (b0 & 15) << 12 | (b1 & 63) << 6 | b2 & 63
//   anything else (NUL, a stray 0x80..0xBF, 0xF0.., a lead byte whose continuation bytes are missing or lie beyond
//   the end of the document)                                 a one-byte unit that matches nothing ("bad")
// Every byte that is not in 0x80..0xBF starts a unit, whatever precedes it.
//
// Eligibility: every key is a sequence of good units (valid UTF-8 up to U+FFFF always is).  Then
//   * an occurrence of a key in ANY text starts at a byte outside 0x80..0xBF, i.e. at a unit start, and its units are
//     segmented in the text exactly as in the key (a unit's length follows from its first byte);
//   * a suffix of the text that is a trie path and ends at a unit boundary is therefore unit aligned, so the
//     byte-level automaton's state at every unit boundary (src/aha/ac.cr:176-192: the longest suffix that is a trie
//     path) is a state at a unit boundary of the keys, and its fail link -- the longest proper suffix that is a trie
//     path -- is one as well (a suffix that starts inside a unit starts with a byte in 0x80..0xBF: no key does);
//   * a state inside a unit ends no key, so no position inside a unit reports (ac.cr:183-185);
//   * a bad unit reports nothing either (its byte-level states are the root or states inside a unit), and after it no
//     trie path reaches back across it: a stray continuation byte or a byte >= 0xF0 is in no key, a lead byte without
//     its continuation bytes is followed by a byte no key has in that place, NUL takes the reference to the root
//     (ac.cr:188-189) -- so "state := root" gives the same states at all later unit boundaries.
// So the automaton over units -- states = the byte-level states at unit boundaries, goto by whole units, the same
// fail links, the same END states and output chains -- visits, at the unit boundaries, exactly the states the
// reference visits there, and reports exactly the same hits.
//
// SYMBOLS.  The kernel does not walk over raw payloads but over a dense alphabet, so that the root's transitions are
// one directly indexed LDS table: the one-byte units keep their byte (1..127; 0 = the bad unit), the two-byte payloads
// the keys use lie in one window [c2lo, c2lo + w2) and become 129 + (payload - c2lo), the three-byte payloads in a
// window [c3lo, c3lo + w3) become n2 + 1 + (payload - c3lo), n2 = 129 + w2.  A good unit outside its window is in no
// key: it becomes the class's "other" symbol (128, n2), which has no transition anywhere.  A = n2 + 1 + w3 symbols.
// Decoding is table driven (four LDS reads, a handful of adds and compares): T0[b0] gives the unit's length, where
// to look up its second and third byte (A1, A2: the payload contribution of a valid continuation byte, a poison
// value otherwise) and the class's first symbol and window width.
//
// IMAGE: an XOR double array with unique bases over the symbols, 8-byte slots, at most 2^22 of them.  A state is
// carried around as ONE word -- the low word of the entry that led to it:
//   transition  lo = child base (22 bits) | filter of the child (7 bits) << 22 | F1 << 29 | NFR << 30 | END << 31
//               hi = symbol (16 bits) | min(hits an event in the child stands for, 255) << 16  (0 unless the child is END)
//   root[symbol]     the same word for the root's transitions (in LDS); 0 = the root itself (base 0, owns no slot)
// `filter` is a 7-bit Bloom filter over the symbols the child has transitions on (bit symbol & 7; symbols with
// symbol & 7 = 7 always probe): most characters that follow a character do not continue a key, and a clear bit
// answers that without the probe.  NFR: the child's fail link is NOT the root.  A miss in a state without NFR (the
// root, every one-character state, most others) is answered by the root's table in the same trip.  A miss in an NFR
// state tries the unit again in the fail state: with F1 -- the fail state is a one-character state, the case of every
// two-character key -- that is root[the symbol that led here], an LDS read; else (rare: a partial match of three
// characters or more whose suffix is one of two or more) the state owns a HEADER: the slot of symbol 0 (no unit
// decodes to 0), slots[base] = {word of the fail state, 0}.  The walk fetches it in a trip of its own: it marks its
// state word as "header pending" (F1 set, NFR clear -- a combination no entry holds), and the next trip's probe is the
// header instead of the unit (round 4: a side array indexed by base, 9.7 MB for cfg 3 and requested beside the probe
// by every lane in such a state, was the second source of L2 misses of the walk).
//
// BIG STATES.  A state with kUBigDegree transitions or more (an ASCII or Cyrillic letter that starts hundreds of keys)
// cannot be fitted into a shared XOR array over a dense alphabet -- all its slots would have to be free at once.
// Round 3 gave each a region of 2^15 slots of its own (cfg 3: 60 states, 15.7 MB: most of the walk's L2 misses and
// five sixths of its fabric traffic).  Now a big state owns a private block of `big_block` slots (a power of two; its
// base is the block's first slot, at or beyond `big_lo`, so base ^ x = base + x for x below the block size):
//   base + s                       s < n_low: the transition on the LOW symbol s (one- and two-byte units), as anywhere else
//   base + g0 + (s >> 5)           s >= n_low: the GROUP RECORD of the 32 symbols around s,
//                                  {bit (s & 31): the state continues on s,  slot of the group's first child}
//   base + 0                       its header, if it has one
// (n_low is a multiple of 32, g0 = n_low - n_low / 32.)  The children on high symbols lie, in symbol order, in a run of
// free slots of the shared array, as ordinary transition entries.  A set bit does not consume the unit: the state
// word becomes "the state whose transition on s is slot t" = (t ^ s) with an all-ones filter, t = first child + the
// number of set bits below s, and the next trip's ordinary probe hits it.  That identity t ^ s is reserved like a
// base, so no real state's probe ever lands on a child's slot.  cfg 3: 60 blocks of 1024 slots (0.5 MB) and 33 k
// children in runs instead of 15.7 MB; the letters after a letter -- most of the hits at such states -- are low
// symbols and take one trip as before, a CJK character after a letter takes two when it continues a key (2.7 %).
#pragma once

#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace aha {

constexpr uint32_t kUMaxSlots = 1u << 23;  // the widest image: 23-bit bases (see BASE WIDTH)
constexpr uint32_t kUBigDegree = 48;     // transitions from which on a state gets a region of its own
constexpr uint32_t kUMaxSyms = 21756;    // root table (4 bytes per symbol) + decode tables + input rows + the waves' event buffers fit 160 KiB of
                                         // LDS (scan_unit.hip u_lds); < 2^15: a big state's region holds every symbol
constexpr uint32_t kUBias = 1u << 17;    // decode sums are kept non-negative
constexpr uint32_t kUPoison = 1u << 24;  // contribution of a byte that is not a continuation byte where one must be
// decode tables, in 32-bit words: T0a[256] {a1 byte offset, a2 byte offset, biased base, biased first symbol},
// T0b[256] {symbols in the class window, length}, A1[768], A2[512]
constexpr uint32_t kUT0a = 0, kUT0b = 1024, kUA1 = 1536, kUA2 = 2304, kUTabWords = 2816;

// decoding of a state word / an entry (device code and the CPU twin in tests/ use the same arithmetic)
// BASE WIDTH.  An image of at most 2^22 slots has 22-bit bases and a 7-bit filter (classes symbol & 7: 0..6 stored, 7
// always probes); a larger one (cfg 5: 1 M keys) 23-bit bases and a 6-bit filter (classes 0..5 stored, 6 and 7 always
// probe).  The flags stay at bits 29..31.
AHA_HD inline uint32_t u_child(uint32_t lo, uint32_t bb = 22u) { return lo & ((1u << bb) - 1u); }
AHA_HD inline uint32_t u_all_filter(uint32_t bb) { return ((1u << (29u - bb)) - 1u) << bb; }
// hits an event record can stand for: the count's low 26 - bb bits ride in the record's first word behind base and lane,
// the rest in the top five bits of its third word (hits of the chunk before the event: below 2^27)
constexpr uint32_t kUMaxC4 = 255u;
// ... but the fused expansion (ku_expand_groups) is built for events of a hit or two: with chains of 16 -- cfg 5 -- it takes
// 11.7 ms where the general post passes take 0.86 + 4.38 (its threads walk a chain serially through the staging
// windows), so key sets with longer chains keep the general passes
constexpr uint32_t kUFusedMaxChain = 15u;
AHA_HD inline uint32_t u_rec_x(uint32_t base, uint32_t lane, uint32_t n, uint32_t bb) {
  return base | lane << bb | (n & ((1u << (26u - bb)) - 1u)) << (bb + 6u);
}
AHA_HD inline uint32_t u_rec_z(uint32_t hits_before, uint32_t n, uint32_t bb) { return hits_before | (n >> (26u - bb)) << 27; }
AHA_HD inline uint32_t u_rec_n(uint32_t x, uint32_t z, uint32_t bb) { return (x >> (bb + 6u)) | (z >> 27) << (26u - bb); }
AHA_HD inline uint32_t u_rec_before(uint32_t z) { return z & 0x7FFFFFFu; }
AHA_HD inline bool u_f1(uint32_t lo) { return ((lo >> 29) & 1u) != 0; }
AHA_HD inline bool u_nfr(uint32_t lo) { return ((lo >> 30) & 1u) != 0; }
AHA_HD inline bool u_end(uint32_t lo) { return (lo >> 31) != 0; }
AHA_HD inline uint32_t u_sym(uint32_t hi) { return hi & 0xFFFFu; }
AHA_HD inline uint32_t u_c4(uint32_t hi) { return (hi >> 16) & 255u; }
AHA_HD inline bool u_hdr_pending(uint32_t lo) { return ((lo >> 29) & 3u) == 1u; }  // F1 without NFR: fetch the header next

// MARKS (scan_skip.hip, the skip-ahead traversal).  While the automaton's state is the root or a one-unit state, its next
// state is a function of the next two units alone: the two-unit state when they spell a trie path, else the one-unit state
// (or the root) of the second.  So a walk in such a state may jump to the next position where a two-unit path STARTS and
// consume that unit from the root -- provided no key is a single unit (nothing can be reported on the way).  A first,
// stateless kernel marks those positions: every unit start p whose units at p and behind it pass a blocked Bloom filter over
// the image's two-unit paths, keyed by the units' RAW bytes (a unit = its one to three bytes as a little-endian integer;
// no symbol decode).  False marks cost the walk a trip, a missing one would lose hits: the filter has no false negatives.
constexpr uint32_t kSkipKA = 0x9E3779u, kSkipKB = 0x85EBCBu;  // (kSkipKA: the default of UnitImage::pair_k1)
constexpr uint32_t kSkipMinLog2 = 10, kSkipMaxLog2 = 14;  // 4 .. 64 KiB of LDS beside ks_mark's input rows
// low 32 bits of the product of the operands' low 24 bits (v_mul_u32_u24 on the device)
AHA_HD inline uint32_t sk_mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
AHA_HD inline uint32_t sk_part(uint32_t c0) {  // the first unit's share of the pair's hash (carried to the next unit)
  const uint32_t g = sk_mul24(c0, kSkipKB);
  return (g >> 11) | (g << 21);
}
AHA_HD inline uint32_t sk_hash(uint32_t part, uint32_t c1, uint32_t k1 = kSkipKA) {
  const uint32_t h = sk_mul24(c1, k1) + part;
  return h ^ (h >> 16);
}
AHA_HD inline uint32_t sk_word(uint32_t h, uint32_t log2_words) { return h >> (32u - log2_words); }
AHA_HD inline uint32_t sk_mask(uint32_t h) { return (1u << (h & 31u)) | (1u << ((h >> 5) & 31u)); }

// PAIR TABLE (scan_pair.hip, the pair engine).  The two-unit paths once more, as a perfect hash table keyed by the units' raw
// bytes: what a stateless pass needs to turn "these two units pass the filter" into "this is the two-unit state X" with ONE
// 16-byte load -- {raw0 | hits of an event there << 24, raw1, event payload (0: the state ends no key), child filter}.  The
// slot of a pair: its hash (sk_hash with pair_k1, chosen so that no two pairs share the 32-bit value) picks a group, the
// group's displacement byte (LDS) the slot (hash-displace: groups in descending size, each displaced by the first byte that
// puts all its keys on free slots).  child filter: bit pt_cls(raw2) for every transition of the two-unit state -- a third
// unit whose bit is clear cannot continue it (no probe), one whose bit is set makes the position a DEEP CANDIDATE.
constexpr uint32_t kPairMix = 0x2545F491u, kPairKC = 0xC2B2AFu;
constexpr uint32_t kPairMaxGroups = 16384;  // displacement bytes in LDS
constexpr uint32_t kPairMaxDeepEnds = 3;    // END states of three units or more on one trie path that the pair engine has room for
AHA_HD inline uint32_t pt_group(uint32_t h, uint32_t n_groups) { return (h >> 7) & (n_groups - 1u); }
AHA_HD inline uint32_t pt_slot(uint32_t h, uint32_t d, uint32_t log2_slots) {
  const uint32_t t = h * kPairMix;
  return ((t >> (32u - log2_slots)) + d * ((t << 1) | 1u)) & ((1u << log2_slots) - 1u);
}
// child filter: two of 32 bits per third unit (a two-unit state with n transitions lets ~(2n / 32)^2 of the other units through)
AHA_HD inline uint32_t pt_cls(uint32_t raw) {
  const uint32_t g = sk_mul24(raw, kPairKC);
  return (1u << (g >> 27)) | (1u << ((g >> 22) & 31u));
}

struct UnitImage {
  bool ok = false;
  const char *why = "";
  uint32_t n_slots = 0;             // a multiple of 2^15 (a base and base ^ symbol share a block of 2^15 slots)
  uint32_t n_shared = 0;            // the shared XOR array (= big_lo); behind it the blocks of the big states
  uint32_t n_big = 0, big_block = 0;  // big states, slots of the block each of them owns
  uint32_t n_low = 0, g0 = 0;       // symbols below n_low index a big state's block directly, the others by groups of 32
  uint32_t base_bits = 22;          // 22 or 23 (BASE WIDTH)
  std::vector<uint64_t> slots;      // [n_slots]
  std::vector<int32_t> end_key;     // [n_slots] key id at the base of an END state, else -1
  std::vector<uint32_t> root;       // [n_syms]
  std::vector<uint32_t> tables;     // [kUTabWords] decode tables
  uint32_t n_syms = 0;              // A
  uint32_t n1 = 128, n2 = 0;        // "other" symbols of the two- and three-byte classes (0 is the bad unit)
  uint32_t c2lo = 0, w2 = 0, c3lo = 0, w3 = 0;
  uint32_t n_states = 0, n_trans = 0, n_nfr = 0;
  uint32_t multi_permille = 0;      // share of the key bytes that lie in two- and three-byte units
  // MARKS: the filter over the two-unit paths (empty: a key is a single unit, or there is no two-unit path)
  std::vector<uint32_t> mark_bloom;  // [1 << mark_log2]
  uint32_t mark_log2 = 0;
  uint32_t n_pairs = 0;              // two-unit paths
  uint32_t mark_fill_permille = 0;   // set bits of the filter, per 1000
  bool unit_key = false;             // a key of one unit: every position could report -- no skipping
  uint32_t pair_k1 = kSkipKA;        // the second unit's multiplier in sk_hash (marks and pair table use the same hash)
  // PAIR TABLE (empty: no marks, a displacement that does not fit a byte, more pairs than the groups serve)
  std::vector<uint32_t> pair_tab;    // [4 << pair_log2]: {raw0 | c4 << 24, raw1, key id of the END state or ~0 (capi.cpp turns it into
                                     // the event payload), child filter}; a free slot is all zero (no unit is 0)
  std::vector<uint8_t> pair_disp;    // [pair_groups]
  uint32_t pair_log2 = 0, pair_groups = 0;
  uint32_t deep_ends_max = 0;        // most END states of three units or more on one trie path
};

// a: the byte-level automaton (build_automaton).  Fills u; u.ok = false + u.why when the key set is not eligible or
// (unless `force`) mostly made of one-byte units.
void build_unit(const Automaton &a, UnitImage &u, bool force = false);

}  // namespace aha
