// hash.hpp -- the HASH image of the character-level automaton (scan_hash.hip walks it).
//
// Why (profiles/r05_hash_walk_proto.txt): ku_traverse is bound by instruction issue, ~170 instructions per character and
// wave, and a third of them turn bytes into symbols of a dense alphabet (three dependent LDS reads) so that the root's
// transitions can be one directly indexed LDS table and a state's transitions one XOR double array.  A walk keyed by the
// characters' RAW BYTES needs neither: a character is its (at most three) UTF-8 bytes as a little-endian integer, a
// one-character state is the character itself, and every transition out of a deeper or one-character state is an entry
// of a hash table keyed by (parent, character).  What random text mostly asks -- "do the last two characters spell a
// two-character trie path?" -- is answered by a blocked Bloom filter over those pairs in LDS (64 KiB) without leaving
// the CU; the prototype walks 1 GiB of cfg 3's text that way in 0.67 ms against 1.26 ms for ku_traverse's trip without
// probes and events.
//
// It is the SAME automaton as the unit image's (unit.hpp: states at character boundaries, the same fail links, END
// states, output chains; src/aha/ac.cr:176-192 visits exactly these states at the character boundaries) with the same
// state identities -- the unit image's bases -- so the event records, uend[] and every post pass are shared.  A state is
// carried as the unit image's word: base (22 bits) | F1 << 29 | NFR << 30 | END << 31 (the filter bits of the unit image
// are not used here), plus a 32-bit CHILD FILTER: bit h_cls(c) is set when the state has a transition on a character of
// that class (a clear bit answers "no" without a probe).
//
//   PAIRS    transitions out of the one-character states, keyed (kHTag | first character, second character): a perfect
//            hash (groups of keys displaced by one byte each, the displacement table in LDS) -- ONE 16-byte load per
//            probe, behind the Bloom filter
//   DEEP     transitions out of the states of two characters or more, keyed (base of the state, character), and the
//            HEADERS (base, kHHdr) of the states whose fail state is neither the root nor a one-character state:
//            a cuckoo table, both candidate slots loaded at once
//   entry    {parent, character | hits an event in the child stands for << 24, word of the child (header: of the fail
//            state, without END), child filter of that state}
//
// One-character keys would need a lookup per character; key sets that hold one keep the unit image's walk.
#pragma once

#include <cstdint>
#include <vector>

#include "automaton.hpp"
#include "unit.hpp"

namespace aha {

constexpr uint32_t kHTag = 1u << 24;       // parent field of a pair: kHTag | the first character
constexpr uint32_t kHHdr = 0xFFFFFFu;      // "character" of a header entry (no UTF-8 character is 0xFFFFFF)
constexpr uint32_t kHK2 = 0x85EBCBu;       // multiplier of the parent's part (and of the filter class)
constexpr uint32_t kHMix = 0x2545F491u, kHMix2 = 0x9E3779B1u;
constexpr uint32_t kHBloomLog2 = 14;       // 2^14 words = 64 KiB
constexpr uint32_t kHMaxGroups = 16384;    // displacement bytes in LDS
constexpr uint32_t kHBaseSalt = 0x5BD1E995u;

// low 32 bits of the product of the operands' low 24 bits (v_mul_u32_u24 on the device: hipcc sees the masks)
AHA_HD inline uint32_t h_mul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
AHA_HD inline uint32_t h_rot(uint32_t g) { return (g >> 11) | (g << 21); }
AHA_HD inline uint32_t h_cls(uint32_t c) { return h_mul24(c, kHK2) >> 27; }                                // filter class of a character
AHA_HD inline uint32_t h_part_char(uint32_t c) { return h_rot(h_mul24(c, kHK2)); }                         // parent = the one-character state of c
AHA_HD inline uint32_t h_part_base(uint32_t b) { return h_rot(h_mul24(b, kHK2)) ^ kHBaseSalt; }            // parent = the state with base b
AHA_HD inline uint32_t h_key(uint32_t part, uint32_t c, uint32_t k1) {
  const uint32_t h = h_mul24(c, k1) + part;
  return h ^ (h >> 16);
}
AHA_HD inline uint32_t h_bloom_word(uint32_t h) { return h >> (32u - kHBloomLog2); }
AHA_HD inline uint32_t h_bloom_mask(uint32_t h) { return (1u << (h & 31u)) | (1u << ((h >> 5) & 31u)); }
// pairs: group, first slot and step of the displacement
AHA_HD inline uint32_t h_group(uint32_t h, uint32_t n_groups) { return (h >> 7) & (n_groups - 1u); }
AHA_HD inline uint32_t h_pair_slot(uint32_t h, uint32_t d, uint32_t log2_slots) {
  const uint32_t t = h * kHMix;
  return ((t >> (32u - log2_slots)) + d * ((t << 1) | 1u)) & ((1u << log2_slots) - 1u);
}
// deep: the two candidate slots
AHA_HD inline uint32_t h_deep_slot1(uint32_t h, uint32_t log2_slots) { return (h * kHMix) >> (32u - log2_slots); }
AHA_HD inline uint32_t h_deep_slot2(uint32_t h, uint32_t log2_slots) { return (h * kHMix * kHMix2) >> (32u - log2_slots); }

struct HEntry {
  uint32_t parent;  // kHTag | first character, or the base of the parent state
  uint32_t ch;      // character (kHHdr: header) | hits an event in the child stands for << 24
  uint32_t word;    // the child as one word (header: the fail state, END cleared)
  uint32_t cf;      // child filter of that state
};

struct HashImage {
  bool ok = false;
  const char *why = "";
  uint32_t k1 = 0x9E3779u;           // multiplier of the character (chosen so that no two pairs share a hash)
  std::vector<uint32_t> bloom;       // [1 << kHBloomLog2]
  std::vector<uint8_t> disp;         // [n_groups]
  std::vector<HEntry> pairs;         // [1 << pair_log2]; parent = 0: free
  std::vector<HEntry> deep;          // [1 << deep_log2]
  uint32_t n_groups = 0, pair_log2 = 0, deep_log2 = 0;
  uint32_t n_pairs = 0, n_deep = 0;
  uint32_t bloom_fill_permille = 0;
};

// u: the unit image with its transition list (UnitImage::htrans).  Fills h; h.ok = false + h.why when the key set keeps the
// unit image's walk (a one-character key, 23-bit bases, a Bloom filter more than half full, ...).
void build_hash(const UnitImage &u, HashImage &h);

}  // namespace aha
