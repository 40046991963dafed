// unit.hpp -- the character-level ("unit") image of the same automaton (scan_unit.hip walks it).
//
// A UNIT is what a position of the text offers as one symbol:
//   b0 < 0x80 (not NUL)                                     the byte itself            code = b0
//   b0 in 0xC0..0xDF followed by one byte in 0x80..0xBF     a two-byte unit            code = 0x80  + (b0 & 31) << 6 | b1 & 63
//   b0 in 0xE0..0xEF followed by two bytes in 0x80..0xBF    a three-byte unit          code = 0x880 + (b0 & 15) << 12 | ...
//   anything else (NUL, a stray 0x80..0xBF, 0xF0.., a lead byte whose continuation bytes are missing or lie beyond
//   the end of the document)                                a one-byte unit that matches nothing ("bad")
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
// reference visits there, and reports exactly the same hits.  It takes one step per character instead of one per
// byte.  Ineligible key sets (a key that ends inside a unit, holds a stray continuation byte or a byte >= 0xF0)
// keep the byte-level engines.
//
// Image: an XOR double array with unique bases over 17-bit unit codes, 8-byte slots, at most 2^21 of them.  The
// entry that leads to a state also carries where that state fails to, so a miss needs no header load:
//   transition  lo = child base (21 bits) | low 10 bits of the child's fail base << 21 | END << 31
//               hi = code (17 bits) | high 11 bits of the fail base << 17 | FFR << 28   (FFR: the fail state's own
//                                                                                        fail link is the root)
//   header      (same fields, code 0, no child)  only for a state that is some state's fail target and does not fail
//               to the root itself: falling INTO it by a fail link is the one way to be in a state without having read
//               the entry that leads to it
// A fail base of 0 is the root (base 0, owns no slot).  The root's transitions live in a directly indexed table:
// root[code] = child base | filter << 21 | END << 31 (a depth-1 state fails to the root).  `filter` is an 8-bit Bloom
// filter over the unit codes the depth-1 state has transitions on (bit u_fbit(code)): most units that follow a
// character do not continue a key, and a clear bit answers that without the probe.
#pragma once

#include <cstdint>
#include <vector>

#include "automaton.hpp"

namespace aha {

constexpr uint32_t kUCode2 = 0x80u;              // first code of the two-byte units
constexpr uint32_t kUCode3 = 0x880u;             // first code of the three-byte units
constexpr uint32_t kUCodes = 0x880u + 0x10000u;  // codes are below this
constexpr uint32_t kUCodeBad = 0x1FFFFu;         // matches nothing
constexpr uint32_t kUMaxSlots = 1u << 21;

// decoding of an entry (device code and the CPU twin in tests/ use the same arithmetic)
AHA_HD inline uint32_t u_child(uint32_t lo) { return lo & 0x1FFFFFu; }
AHA_HD inline bool u_end(uint32_t lo) { return (lo >> 31) != 0; }
AHA_HD inline uint32_t u_code(uint32_t hi) { return hi & 0x1FFFFu; }
AHA_HD inline uint32_t u_fail(uint32_t lo, uint32_t hi) { return ((lo >> 21) & 0x3FFu) | (((hi >> 17) & 0x7FFu) << 10); }
AHA_HD inline bool u_ffr(uint32_t hi) { return ((hi >> 28) & 1u) != 0; }
AHA_HD inline uint32_t u_fbit(uint32_t code) { return (code ^ (code >> 4) ^ (code >> 9)) & 7u; }
AHA_HD inline uint32_t u_filter(uint32_t root_entry) { return (root_entry >> 21) & 0xFFu; }

struct UnitImage {
  bool ok = false;
  const char *why = "";
  uint32_t n_slots = 0;              // a multiple of 2^17 (a base and base ^ code share a block of 2^17 slots)
  std::vector<uint64_t> slots;       // [n_slots]
  std::vector<uint32_t> end_info;    // [n_slots] key id | min(chain length, 255) << 24 at the base of an END state
  std::vector<uint32_t> root;        // [kUCodes]
  uint32_t lo3 = 0, n3 = 0;          // three-byte units whose first byte is 0xE0 + lo3 .. 0xE0 + lo3 + n3 - 1 hold
                                     // nearly all root transitions: that part of root[] is kept in LDS
  uint32_t n_states = 0, n_trans = 0, n_headers = 0;
  uint32_t multi_permille = 0;       // share of the key bytes that lie in two- and three-byte units
};

// a: the byte-level automaton (build_automaton).  Fills u; u.ok = false + u.why when the key set is not eligible or
// (unless `force`) mostly made of one-byte units.
void build_unit(const Automaton &a, UnitImage &u, bool force = false);

}  // namespace aha
