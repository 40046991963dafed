// automaton.hpp -- host-side (CPU, C++17) construction of the Aho-Corasick
// automaton the HIP kernels traverse.  This is product code: it re-derives,
// from the keys alone, the automaton whose *observable* behaviour is that of
// the reference's Aha::AC.compile (src/aha/ac.cr:62-112) -- goto = trie edges,
// fail = standard AC failure links (ac.cr:94-105), output chain truncated at
// the first non-end fail ancestor (ac.cr:106-108, 265-278).  Node ids are not
// observable through #match, so the build uses its own BFS numbering and its
// own double-array placement (hot shallow states first) instead of Cedar's.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define AHA_HD __host__ __device__
#else
#define AHA_HD
#endif

namespace aha {

// Abstract automaton, states numbered in BFS order (0 = root).  The children
// of a state are consecutive ids, sorted by label.
struct Automaton {
  uint32_t n_states = 0;
  uint32_t n_keys = 0;
  uint32_t max_key_len = 0;

  std::vector<uint32_t> first_child;  // [n_states]   id of first child
  std::vector<uint16_t> n_child;      // [n_states]   number of children (<=255)
  std::vector<uint8_t> in_label;      // [n_states]   label on the edge parent->s
  std::vector<uint32_t> parent;       // [n_states]
  std::vector<uint32_t> fail;         // [n_states]
  std::vector<int32_t> key_of;        // [n_states]   key ending here, or -1
  std::vector<uint32_t> depth;        // [n_states]   bytes from the root

  // per key (emission tables; index = key id = Hit#value)
  std::vector<uint32_t> key_len;    // bytes                      (ac.cr:89-93 key_lens)
  std::vector<int32_t> key_next;    // next key on the output chain or -1 (ac.cr:106-108)
  std::vector<uint32_t> key_cnt;    // hits emitted when this key's state is reached
  std::vector<uint32_t> key_kc;     // #UTF-8 lead bytes in key[1..len) (char-offset mode)
  std::vector<uint32_t> key_state;  // state id where the key ends

  // key storage for AC#[](id) / AC#[](key)
  std::vector<uint8_t> blob;
  std::vector<uint64_t> offs;

  // child(s, label) or UINT32_MAX
  uint32_t child(uint32_t s, uint8_t label) const;
  // key id of an exact key, or -1
  int32_t find_key(const uint8_t *key, int64_t len) const;
};

// Error codes match include/aha_hip.h.
struct BuildError {
  int32_t code = 0;
  uint32_t key_index = 0;
};

// Builds the abstract automaton.  Returns false and fills err on the first
// (lowest index) key at which the reference would raise.
bool build_automaton(const uint8_t *blob, const uint64_t *offs, uint32_t n_keys, Automaton &out,
                     BuildError &err);

// XOR double-array placement with UNIQUE bases: state s owns slot base[s]
// (its header; label 0 is never a goto because keys hold no NUL byte,
// cedar.cr:235) and slot base[s]^label for each child.  Because bases are
// unique, "slot.label == label" is a complete ownership check.
// Placement is depth-segmented for the shallow levels: when the BFS reaches
// depth d <= kSegDepth a fresh 256-slot block is started and earlier blocks
// are closed, so  seg_start[d] <= base[s]  <=>  depth(s) >= d  (d <= kSegDepth)
// and every slot below seg_start[d] belongs to a state of depth < d.
constexpr uint32_t kSegDepth = 5;
struct Placement {
  std::vector<uint32_t> base;  // [n_states]
  uint32_t n_slots = 0;        // multiple of 256
  uint32_t seg_start[kSegDepth + 2] = {0};  // first slot of depth d; [kSegDepth+1] = n_slots sentinel if shallower
  // defer_deep_fail: states of depth >= 3 whose fail target has depth >= 3 are placed last, from this slot on
  // (the seg_start equivalence then holds for d <= 3 only); n_slots when there is none or the option is off
  uint32_t deep_fail_start = 0;
  // headerless (needs defer_deep_fail): only the root and the deep-fail states own a header slot.  Every other
  // fail link is a function of the last two input bytes (depth 1: root; depth 2: the depth-1 state of the last
  // byte; depth >= 3 below deep_fail_start: the depth<=2 state of the last two bytes), so its slot is never read:
  // the base is then a pure identity (unique, but it may coincide with another state's transition slot), the
  // image holds little more than one slot per transition, and full 64-label rows pack four to a block.
  bool headerless = false;
};
void place_states(const Automaton &a, Placement &p, bool defer_deep_fail = false, bool headerless = false);
bool needs_header(const Automaton &a, const Placement &p, uint32_t s);

// ---- device image formats -------------------------------------------------
// Wide slot (8 bytes, one 64-bit load):
//   transition: lo = child_base | W_END (child ends a key) | W_FAILROOT (child's fail is root)
//               hi = label | key_id << 8   (key id of the child when W_END)
//   header    : lo = fail_base ; hi = 0
//   free      : 0
constexpr uint32_t W_END = 0x80000000u;
constexpr uint32_t W_FAILROOT = 0x40000000u;
constexpr uint32_t W_BASE_MASK = 0x3FFFFFFFu;

// Compact slot (4 bytes): bits 0..7 label, bits 8..29 base (22 bits),
// bit 30 = child's fail is root, bit 31 = child ends a key.  Header: label 0,
// base field = fail base.  The key id of an end state lives in the side
// array end_key[base] (only read on emission).
constexpr uint32_t C_END = 0x80000000u;
constexpr uint32_t C_FAILROOT = 0x40000000u;
constexpr uint32_t C_BASE_SHIFT = 8;
constexpr uint32_t C_BASE_MASK = 0x3FFFFFu;
constexpr uint32_t C_MAX_SLOTS = 1u << 22;

struct Image {
  bool compact = false;
  uint32_t n_slots = 0;
  uint32_t root_base = 0;
  std::vector<uint64_t> wide;      // [n_slots] when !compact
  std::vector<uint32_t> narrow;    // [n_slots] when compact
  std::vector<int32_t> end_key;    // [n_slots] when compact: key id at header slots of end states, else -1
};
// Returns false if the automaton does not fit the format limits.
bool encode_image(const Automaton &a, const Placement &p, bool force_wide, Image &img);

}  // namespace aha
