// automaton.cpp -- see automaton.hpp.
#include "automaton.hpp"

#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>

#include "../../include/aha_hip.h"

namespace aha {

uint32_t Automaton::child(uint32_t s, uint8_t label) const {
  uint32_t lo = first_child[s], n = n_child[s];
  const uint8_t *l = in_label.data() + lo;
  // children are sorted by label
  const uint8_t *e = l + n;
  const uint8_t *it = std::lower_bound(l, e, label);
  if (it != e && *it == label) return lo + (uint32_t)(it - l);
  return UINT32_MAX;
}

int32_t Automaton::find_key(const uint8_t *key, int64_t len) const {
  if (len <= 0) return -1;
  uint32_t s = 0;
  for (int64_t i = 0; i < len; i++) {
    s = child(s, key[i]);
    if (s == UINT32_MAX) return -1;
  }
  return key_of[s];
}

namespace {
struct Range {
  uint32_t lo, hi;  // range of sorted key positions sharing the node's prefix
};
}  // namespace

bool build_automaton(const uint8_t *blob, const uint64_t *offs, uint32_t K, Automaton &a,
                     BuildError &err) {
  a = Automaton();
  a.n_keys = K;
  a.offs.assign(offs, offs + K + 1);
  a.blob.assign(blob + offs[0], blob + offs[K]);
  for (auto &o : a.offs) o -= offs[0];
  const uint8_t *B = a.blob.data();
  auto klen = [&](uint32_t k) { return (uint32_t)(a.offs[k + 1] - a.offs[k]); };
  auto kptr = [&](uint32_t k) { return B + a.offs[k]; };

  // The reference inserts keys in order and raises at the first offending
  // index: empty key (cedar.cr:756), NUL byte (cedar.cr:235), duplicate
  // (ac.cr:66).  Find the lowest such index.
  uint32_t bad = UINT32_MAX;
  int32_t bad_code = 0;
  for (uint32_t k = 0; k < K; k++) {
    uint32_t n = klen(k);
    if (n == 0) {
      bad = k;
      bad_code = AHA_E_EMPTY_KEY;
      break;
    }
    if (n >= 0x7FFFFFFFu) {  // key_lens masks 31 bits (ac.cr:237)
      bad = k;
      bad_code = AHA_E_TOO_LARGE;
      break;
    }
    if (memchr(kptr(k), 0, n)) {
      bad = k;
      bad_code = AHA_E_ZERO_BYTE;
      break;
    }
  }
  uint32_t K_ok = bad == UINT32_MAX ? K : bad;  // keys before the first bad one

  std::vector<uint32_t> order(K_ok);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    uint32_t nx = klen(x), ny = klen(y);
    int c = memcmp(kptr(x), kptr(y), std::min(nx, ny));
    if (c != 0) return c < 0;
    if (nx != ny) return nx < ny;
    return x < y;
  });
  // duplicates are adjacent; the reference raises at the later index
  for (uint32_t i = 1; i < K_ok; i++) {
    uint32_t x = order[i - 1], y = order[i];
    if (klen(x) == klen(y) && memcmp(kptr(x), kptr(y), klen(x)) == 0) {
      if (y < bad) {
        bad = y;
        bad_code = AHA_E_DUP_KEY;
      }
    }
  }
  if (bad != UINT32_MAX) {
    err.code = bad_code;
    err.key_index = bad;
    return false;
  }

  // ---- BFS trie construction over the sorted keys -------------------------
  uint64_t total = a.offs[K];
  size_t reserve = (size_t)std::min<uint64_t>(total + 1, 0xFFFFFFF0ull);
  a.first_child.reserve(reserve);
  a.n_child.reserve(reserve);
  a.in_label.reserve(reserve);
  a.parent.reserve(reserve);
  a.key_of.reserve(reserve);
  std::vector<Range> ranges;
  ranges.reserve(reserve);
  std::vector<uint32_t> &depth = a.depth;
  depth.reserve(reserve);

  auto push_state = [&](uint32_t par, uint8_t label, uint32_t d, Range r) {
    a.first_child.push_back(0);
    a.n_child.push_back(0);
    a.in_label.push_back(label);
    a.parent.push_back(par);
    a.key_of.push_back(-1);
    depth.push_back(d);
    ranges.push_back(r);
  };
  push_state(0, 0, 0, Range{0, K});
  a.key_state.assign(K, 0);
  for (uint32_t s = 0; s < a.first_child.size(); s++) {
    Range r = ranges[s];
    uint32_t d = depth[s];
    uint32_t lo = r.lo;
    // the shortest key of the range sorts first; it ends here iff len == d
    if (lo < r.hi && klen(order[lo]) == d) {
      a.key_of[s] = (int32_t)order[lo];
      a.key_state[order[lo]] = s;
      lo++;
    }
    a.first_child[s] = (uint32_t)a.first_child.size();
    uint32_t nc = 0;
    while (lo < r.hi) {
      uint8_t lab = kptr(order[lo])[d];
      uint32_t e = lo + 1;
      while (e < r.hi && kptr(order[e])[d] == lab) e++;
      if (a.first_child.size() >= 0xFFFFFFF0ull) {
        err.code = AHA_E_TOO_LARGE;
        return false;
      }
      push_state(s, lab, d + 1, Range{lo, e});
      nc++;
      lo = e;
    }
    a.n_child[s] = (uint16_t)nc;
  }
  a.n_states = (uint32_t)a.first_child.size();
  ranges.clear();
  ranges.shrink_to_fit();

  // ---- failure links, standard BFS (ac.cr:79-105) --------------------------
  a.fail.assign(a.n_states, 0);
  for (uint32_t s = 1; s < a.n_states; s++) {
    uint32_t p = a.parent[s];
    if (p == 0) {
      a.fail[s] = 0;
      continue;
    }
    uint8_t lab = a.in_label[s];
    uint32_t f = a.fail[p];
    uint32_t t;
    for (;;) {
      t = a.child(f, lab);
      if (t != UINT32_MAX || f == 0) break;
      f = a.fail[f];
    }
    a.fail[s] = (t == UINT32_MAX) ? 0 : t;
  }

  // ---- emission tables (ac.cr:89-93, 106-108, 265-278) ---------------------
  a.key_len.assign(K, 0);
  a.key_next.assign(K, -1);
  a.key_cnt.assign(K, 0);
  a.key_kc.assign(K, 0);
  a.max_key_len = 0;
  for (uint32_t s = 1; s < a.n_states; s++) {  // BFS order: fail[s] is shallower, already done
    int32_t k = a.key_of[s];
    if (k < 0) continue;
    uint32_t n = klen((uint32_t)k);
    a.key_len[k] = n;
    a.max_key_len = std::max(a.max_key_len, n);
    int32_t nk = a.key_of[a.fail[s]];
    a.key_next[k] = nk;
    a.key_cnt[k] = 1 + (nk >= 0 ? a.key_cnt[nk] : 0);
    uint32_t kc = 0;
    const uint8_t *p = kptr((uint32_t)k);
    for (uint32_t i = 1; i < n; i++) kc += (p[i] & 0xC0) != 0x80;
    a.key_kc[k] = kc;
  }
  return true;
}

// ---------------------------------------------------------------- placement
namespace {
using Mask = std::array<uint64_t, 4>;

inline uint64_t perm6(uint64_t x, unsigned c) {
  // result bit i = x bit (i ^ c)
  if (c & 1) x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
  if (c & 2) x = ((x & 0x3333333333333333ull) << 2) | ((x >> 2) & 0x3333333333333333ull);
  if (c & 4) x = ((x & 0x0F0F0F0F0F0F0F0Full) << 4) | ((x >> 4) & 0x0F0F0F0F0F0F0F0Full);
  if (c & 8) x = ((x & 0x00FF00FF00FF00FFull) << 8) | ((x >> 8) & 0x00FF00FF00FF00FFull);
  if (c & 16) x = ((x & 0x0000FFFF0000FFFFull) << 16) | ((x >> 16) & 0x0000FFFF0000FFFFull);
  if (c & 32) x = (x << 32) | (x >> 32);
  return x;
}

// positions x such that x^l is free for all labels l
inline Mask children_free(const Mask &freeb, const uint8_t *labels, unsigned n) {
  Mask m = Mask{~0ull, ~0ull, ~0ull, ~0ull};
  for (unsigned i = 0; i < n; i++) {
    unsigned l = labels[i];
    unsigned wsel = l >> 6, c = l & 63;
    for (unsigned j = 0; j < 4; j++) m[j] &= perm6(freeb[j ^ wsel], c);
    if (!(m[0] | m[1] | m[2] | m[3])) break;
  }
  return m;
}
inline int first_bit(const Mask &m) {
  for (int j = 0; j < 4; j++)
    if (m[j]) return j * 64 + __builtin_ctzll(m[j]);
  return -1;
}
inline int popcnt(const Mask &m) {
  return __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]) +
         __builtin_popcountll(m[3]);
}
}  // namespace

bool needs_header(const Automaton &a, const Placement &p, uint32_t s) {
  if (!p.headerless || s == 0) return true;
  return a.depth[s] >= 3 && a.depth[a.fail[s]] >= 3;  // deep-fail: the only fail links that are looked up
}

void place_states(const Automaton &a, Placement &p, bool defer_deep_fail, bool headerless) {
  p.headerless = headerless && defer_deep_fail;
  const uint32_t S = a.n_states;
  p.base.assign(S, 0);
  std::vector<Mask> freeb;        // per block free bitmap
  std::vector<Mask> baseb;        // per block: positions that are not yet the base of a state
  std::vector<uint16_t> nfree;    // per block free count
  std::vector<uint16_t> nbase;    // per block: base ids left
  std::vector<uint8_t> nfail;     // failed placement attempts
  std::vector<uint32_t> open;     // blocks tried for multi-slot states (oldest first)
  std::vector<uint32_t> leafpool; // retired blocks that still have single free slots
  std::vector<uint32_t> basepool; // blocks without free slots that still have base ids (header-less leaves)
  size_t open_head = 0;
  constexpr unsigned TRIES = 6, MAXFAIL = 6;

  auto new_block = [&]() -> uint32_t {
    freeb.push_back(Mask{~0ull, ~0ull, ~0ull, ~0ull});
    baseb.push_back(Mask{~0ull, ~0ull, ~0ull, ~0ull});
    nfree.push_back(256);
    nbase.push_back(256);
    nfail.push_back(0);
    return (uint32_t)freeb.size() - 1;
  };
  auto take = [&](uint32_t blk, unsigned x) {
    freeb[blk][x >> 6] &= ~(1ull << (x & 63));
    nfree[blk]--;
  };

  // Placement order: level by level (BFS ids are grouped by depth), and inside a
  // level the states with the most keys below them first.  Text that resembles the
  // keys visits those states most, so the LDS prefix (and, deeper, the same cache
  // lines) holds the rows that are probed most instead of the first labels.
  std::vector<uint32_t> weight(S, 0), order(S);
  for (uint32_t s = S; s-- > 0;) {
    weight[s] += a.key_of[s] >= 0 ? 1u : 0u;
    if (s) weight[a.parent[s]] += weight[s];
  }
  // States of depth >= 3 whose fail link is itself deeper than 2 ("deep-fail"; rare) go last, into their own
  // region: for every other state of depth >= 3 the fail target is a function of the last two bytes, which the
  // traversal keeps at hand, so base in [seg_start[3], deep_fail_start) means "no fail header needed".
  order.clear();
  std::vector<uint32_t> deferred;
  for (uint32_t lo = 0; lo < S;) {
    uint32_t hi = lo;
    while (hi < S && a.depth[hi] == a.depth[lo]) hi++;
    const size_t at = order.size();
    for (uint32_t s = lo; s < hi; s++) {
      if (defer_deep_fail && a.depth[s] >= 3 && a.depth[a.fail[s]] >= 3)
        deferred.push_back(s);
      else
        order.push_back(s);
    }
    std::stable_sort(order.begin() + at, order.end(), [&](uint32_t x, uint32_t y) { return weight[x] > weight[y]; });
    lo = hi;
  }
  const size_t n_regular = order.size();
  order.insert(order.end(), deferred.begin(), deferred.end());
  p.deep_fail_start = 0;

  uint32_t cur_depth = 0;
  for (uint32_t d = 0; d < kSegDepth + 2; d++) p.seg_start[d] = 0;
  for (uint32_t oi = 0; oi < S; oi++) {
    const uint32_t s = order[oi];
    const unsigned nc = a.n_child[s];
    const uint8_t *labels = a.in_label.data() + a.first_child[s];
    if (oi == n_regular) {  // the deep-fail region starts in fresh blocks
      open.clear();
      open_head = 0;
      leafpool.clear();
      basepool.clear();
      p.deep_fail_start = (uint32_t)freeb.size() * 256u;
    }
    if (oi < n_regular && a.depth[s] != cur_depth) {
      // BFS reached the next level: for the shallow levels close every open
      // block so that deeper states never fill holes of shallower regions
      cur_depth = a.depth[s];
      if (cur_depth <= kSegDepth + 1) {
        open.clear();
        open_head = 0;
        leafpool.clear();
        basepool.clear();
        p.seg_start[cur_depth] = (uint32_t)freeb.size() * 256u;
      }
    }
    // A state owns a unique base (its identity) and the slots base^label of its children; the slot at the base
    // itself (the header: fail link) only where fail links are looked up (needs_header).
    const bool hdr = needs_header(a, p, s);
    const unsigned need = nc + (hdr ? 1u : 0u);
    auto cand = [&](uint32_t b) -> Mask {
      Mask m = baseb[b];
      if (hdr)
        for (int j = 0; j < 4; j++) m[j] &= freeb[b][j];
      if (nc) {
        Mask c = children_free(freeb[b], labels, nc);
        for (int j = 0; j < 4; j++) m[j] &= c[j];
      }
      return m;
    };
    uint32_t blk = UINT32_MAX;
    int x = -1;
    if (need == 0) {
      // a header-less leaf needs nothing but a base id: drain blocks that are full of slots but not of ids
      while (!basepool.empty() && nbase[basepool.back()] == 0) basepool.pop_back();
      if (!basepool.empty()) {
        blk = basepool.back();
        x = first_bit(baseb[blk]);
      }
    }
    if (need == 1 && hdr) {
      // leaf with a header: any single free slot; drain retired blocks first
      while (!leafpool.empty() && (nfree[leafpool.back()] == 0 || nbase[leafpool.back()] == 0)) leafpool.pop_back();
      if (!leafpool.empty()) {
        Mask m = cand(leafpool.back());
        int c = first_bit(m);
        if (c >= 0) {
          blk = leafpool.back();
          x = c;
        }
      }
    }
    // youngest open blocks first (most recently opened = most room); failures here are not held against a block
    auto try_youngest = [&]() {
      unsigned tried = 0;
      for (size_t i = open.size(); i-- > open_head && tried < 2 * TRIES;) {
        const uint32_t b = open[i];
        if (b == UINT32_MAX || nfree[b] < need || nbase[b] == 0) continue;
        tried++;
        const int c = first_bit(cand(b));
        if (c >= 0) {
          blk = b;
          x = c;
          return;
        }
      }
    };
    // a wide row needs whole label groups: it looks at the youngest blocks before the old, fragmented ones
    if (x < 0 && need > 16) try_youngest();
    if (x < 0) {
      unsigned tried = 0, scanned = 0;
      for (size_t i = open_head; i < open.size() && tried < TRIES && scanned < 48; i++) {
        uint32_t b = open[i];
        if (b == UINT32_MAX) {
          if (i == open_head) open_head++;
          continue;
        }
        if (nfree[b] == 0 || nbase[b] == 0) {  // exhausted: out of the scan window for good
          if (nbase[b]) basepool.push_back(b);
          open[i] = UINT32_MAX;
          continue;
        }
        scanned++;
        if (nfree[b] < need) {
          // cannot host this state; count as a failure only for small states
          if (need <= 3 && ++nfail[b] >= MAXFAIL) {
            if (nfree[b] && nbase[b]) leafpool.push_back(b);
            open[i] = UINT32_MAX;
          }
          continue;
        }
        tried++;
        Mask m = cand(b);
        int c = first_bit(m);
        if (c >= 0) {
          blk = b;
          x = c;
          break;
        }
        if (++nfail[b] >= MAXFAIL) {
          if (nfree[b] && nbase[b]) leafpool.push_back(b);
          open[i] = UINT32_MAX;
        }
      }
    }
    if (x < 0 && need <= 16) try_youngest();  // the old blocks had no room for it
    if (x < 0) {
      blk = new_block();
      open.push_back(blk);
      x = first_bit(cand(blk));  // a fresh block hosts any state (at position 0 unless a label is 0, which none is)
    }
    baseb[blk][(unsigned)x >> 6] &= ~(1ull << ((unsigned)x & 63));
    nbase[blk]--;
    if (hdr) take(blk, (unsigned)x);
    for (unsigned i = 0; i < nc; i++) take(blk, (unsigned)x ^ labels[i]);
    p.base[s] = blk * 256u + (uint32_t)x;
    (void)popcnt;
  }
  p.n_slots = (uint32_t)freeb.size() * 256u;
  if (n_regular == S) p.deep_fail_start = p.n_slots;
  for (uint32_t d = cur_depth + 1; d < kSegDepth + 2; d++) p.seg_start[d] = p.n_slots;
}

bool encode_image(const Automaton &a, const Placement &p, bool force_wide, Image &img) {
  img = Image();
  img.n_slots = p.n_slots;
  img.root_base = p.base[0];
  const uint32_t S = a.n_states;
  if (p.n_slots > W_BASE_MASK) return false;
  img.compact = !force_wide && p.n_slots <= C_MAX_SLOTS;
  if (!img.compact) {
    if (a.n_keys > (1u << 24)) return false;
    img.wide.assign(p.n_slots, 0);
    for (uint32_t s = 0; s < S; s++) {
      uint32_t b = p.base[s];
      // header: label 0, lo = fail base | W_FAILROOT when the fail state's own fail is root
      if (needs_header(a, p, s))
        img.wide[b] = (uint64_t)(p.base[a.fail[s]] | (a.fail[a.fail[s]] == 0 ? W_FAILROOT : 0u));
      for (uint32_t j = 0; j < a.n_child[s]; j++) {
        uint32_t c = a.first_child[s] + j;
        uint8_t lab = a.in_label[c];
        uint32_t lo = p.base[c];
        uint32_t hi = lab;
        if (a.key_of[c] >= 0) {
          lo |= W_END;
          hi |= (uint32_t)a.key_of[c] << 8;
        }
        if (a.fail[c] == 0) lo |= W_FAILROOT;
        img.wide[b ^ lab] = ((uint64_t)hi << 32) | lo;
      }
    }
  } else {
    img.narrow.assign(p.n_slots, 0);
    img.end_key.assign(p.n_slots, -1);
    for (uint32_t s = 0; s < S; s++) {
      uint32_t b = p.base[s];
      // header: label 0, fail base, C_FAILROOT when the fail state's own fail is root
      if (needs_header(a, p, s))
        img.narrow[b] = (p.base[a.fail[s]] << C_BASE_SHIFT) | (a.fail[a.fail[s]] == 0 ? C_FAILROOT : 0u);
      if (a.key_of[s] >= 0) img.end_key[b] = a.key_of[s];
      for (uint32_t j = 0; j < a.n_child[s]; j++) {
        uint32_t c = a.first_child[s] + j;
        uint8_t lab = a.in_label[c];
        uint32_t v = (p.base[c] << C_BASE_SHIFT) | lab;
        if (a.key_of[c] >= 0) v |= C_END;
        if (a.fail[c] == 0) v |= C_FAILROOT;
        img.narrow[b ^ lab] = v;
      }
    }
  }
  return true;
}

}  // namespace aha
