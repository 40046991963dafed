// unit.cpp -- see unit.hpp.
#include "unit.hpp"

#include <algorithm>
#include <cstring>

namespace aha {

namespace {

struct UTrans {
  uint32_t code;
  uint32_t child;  // byte-level state id (at a unit boundary)
};

}  // namespace

void build_unit(const Automaton &a, UnitImage &u, bool force) {
  u = UnitImage();
  const uint32_t S = a.n_states;
  if (a.n_keys == 0 || S < 2) {
    u.why = "no keys";
    return;
  }
  if (a.n_keys > (1u << 24)) {
    u.why = "more than 2^24 keys";
    return;
  }
  // ---- which byte-level states lie at unit boundaries; is every key a sequence of good units?
  std::vector<uint8_t> need(S, 0);  // continuation bytes the unit entered on the way to s still expects
  for (uint32_t s = 0; s < S; s++) {  // BFS numbering: a parent comes before its children
    for (uint32_t j = 0; j < a.n_child[s]; j++) {
      const uint32_t c = a.first_child[s] + j;
      const uint32_t b = a.in_label[c];
      if (need[s]) {
        if ((b & 0xC0u) != 0x80u) {
          u.why = "a key holds a lead byte without its continuation bytes";
          return;
        }
        need[c] = (uint8_t)(need[s] - 1);
      } else if (b < 0x80u) {
        need[c] = 0;
      } else if ((b & 0xE0u) == 0xC0u) {
        need[c] = 1;
      } else if ((b & 0xF0u) == 0xE0u) {
        need[c] = 2;
      } else {
        u.why = "a key holds a stray continuation byte or a byte >= 0xF0";
        return;
      }
    }
    if (a.key_of[s] >= 0 && need[s]) {
      u.why = "a key ends inside a unit";
      return;
    }
  }
  for (uint32_t s = 1; s < S; s++)
    if (need[s] == 0 && need[a.fail[s]] != 0) {  // cannot happen for an eligible key set (unit.hpp)
      u.why = "a fail link leaves the unit boundaries";
      return;
    }
  uint64_t multi = 0, all = 0;
  for (uint32_t s = 1; s < S; s++) {
    const uint32_t b = a.in_label[s];
    all++;
    if (b >= 0x80u) multi++;
  }
  u.multi_permille = (uint32_t)(multi * 1000 / std::max<uint64_t>(all, 1));
  if (!force && u.multi_permille < 300) {  // mostly one-byte units: one step per unit is one step per byte
    u.why = "fewer than 30 % of the key bytes lie in multi-byte units";
    return;
  }

  // ---- unit transitions of every boundary state (byte-level walks of one, two or three edges)
  std::vector<uint32_t> first(S + 1, 0);
  std::vector<UTrans> tr;
  tr.reserve(S);
  std::vector<uint32_t> ustates;  // boundary states in BFS order
  for (uint32_t s = 0; s < S; s++) {
    first[s] = (uint32_t)tr.size();
    if (need[s]) continue;
    ustates.push_back(s);
    for (uint32_t j = 0; j < a.n_child[s]; j++) {
      const uint32_t c1 = a.first_child[s] + j;
      const uint32_t b0 = a.in_label[c1];
      if (need[c1] == 0) {
        tr.push_back({b0, c1});
        continue;
      }
      for (uint32_t j2 = 0; j2 < a.n_child[c1]; j2++) {
        const uint32_t c2 = a.first_child[c1] + j2;
        const uint32_t b1 = a.in_label[c2];
        if (need[c2] == 0) {
          tr.push_back({kUCode2 + (((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu)), c2});
          continue;
        }
        for (uint32_t j3 = 0; j3 < a.n_child[c2]; j3++) {
          const uint32_t c3 = a.first_child[c2] + j3;
          const uint32_t b2 = a.in_label[c3];
          tr.push_back({kUCode3 + (((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu)), c3});
        }
      }
    }
  }
  first[S] = (uint32_t)tr.size();
  u.n_states = (uint32_t)ustates.size();
  u.n_trans = (uint32_t)tr.size();

  // ---- placement: unique bases, the root at base 0 without slots; a header only for a fail target that does not
  // itself fail to the root
  std::vector<uint8_t> hdr(S, 0);
  for (uint32_t s : ustates)
    if (s != 0 && a.fail[s] != 0 && a.fail[a.fail[s]] != 0) hdr[a.fail[s]] = 1;
  uint64_t want = 0;
  for (uint32_t s : ustates)
    if (s != 0) want += (first[s + 1] - first[s]) + hdr[s];
  uint32_t n_slots = (uint32_t)(((want * 4 / 3 + 4096) + (1u << 17) - 1) >> 17) << 17;  // load <= 3/4
  if (n_slots > kUMaxSlots || u.n_states >= kUMaxSlots / 2) {
    u.why = "more transitions than the 21-bit bases address";
    return;
  }
  std::vector<uint8_t> used(n_slots, 0), is_base(n_slots, 0);
  std::vector<uint32_t> base(S, 0);
  used[0] = 1;  // index 0 stays empty: base 0 is the root
  is_base[0] = 1;
  uint32_t cursor = 1;  // lowest slot that may be free
  uint32_t idc = 1;     // lowest identity that may be unused
  for (uint32_t s : ustates) {
    if (s == 0) continue;
    const uint32_t lo = first[s], hi = first[s + 1];
    uint32_t b = 0;
    if (lo == hi && !hdr[s]) {  // owns no slot: any unused identity
      while (is_base[idc]) idc++;  // (fewer states than slots: never runs off the end)
      b = idc;
    } else {
      // candidates: put the first code (or, without children, the header) on the free slots in turn
      const uint32_t c0 = lo < hi ? tr[lo].code : 0u;
      while (used[cursor]) cursor = cursor + 1 < n_slots ? cursor + 1 : 1;
      uint32_t f = cursor;
      for (uint32_t tries = 0;; tries++, f = f + 1 < n_slots ? f + 1 : 1) {
        if (tries > n_slots) {
          u.why = "placement failed";
          return;
        }
        if (used[f]) continue;
        const uint32_t cand = f ^ c0;  // same block of 2^17 slots as f
        if (cand == 0 || is_base[cand]) continue;
        bool okc = !hdr[s] || !used[cand];
        for (uint32_t t = lo; okc && t < hi; t++) okc = !used[cand ^ tr[t].code];
        if (okc) {
          b = cand;
          break;
        }
      }
    }
    base[s] = b;
    is_base[b] = 1;
    if (hdr[s]) used[b] = 1;
    for (uint32_t t = lo; t < hi; t++) used[b ^ tr[t].code] = 1;
  }

  // ---- the image
  auto entry = [&](uint32_t code, uint32_t child_base, bool end, uint32_t st) -> uint64_t {
    // st: the state whose fail link the entry carries
    const uint32_t fb = base[a.fail[st]];
    const bool ffr = a.fail[a.fail[st]] == 0;
    const uint32_t lo = child_base | ((fb & 0x3FFu) << 21) | (end ? 0x80000000u : 0u);
    const uint32_t hi = code | ((fb >> 10) << 17) | (ffr ? (1u << 28) : 0u);
    return ((uint64_t)hi << 32) | lo;
  };
  u.n_slots = n_slots;
  u.slots.assign(n_slots, 0ull);
  u.end_info.assign(n_slots, 0xFFFFFFFFu);
  u.root.assign(kUCodes, 0u);
  uint32_t cnt3[16] = {0};
  for (uint32_t s : ustates) {
    const uint32_t b = base[s];
    if (hdr[s]) {
      u.slots[b] = entry(0, 0, false, s);
      u.n_headers++;
    }
    if (a.key_of[s] >= 0) {
      const uint32_t k = (uint32_t)a.key_of[s];
      u.end_info[b] = k | (std::min<uint32_t>(a.key_cnt[k], 255u) << 24);
    }
    for (uint32_t t = first[s]; t < first[s + 1]; t++) {
      const uint32_t c = tr[t].child;
      if (s == 0) {
        uint32_t flt = 0;
        for (uint32_t q = first[c]; q < first[c + 1]; q++) flt |= 1u << u_fbit(tr[q].code);
        u.root[tr[t].code] = base[c] | (flt << 21) | (a.key_of[c] >= 0 ? 0x80000000u : 0u);
        if (tr[t].code >= kUCode3) cnt3[(tr[t].code - kUCode3) >> 12]++;
      } else {
        u.slots[b ^ tr[t].code] = entry(tr[t].code, base[c], a.key_of[c] >= 0, c);
      }
    }
  }
  // the window of first bytes (0xE0 + k) whose three-byte root transitions are worth LDS: the shortest run of at most
  // 7 that holds 98 % of them, else the best run of 7
  uint32_t total3 = 0;
  for (uint32_t k = 0; k < 16; k++) total3 += cnt3[k];
  for (uint32_t n = 1; n <= 7 && total3; n++) {
    uint32_t best = 0, best_lo = 0;
    for (uint32_t lo = 0; lo + n <= 16; lo++) {
      uint32_t c = 0;
      for (uint32_t k = 0; k < n; k++) c += cnt3[lo + k];
      if (c > best) {
        best = c;
        best_lo = lo;
      }
    }
    u.lo3 = best_lo;
    u.n3 = n;
    if ((uint64_t)best * 100 >= (uint64_t)total3 * 98) break;
  }
  u.ok = true;
}

}  // namespace aha
