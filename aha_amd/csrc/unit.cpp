// unit.cpp -- see unit.hpp.
#include "unit.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace aha {

namespace {

struct UTrans {
  uint32_t sym;    // while collecting: class << 16 | payload; afterwards the symbol
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
    all++;
    if (a.in_label[s] >= 0x80u) multi++;
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
  uint32_t lo2 = ~0u, hi2 = 0, lo3 = ~0u, hi3 = 0;  // payload ranges the keys use
  for (uint32_t s = 0; s < S; s++) {
    first[s] = (uint32_t)tr.size();
    if (need[s]) continue;
    ustates.push_back(s);
    for (uint32_t j = 0; j < a.n_child[s]; j++) {
      const uint32_t c1 = a.first_child[s] + j;
      const uint32_t b0 = a.in_label[c1];
      if (need[c1] == 0) {
        tr.push_back({(1u << 16) | b0, c1});
        continue;
      }
      for (uint32_t j2 = 0; j2 < a.n_child[c1]; j2++) {
        const uint32_t c2 = a.first_child[c1] + j2;
        const uint32_t b1 = a.in_label[c2];
        if (need[c2] == 0) {
          const uint32_t p = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
          lo2 = std::min(lo2, p);
          hi2 = std::max(hi2, p + 1);
          tr.push_back({(2u << 16) | p, c2});
          continue;
        }
        for (uint32_t j3 = 0; j3 < a.n_child[c2]; j3++) {
          const uint32_t c3 = a.first_child[c2] + j3;
          const uint32_t b2 = a.in_label[c3];
          const uint32_t p = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
          lo3 = std::min(lo3, p);
          hi3 = std::max(hi3, p + 1);
          tr.push_back({(3u << 16) | p, c3});
        }
      }
    }
  }
  first[S] = (uint32_t)tr.size();
  u.n_states = (uint32_t)ustates.size();
  u.n_trans = (uint32_t)tr.size();

  // ---- the dense alphabet (unit.hpp, SYMBOLS)
  u.c2lo = hi2 ? lo2 : 0;
  u.w2 = hi2 ? hi2 - lo2 : 0;
  u.c3lo = hi3 ? lo3 : 0;
  u.w3 = hi3 ? hi3 - lo3 : 0;
  u.n1 = 128;
  u.n2 = 129 + u.w2;
  u.n_syms = u.n2 + 1 + u.w3;
  if (u.n_syms > kUMaxSyms) {
    u.why = "the keys' characters span more symbols than the root table holds in LDS";
    return;
  }
  for (UTrans &t : tr) {
    const uint32_t cls = t.sym >> 16, p = t.sym & 0xFFFFu;
    t.sym = cls == 1 ? p : (cls == 2 ? 129 + (p - u.c2lo) : u.n2 + 1 + (p - u.c3lo));
  }
  u.tables.assign(kUTabWords, 0u);
  for (uint32_t b = 0; b < 256; b++) {
    uint32_t a1 = 0, a2 = 0, L = 1, lo = 1, span = 127;
    int64_t base = 0;  // a bad byte: 0 is below the class's first symbol, so it decodes to "other" = 0
    if (b >= 1 && b < 0x80) {
      base = b;
    } else if ((b & 0xE0u) == 0xC0u) {
      L = 2;
      a1 = (kUA1 + 256) * 4;
      lo = 129;
      span = u.w2;
      base = 129 + (int64_t)((b & 0x1Fu) << 6) - u.c2lo;
    } else if ((b & 0xF0u) == 0xE0u) {
      L = 3;
      a1 = (kUA1 + 512) * 4;
      a2 = (kUA2 + 256) * 4;
      lo = u.n2 + 1;
      span = u.w3;
      base = (int64_t)u.n2 + 1 + (int64_t)((b & 0x0Fu) << 12) - u.c3lo;
    }
    if (L == 1) {
      a1 = kUA1 * 4;  // the all-zero parts of A1 / A2
      a2 = kUA2 * 4;
    } else if (L == 2) {
      a2 = kUA2 * 4;
    }
    u.tables[kUT0a + 4 * b + 0] = a1;
    u.tables[kUT0a + 4 * b + 1] = a2;
    u.tables[kUT0a + 4 * b + 2] = (uint32_t)(base + kUBias);
    u.tables[kUT0a + 4 * b + 3] = lo + kUBias;
    u.tables[kUT0b + 2 * b + 0] = span;
    u.tables[kUT0b + 2 * b + 1] = L;
    const bool cont = (b & 0xC0u) == 0x80u;
    u.tables[kUA1 + 256 + b] = cont ? (b & 0x3Fu) : kUPoison;
    u.tables[kUA1 + 512 + b] = cont ? ((b & 0x3Fu) << 6) : kUPoison;
    u.tables[kUA2 + 256 + b] = cont ? (b & 0x3Fu) : kUPoison;
  }

  // ---- placement: unique bases, the root at base 0 without slots; states with kUBigDegree transitions or more get a
  // region of 2^15 slots each behind the shared array (unit.hpp)
  uint64_t want = 0;
  uint32_t n_big = 0;
  for (uint32_t s : ustates) {
    if (s == 0) continue;
    const uint32_t deg = first[s + 1] - first[s];
    if (deg >= kUBigDegree)
      n_big++;
    else
      want += deg;
  }
  const uint32_t n_shared = (uint32_t)(((want * 4 / 3 + 4096) + (1u << 16) - 1) >> 16) << 16;  // load <= 3/4
  const uint64_t n_total = (((uint64_t)n_shared + (uint64_t)n_big * 32768u) + 65535u) & ~65535ull;
  if (n_total > kUMaxSlots || u.n_states >= n_shared) {
    u.why = "more transitions than the 22-bit bases address";
    return;
  }
  const uint32_t n_slots = (uint32_t)n_total;
  std::vector<uint8_t> used(n_slots, 0), is_base(n_slots, 0);
  std::vector<uint32_t> base(S, 0);
  used[0] = 1;  // index 0 stays empty: base 0 is the root
  is_base[0] = 1;
  uint32_t cursor = 1;  // lowest slot that may be free
  uint32_t idc = 1;     // lowest identity that may be unused
  uint32_t big_next = 0;
  for (uint32_t s : ustates) {
    if (s == 0) continue;
    const uint32_t lo = first[s], hi = first[s + 1];
    uint32_t b = 0;
    if (hi - lo >= kUBigDegree) {  // a region of its own: base + symbol
      b = n_shared + big_next * 32768u;
      big_next++;
    } else if (lo == hi) {  // owns no slot: any unused identity
      while (is_base[idc]) idc++;  // (fewer states than slots: never runs off the end)
      b = idc;
    } else {
      // candidates: put the first symbol on the free slots in turn
      const uint32_t c0 = tr[lo].sym;
      while (used[cursor]) cursor = cursor + 1 < n_shared ? cursor + 1 : 1;
      uint32_t f = cursor;
      for (uint32_t tries = 0;; tries++, f = f + 1 < n_shared ? f + 1 : 1) {
        if (tries > n_shared) {
          if (getenv("AHA_DEBUG")) {
            fprintf(stderr, "unit placement: state %u depth %u with %u children, cursor %u of %u\n", s, a.depth[s], hi - lo,
                    cursor, n_slots);
            for (uint32_t blk = 0; blk < n_slots >> 16; blk++) {
              uint32_t nu = 0, nbs = 0;
              for (uint32_t i = 0; i < 65536; i++) nu += used[(blk << 16) + i], nbs += is_base[(blk << 16) + i];
              fprintf(stderr, "  block %u: %u used, %u bases\n", blk, nu, nbs);
            }
            uint32_t mx = 0;
            for (uint32_t t = lo; t < hi; t++) mx = std::max(mx, tr[t].sym);
            fprintf(stderr, "  first sym %u max sym %u\n", tr[lo].sym, mx);
          }
          u.why = "placement failed";
          return;
        }
        if (used[f]) continue;
        const uint32_t cand = f ^ c0;  // same block of 2^16 slots as f
        if (cand == 0 || is_base[cand]) continue;
        bool okc = true;
        for (uint32_t t = lo; okc && t < hi; t++) okc = !used[cand ^ tr[t].sym];
        if (okc) {
          b = cand;
          break;
        }
      }
    }
    base[s] = b;
    is_base[b] = 1;
    for (uint32_t t = lo; t < hi; t++) used[b ^ tr[t].sym] = 1;
  }

  // ---- the image
  std::vector<uint8_t> udepth(S, 0);  // characters from the root, saturating
  for (uint32_t s : ustates)           // BFS order: a state comes before the states its transitions lead to
    for (uint32_t t = first[s]; t < first[s + 1]; t++) udepth[tr[t].child] = (uint8_t)std::min<uint32_t>(udepth[s] + 1u, 255u);
  auto word = [&](uint32_t st) -> uint32_t {  // the state as one word (unit.hpp)
    uint32_t flt = 0;
    for (uint32_t q = first[st]; q < first[st + 1]; q++) flt |= 1u << (tr[q].sym & 7u);
    const bool nfr = a.fail[st] != 0, f1 = nfr && udepth[a.fail[st]] == 1;
    return base[st] | ((flt & 0x7Fu) << 22) | (f1 ? (1u << 29) : 0u) | (nfr ? (1u << 30) : 0u) |
           (a.key_of[st] >= 0 ? 0x80000000u : 0u);
  };
  auto u_c4_of = [](const Automaton &au, uint32_t st) -> uint32_t {  // hits an event in this state stands for, at most 15
    return au.key_of[st] >= 0 ? std::min<uint32_t>(au.key_cnt[au.key_of[st]], 15u) : 0u;
  };
  u.n_slots = n_slots;
  u.n_shared = n_shared;
  u.slots.assign(n_slots, 0ull);
  u.fail_tab.assign(n_slots, 0u);
  u.end_key.assign(n_slots, -1);
  u.root.assign(u.n_syms, 0u);
  for (uint32_t s : ustates) {
    const uint32_t b = base[s];
    if (s != 0 && a.fail[s] != 0 && udepth[a.fail[s]] != 1) {
      u.fail_tab[b] = word(a.fail[s]) & 0x7FFFFFFFu;  // (falling into a state reports nothing: END is not carried)
      u.n_nfr++;
    }
    if (a.key_of[s] >= 0) u.end_key[b] = a.key_of[s];
    for (uint32_t t = first[s]; t < first[s + 1]; t++) {
      const uint32_t c = tr[t].child;
      if (s == 0)
        u.root[tr[t].sym] = word(c);
      else
        u.slots[b ^ tr[t].sym] = ((uint64_t)(tr[t].sym | u_c4_of(a, c) << 16) << 32) | word(c);
    }
  }
  u.ok = true;
}

}  // namespace aha
