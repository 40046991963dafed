// unit.cpp -- see unit.hpp.
#include "unit.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

namespace aha {

namespace {

struct UTrans {
  uint32_t sym;    // while collecting: class << 16 | payload; afterwards the symbol
  uint32_t child;  // byte-level state id (at a unit boundary)
  uint32_t raw;    // the unit's bytes as a little-endian integer (unit.hpp, MARKS)
};

}  // namespace

static void build_unit_bits(const Automaton &a, UnitImage &u, bool force, uint32_t base_bits) {
  u = UnitImage();
  u.base_bits = base_bits;
  const uint32_t max_slots = 1u << base_bits;
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
        tr.push_back({(1u << 16) | b0, c1, b0});
        continue;
      }
      for (uint32_t j2 = 0; j2 < a.n_child[c1]; j2++) {
        const uint32_t c2 = a.first_child[c1] + j2;
        const uint32_t b1 = a.in_label[c2];
        if (need[c2] == 0) {
          const uint32_t p = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
          lo2 = std::min(lo2, p);
          hi2 = std::max(hi2, p + 1);
          tr.push_back({(2u << 16) | p, c2, b0 | b1 << 8});
          continue;
        }
        for (uint32_t j3 = 0; j3 < a.n_child[c2]; j3++) {
          const uint32_t c3 = a.first_child[c2] + j3;
          const uint32_t b2 = a.in_label[c3];
          const uint32_t p = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
          lo3 = std::min(lo3, p);
          hi3 = std::max(hi3, p + 1);
          tr.push_back({(3u << 16) | p, c3, b0 | b1 << 8 | b2 << 16});
        }
      }
    }
  }
  first[S] = (uint32_t)tr.size();
  u.n_states = (uint32_t)ustates.size();
  u.n_trans = (uint32_t)tr.size();
  if (getenv("AHA_DEBUG")) fprintf(stderr, "aha: unit image: %u states, %u transitions, %u-bit bases\n", u.n_states, u.n_trans, base_bits);
  if ((uint64_t)u.n_states + u.n_states / 8 + 1 > max_slots) {  // every state needs an identity below the array's end
    u.why = "more states than the bases address";
    return;
  }

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
  for (uint32_t s : ustates)  // ascending symbols (the byte-level children come in label order: already the case)
    if (!std::is_sorted(tr.begin() + first[s], tr.begin() + first[s + 1], [](const UTrans &x, const UTrans &y) { return x.sym < y.sym; }))
      std::sort(tr.begin() + first[s], tr.begin() + first[s + 1], [](const UTrans &x, const UTrans &y) { return x.sym < y.sym; });
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

  // ---- characters from the root, saturating (BFS order: a state comes before the states its transitions lead to)
  std::vector<uint8_t> udepth(S, 0);
  for (uint32_t s : ustates)
    for (uint32_t t = first[s]; t < first[s + 1]; t++) udepth[tr[t].child] = (uint8_t)std::min<uint32_t>(udepth[s] + 1u, 255u);
  // ---- MARKS and PAIR TABLE (unit.hpp): the two-unit paths keyed by raw bytes -- a blocked Bloom filter (the smallest of
  // 2^10 .. 2^14 words that stays under 1/16 full, else the largest) and a perfect hash table behind it
  for (uint32_t s : ustates)
    if (udepth[s] == 1 && a.key_of[s] >= 0) u.unit_key = true;
  {  // END states of three units or more along one trie path (BFS order: parents first)
    std::vector<uint8_t> de(S, 0);
    for (uint32_t s : ustates)
      for (uint32_t t = first[s]; t < first[s + 1]; t++) {
        const uint32_t c = tr[t].child;
        de[c] = (uint8_t)std::min<uint32_t>(de[s] + ((udepth[c] >= 3 && a.key_of[c] >= 0) ? 1u : 0u), 255u);
        u.deep_ends_max = std::max<uint32_t>(u.deep_ends_max, de[c]);
      }
  }
  if (!u.unit_key) {
    struct Pair {
      uint32_t raw0, raw1, child, h;
    };
    std::vector<Pair> pairs;
    for (uint32_t t0 = first[0]; t0 < first[1]; t0++) {
      const uint32_t c0 = tr[t0].child;
      for (uint32_t t1 = first[c0]; t1 < first[c0 + 1]; t1++) pairs.push_back({tr[t0].raw, tr[t1].raw, tr[t1].child, 0u});
    }
    u.n_pairs = (uint32_t)pairs.size();
    // the second unit's multiplier: no two pairs may share the 32-bit hash (they would share every slot of the table)
    static const uint32_t k1s[] = {kSkipKA, 0xC2B2AFu, 0x27D4EBu, 0x165667u, 0xD3A264u, 0xFD7047u, 0xB55A4Fu, 0x7FEB35u};
    bool placed = false;
    for (uint32_t k1 : k1s) {
      if (pairs.empty() || pairs.size() > (size_t)kPairMaxGroups * 8) break;
      for (Pair &p : pairs) p.h = sk_hash(sk_part(p.raw0), p.raw1, k1);
      std::vector<uint32_t> sorted(pairs.size());
      for (size_t i = 0; i < pairs.size(); i++) sorted[i] = pairs[i].h;
      std::sort(sorted.begin(), sorted.end());
      if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) continue;
      uint32_t G = 256;
      while (G < kPairMaxGroups && G * 4u < pairs.size()) G <<= 1;
      std::vector<std::vector<uint32_t>> groups(G);
      for (uint32_t i = 0; i < pairs.size(); i++) groups[pt_group(pairs[i].h, G)].push_back(i);
      std::vector<uint32_t> order(G);
      for (uint32_t g = 0; g < G; g++) order[g] = g;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return groups[x].size() > groups[y].size(); });
      uint32_t lg = 8;
      while (((size_t)1 << lg) * 2u < pairs.size() * 3u) lg++;  // load <= 2/3
      for (; lg <= 21 && !placed; lg++) {
        std::vector<uint32_t> tab((size_t)4 << lg, 0u);
        std::vector<uint8_t> disp(G, 0);
        bool okp = true;
        std::vector<uint32_t> slots;
        for (uint32_t gi = 0; okp && gi < G; gi++) {
          const std::vector<uint32_t> &grp = groups[order[gi]];
          if (grp.empty()) break;
          uint32_t d = 0;
          for (; d < 256; d++) {
            slots.clear();
            bool fits = true;
            for (uint32_t i : grp) {
              const uint32_t sl = pt_slot(pairs[i].h, d, lg);
              if (tab[4 * (size_t)sl] != 0 || std::find(slots.begin(), slots.end(), sl) != slots.end()) {
                fits = false;
                break;
              }
              slots.push_back(sl);
            }
            if (fits) break;
          }
          if (d == 256) {
            okp = false;
            break;
          }
          disp[order[gi]] = (uint8_t)d;
          for (size_t j = 0; j < grp.size(); j++) {
            const Pair &p = pairs[grp[j]];
            uint32_t cf = 0;
            for (uint32_t t2 = first[p.child]; t2 < first[p.child + 1]; t2++) cf |= pt_cls(tr[t2].raw);
            const int32_t key = a.key_of[p.child];
            const uint32_t c4 = key >= 0 ? std::min<uint32_t>(a.key_cnt[key], kUMaxC4) : 0u;
            uint32_t *e = &tab[4 * (size_t)slots[j]];
            e[0] = p.raw0 | c4 << 24;
            e[1] = p.raw1;
            e[2] = (uint32_t)key;
            e[3] = cf;
          }
        }
        if (okp) {
          u.pair_tab.swap(tab);
          u.pair_disp.swap(disp);
          u.pair_log2 = lg;
          u.pair_groups = G;
          u.pair_k1 = k1;
          placed = true;
        }
      }
      if (placed) break;
    }
    if (!placed) {  // (the marks alone need no table: the first multiplier)
      u.pair_k1 = kSkipKA;
      for (Pair &p : pairs) p.h = sk_hash(sk_part(p.raw0), p.raw1, u.pair_k1);
    }
    for (uint32_t lg = kSkipMinLog2; !pairs.empty() && lg <= kSkipMaxLog2; lg++) {
      u.mark_bloom.assign((size_t)1 << lg, 0u);
      for (const Pair &p : pairs) u.mark_bloom[sk_word(p.h, lg)] |= sk_mask(p.h);
      uint64_t bits = 0;
      for (uint32_t x : u.mark_bloom) bits += (uint64_t)__builtin_popcount(x);
      u.mark_log2 = lg;
      u.mark_fill_permille = (uint32_t)(bits * 1000 / ((uint64_t)32 << lg));
      if (bits * 16 <= ((uint64_t)32 << lg)) break;
    }
  }
  auto has_header = [&](uint32_t s) {  // the fail state is neither the root nor a one-character state: its word is kept in a slot
    return s != 0 && a.fail[s] != 0 && udepth[a.fail[s]] != 1;
  };

  // ---- placement (unit.hpp, IMAGE): unique bases, the root at base 0 without slots.
  // Big states -- kUBigDegree transitions or more -- get a private block of `bb` slots each behind the shared array; the
  // children on high symbols go, group by group, into runs of free slots of the shared array.
  u.n_low = std::min<uint32_t>((u.n2 + 1 + 31u) & ~31u, 512u);
  u.g0 = u.n_low - (u.n_low >> 5);
  const uint32_t n_groups = u.n_syms > u.n_low ? (u.n_syms - u.n_low + 31u) / 32u : 0u;
  uint32_t bb = 64;
  while (bb < u.n_low + n_groups) bb <<= 1;
  u.big_block = bb;
  constexpr uint32_t kBlk = 1u << 15;  // a base and base ^ symbol share a block of 2^15 slots (symbols are below 2^15)
  constexpr uint32_t kMaxBlocks = kUMaxSlots / kBlk;
  struct Bits {
    std::vector<uint64_t> w;
    explicit Bits(size_t n) : w((n + 63) / 64, 0ull) {}
    bool get(uint32_t i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
    void set(uint32_t i) { w[i >> 6] |= 1ull << (i & 63); }
  };
  Bits used(kUMaxSlots), isb(kUMaxSlots);
  std::vector<uint32_t> freec(kMaxBlocks, kBlk), curw(kMaxBlocks, 0);
  std::vector<uint32_t> idfree(kMaxBlocks, kBlk);  // identities of the block that are no state's yet (a base lies in its slots' block)
  std::vector<uint32_t> base(S, 0);
  used.set(0);  // index 0 stays empty: base 0 is the root
  isb.set(0);
  freec[0]--;
  idfree[0]--;
  uint32_t n_open = 1, n_big = 0, n_bases = 1;
  std::vector<uint32_t> order, bigs;
  for (uint32_t s : ustates) {
    if (s == 0) continue;
    const uint32_t deg = first[s + 1] - first[s];
    if (deg >= kUBigDegree) {
      bigs.push_back(s);
    } else if (deg + (has_header(s) ? 1u : 0u) > 0) {
      order.push_back(s);
    }
  }
  n_big = (uint32_t)bigs.size();
  auto nsym = [&](uint32_t s) { return first[s + 1] - first[s] + (has_header(s) ? 1u : 0u); };
  // one-character states first (their transitions are what random text probes: they end up in the low blocks), then
  // the others; wide states before narrow ones inside each class, so that they meet empty blocks and the single
  // transitions of the deep states fill the holes
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
    const uint32_t cx = udepth[x] == 1 ? 0u : 1u, cy = udepth[y] == 1 ? 0u : 1u;
    if (cx != cy) return cx < cy;
    return nsym(x) > nsym(y);
  });
  auto open_block = [&]() -> bool {
    if ((uint64_t)(n_open + 1) * kBlk + (uint64_t)n_big * bb > max_slots) {
      if (getenv("AHA_DEBUG"))
        fprintf(stderr, "aha: unit image: %u blocks open, %u big states of %u slots, %u bases so far: no room for another block\n",
                n_open, n_big, bb, n_bases);
      return false;
    }
    n_open++;
    return true;
  };
  // a base for the state with symbols sy[0..k) (0 = its header) in block blk, or 0
  uint32_t sy[kUBigDegree + 1];
  auto try_block = [&](uint32_t blk, uint32_t k) -> uint32_t {
    const uint32_t fr = freec[blk];
    if (fr < k) return 0;
    uint64_t budget = fr;
    // expected number of free slots whose identity is still free (with millions of single transitions the blocks run
    // out of identities before they run out of slots, and every later state would search them to the end)
    if ((double)fr * idfree[blk] / kBlk < 0.5) return 0;
    if (k > 1) {  // expected share of the candidates whose other k - 1 slots are free as well
      double p = (double)idfree[blk] / kBlk;
      const double fl = (double)fr / kBlk;
      for (uint32_t i = 1; i < k; i++) p *= fl;
      if (p * fr < 0.5) return 0;
      // (+ 1024: the symbols of wide states cluster -- the letters --, so neighbouring candidates fail together)
      budget = std::min<uint64_t>(fr, (uint64_t)(4.0 / p) + 1024);
    }
    const uint32_t w0 = blk * (kBlk / 64), nw = kBlk / 64;
    uint32_t wi = curw[blk];
    for (uint32_t step = 0; step < nw && budget; step++, wi = wi + 1 < nw ? wi + 1 : 0) {
      uint64_t fw = ~used.w[w0 + wi];
      while (fw && budget) {
        const uint32_t f = ((w0 + wi) << 6) + (uint32_t)__builtin_ctzll(fw);
        fw &= fw - 1;
        budget--;
        const uint32_t cand = f ^ sy[0];
        if (cand == 0 || isb.get(cand)) continue;
        bool okc = true;
        for (uint32_t i = 1; okc && i < k; i++) okc = !used.get(cand ^ sy[i]);
        if (okc) {
          curw[blk] = wi;
          return cand;
        }
      }
    }
    return 0;
  };
  uint32_t lowest = 0;  // lowest block that may have a free slot
  // dead[blk]: a state of that many slots found no place there; blocks only fill up, so wider ones need not look (with a
  // million keys the one-character states alone are 20 000 states of ~30 transitions: without this every one of them
  // walks through every block that is a quarter full -- minutes)
  std::vector<uint32_t> dead(kMaxBlocks, ~0u);
  auto place = [&](uint32_t k) -> uint32_t {
    while (lowest < n_open && freec[lowest] == 0) lowest++;
    for (uint32_t blk = lowest;; blk++) {
      if (blk == n_open && !open_block()) return 0;
      if (k >= dead[blk]) continue;
      const uint32_t b = try_block(blk, k);
      if (!b && (k > 1 || (double)freec[blk] * idfree[blk] / kBlk < 8.0)) dead[blk] = k;
      if (b) {
        for (uint32_t i = 0; i < k; i++) used.set(b ^ sy[i]);
        freec[blk] -= k;
        isb.set(b);
        idfree[blk]--;
        n_bases++;
        return b;
      }
    }
  };
  // runs of free slots for the big states' children on high symbols: {first slot} per (big state, group); a child's
  // slot t is probed as "state t ^ symbol" (unit.hpp), so that identity must not be a real state's, now or later
  std::vector<std::vector<uint32_t>> big_runs(n_big);
  auto place_runs = [&]() -> bool {
    uint32_t t = kBlk;  // not in block 0: t ^ symbol != 0
    for (uint32_t bi = 0; bi < n_big; bi++) {
      const uint32_t s = bigs[bi];
      big_runs[bi].assign(n_groups, 0u);
      for (uint32_t q = first[s]; q < first[s + 1];) {
        if (tr[q].sym < u.n_low) {
          q++;
          continue;
        }
        const uint32_t g = (tr[q].sym - u.n_low) >> 5;
        uint32_t q2 = q;
        while (q2 < first[s + 1] && ((tr[q2].sym - u.n_low) >> 5) == g) q2++;
        const uint32_t cnt = q2 - q;
        for (;; t++) {
          if (t + cnt > n_open * kBlk) {
            if (!open_block()) return false;
          }
          if ((t & (kBlk - 1)) + cnt > kBlk) {  // a run stays inside one block
            t = (t | (kBlk - 1));
            continue;
          }
          bool okr = true;
          for (uint32_t j = 0; okr && j < cnt; j++) okr = !used.get(t + j) && !isb.get((t + j) ^ tr[q + j].sym);
          if (okr) break;
        }
        for (uint32_t j = 0; j < cnt; j++) {
          used.set(t + j);
          if (!isb.get((t + j) ^ tr[q + j].sym)) {
            isb.set((t + j) ^ tr[q + j].sym);
            idfree[t / kBlk]--;
            n_bases++;
          }
        }
        freec[t / kBlk] -= cnt;
        big_runs[bi][g] = t;
        t += cnt;
        q = q2;
      }
    }
    return true;
  };
  bool runs_done = n_big == 0 || n_groups == 0;
  const auto t_place = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (getenv("AHA_DEBUG"))
      fprintf(stderr, "aha: unit image: %s at %.2f s, %u blocks\n", what,
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t_place).count(), n_open);
  };
  for (size_t oi = 0; oi < order.size(); oi++) {
    const uint32_t s = order[oi];
    if (!runs_done && udepth[s] != 1) {  // behind the one-character states
      if (!place_runs()) {
        u.why = "more transitions than the bases address";
        return;
      }
      runs_done = true;
    }
    uint32_t k = 0;
    for (uint32_t t = first[s]; t < first[s + 1]; t++) sy[k++] = tr[t].sym;
    if (has_header(s)) sy[k++] = 0u;
    const uint32_t b = place(k);
    if (!b) {
      u.why = "more transitions than the bases address";
      return;
    }
    base[s] = b;
  }
  if (!runs_done && !place_runs()) {
    u.why = "more transitions than the bases address";
    return;
  }
  lap("states with slots placed");
  // states that own no slot: any unused identity (fewer states than slots: blocks are opened for them if need be)
  {
    uint32_t n_ids = 0;
    for (uint32_t s : ustates)
      if (s != 0 && first[s + 1] == first[s] && !has_header(s)) n_ids++;
    while ((uint64_t)n_open * kBlk < (uint64_t)n_bases + n_ids + 1)
      if (!open_block()) {
        u.why = "more states than the bases address";
        return;
      }
    uint32_t idc = 1;
    for (uint32_t s : ustates) {
      if (s == 0 || first[s + 1] != first[s] || has_header(s)) continue;
      while (isb.get(idc)) idc++;
      base[s] = idc;
      isb.set(idc);
    }
  }
  const uint32_t n_shared = n_open * kBlk;
  const uint32_t n_slots = ((n_shared + n_big * bb) + kBlk - 1) & ~(kBlk - 1);
  for (uint32_t bi = 0; bi < n_big; bi++) base[bigs[bi]] = n_shared + bi * bb;

  // ---- the image
  auto word = [&](uint32_t st) -> uint32_t {  // the state as one word (unit.hpp)
    uint32_t flt = 0;
    for (uint32_t q = first[st]; q < first[st + 1]; q++) flt |= 1u << (tr[q].sym & 7u);
    if (first[st + 1] - first[st] >= kUBigDegree) flt = 0xFFu;  // (a big state always probes: its group records answer)
    const bool nfr = a.fail[st] != 0, f1 = nfr && udepth[a.fail[st]] == 1;
    // (the filter's classes beyond its stored bits -- 7, and 6 with 23-bit bases -- always probe)
    return base[st] | ((flt << base_bits) & u_all_filter(base_bits)) | (f1 ? (1u << 29) : 0u) | (nfr ? (1u << 30) : 0u) |
           (a.key_of[st] >= 0 ? 0x80000000u : 0u);
  };
  auto u_c4_of = [](const Automaton &au, uint32_t st) -> uint32_t {  // hits an event in this state stands for, capped
    return au.key_of[st] >= 0 ? std::min<uint32_t>(au.key_cnt[au.key_of[st]], kUMaxC4) : 0u;
  };
  u.n_slots = n_slots;
  u.n_shared = n_shared;
  u.n_big = n_big;
  u.slots.assign(n_slots, 0ull);
  u.end_key.assign(n_slots, -1);
  u.root.assign(u.n_syms, 0u);
  std::vector<uint32_t> big_index(n_big ? S : 0, 0u);
  for (uint32_t bi = 0; bi < n_big; bi++) big_index[bigs[bi]] = bi;
  for (uint32_t s : ustates) {
    const uint32_t b = base[s];
    const bool big = s != 0 && first[s + 1] - first[s] >= kUBigDegree;
    if (has_header(s)) {
      u.slots[b] = (uint64_t)(word(a.fail[s]) & 0x7FFFFFFFu);  // symbol 0 (falling into a state reports nothing: END is not carried)
      u.n_nfr++;
    }
    if (a.key_of[s] >= 0) u.end_key[b] = a.key_of[s];
    for (uint32_t t = first[s]; t < first[s + 1]; t++) {
      const uint32_t c = tr[t].child, sym = tr[t].sym;
      const uint64_t entry = ((uint64_t)(sym | u_c4_of(a, c) << 16) << 32) | word(c);
      if (s == 0) {
        u.root[sym] = word(c);
      } else if (!big || sym < u.n_low) {
        u.slots[b ^ sym] = entry;
      } else {  // group record {symbols present, slot of the group's first child} + the child in the group's run
        const uint32_t g = (sym - u.n_low) >> 5, run = big_runs[big_index[s]][g];
        uint64_t &rec = u.slots[b + u.n_low + g];
        const uint32_t bits = (uint32_t)rec, rank = (uint32_t)__builtin_popcount(bits);  // (symbols come in ascending order)
        rec = ((uint64_t)run << 32) | (bits | 1u << ((sym - u.n_low) & 31u));
        u.slots[run + rank] = entry;
      }
    }
  }
  u.ok = true;
}

void build_unit(const Automaton &a, UnitImage &u, bool force) {
  // AHA_UNIT_BASE_BITS=23 (tests, the fuzzer): the wide format for every image, so that its kernels meet small random automata
  const char *bw = getenv("AHA_UNIT_BASE_BITS");
  if (bw && strcmp(bw, "23") == 0) {
    build_unit_bits(a, u, force, 23u);
    return;
  }
  build_unit_bits(a, u, force, 22u);
  // (only the size of the image asks for wider bases: every other refusal holds for them too)
  if (!u.ok && strstr(u.why, "bases address")) build_unit_bits(a, u, force, 23u);
}

}  // namespace aha
