// hash.cpp -- see hash.hpp.
#include "hash.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace aha {

namespace {

uint32_t key_hash(const UnitImage::HTrans &t, uint32_t k1) {
  const uint32_t part = (t.parent & kHTag) ? h_part_char(t.parent & 0xFFFFFFu) : h_part_base(t.parent);
  return h_key(part, t.ch, k1);
}

HEntry entry_of(const UnitImage::HTrans &t) { return HEntry{t.parent, t.ch | t.c4 << 24, t.word, t.cf}; }

// perfect hash of the pairs: groups in descending size, each displaced by the first byte that puts all its keys on free slots
bool place_pairs(const std::vector<UnitImage::HTrans> &pairs, const std::vector<uint32_t> &hs, HashImage &h) {
  const uint32_t n = (uint32_t)pairs.size();
  uint32_t G = 256;
  while (G < kHMaxGroups && G * 4u < n) G <<= 1;
  h.n_groups = G;
  std::vector<std::vector<uint32_t>> groups(G);
  for (uint32_t i = 0; i < n; i++) groups[h_group(hs[i], G)].push_back(i);
  std::vector<uint32_t> order(G);
  for (uint32_t g = 0; g < G; g++) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return groups[x].size() > groups[y].size(); });
  uint32_t lg = 8;
  while ((1u << lg) * 2u < n * 3u) lg++;  // load <= 2/3
  for (; lg <= 22; lg++) {
    h.pair_log2 = lg;
    h.pairs.assign((size_t)1 << lg, HEntry{0, 0, 0, 0});
    h.disp.assign(G, 0);
    bool ok = true;
    std::vector<uint32_t> slots;
    for (uint32_t gi = 0; ok && gi < G; gi++) {
      const std::vector<uint32_t> &grp = groups[order[gi]];
      if (grp.empty()) break;
      uint32_t d = 0;
      for (; d < 256; d++) {
        slots.clear();
        bool fits = true;
        for (uint32_t i : grp) {
          const uint32_t sl = h_pair_slot(hs[i], d, lg);
          if (h.pairs[sl].parent != 0 || std::find(slots.begin(), slots.end(), sl) != slots.end()) {
            fits = false;
            break;
          }
          slots.push_back(sl);
        }
        if (fits) break;
      }
      if (d == 256) {
        ok = false;
        break;
      }
      h.disp[order[gi]] = (uint8_t)d;
      for (size_t j = 0; j < grp.size(); j++) h.pairs[slots[j]] = entry_of(pairs[grp[j]]);
    }
    if (ok) return true;
  }
  return false;
}

bool place_deep(const std::vector<UnitImage::HTrans> &deep, const std::vector<uint32_t> &hs, HashImage &h) {
  const uint32_t n = (uint32_t)deep.size();
  uint32_t lg = 8;
  while ((1u << lg) * 2u < n * 5u) lg++;  // load <= 0.4
  for (; lg <= 24; lg++) {
    h.deep_log2 = lg;
    h.deep.assign((size_t)1 << lg, HEntry{0, 0, 0, 0});
    std::vector<uint32_t> who((size_t)1 << lg, ~0u);
    bool ok = true;
    for (uint32_t i = 0; ok && i < n; i++) {
      uint32_t cur = i, sl = h_deep_slot1(hs[cur], lg);
      for (uint32_t kicks = 0;; kicks++) {
        if (who[sl] == ~0u) {
          who[sl] = cur;
          break;
        }
        if (kicks == 500) {
          ok = false;
          break;
        }
        std::swap(cur, who[sl]);  // evict: the evicted key moves to its other slot
        const uint32_t s1 = h_deep_slot1(hs[cur], lg), s2 = h_deep_slot2(hs[cur], lg);
        sl = sl == s1 ? s2 : s1;
      }
    }
    if (!ok) continue;
    for (size_t s = 0; s < who.size(); s++)
      if (who[s] != ~0u) h.deep[s] = entry_of(deep[who[s]]);
    return true;
  }
  return false;
}

}  // namespace

void build_hash(const UnitImage &u, HashImage &h) {
  h = HashImage();
  if (!u.ok) {
    h.why = "no unit image";
    return;
  }
  if (u.base_bits != 22) {
    h.why = "23-bit bases";
    return;
  }
  if (u.has_len1_key) {
    h.why = "a key of one character";
    return;
  }
  std::vector<UnitImage::HTrans> pairs, deep;
  for (const UnitImage::HTrans &t : u.htrans) ((t.parent & kHTag) ? pairs : deep).push_back(t);
  h.n_pairs = (uint32_t)pairs.size();
  h.n_deep = (uint32_t)deep.size();
  if (pairs.empty()) {
    h.why = "no two-character path";
    return;
  }
  if (pairs.size() > (size_t)kHMaxGroups * 8) {
    h.why = "more two-character paths than the displacement table in LDS serves";
    return;
  }
  // the character's multiplier: no two pairs may share the 32-bit hash (they would share every slot)
  static const uint32_t k1s[] = {0x9E3779u, 0xC2B2AFu, 0x27D4EBu, 0x165667u, 0xD3A264u, 0xFD7047u, 0xB55A4Fu, 0x7FEB35u};
  std::vector<uint32_t> hp(pairs.size()), hd(deep.size());
  for (uint32_t k1 : k1s) {
    h.k1 = k1;
    for (size_t i = 0; i < pairs.size(); i++) hp[i] = key_hash(pairs[i], k1);
    std::vector<uint32_t> sorted = hp;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) continue;
    if (!place_pairs(pairs, hp, h)) continue;
    for (size_t i = 0; i < deep.size(); i++) hd[i] = key_hash(deep[i], k1);
    if (!place_deep(deep, hd, h)) continue;
    h.bloom.assign((size_t)1 << kHBloomLog2, 0u);
    for (uint32_t x : hp) h.bloom[h_bloom_word(x)] |= h_bloom_mask(x);
    uint64_t bits = 0;
    for (uint32_t w : h.bloom) bits += (uint64_t)__builtin_popcount(w);
    h.bloom_fill_permille = (uint32_t)(bits * 1000 / ((uint64_t)32 << kHBloomLog2));
    if (h.bloom_fill_permille > 500) {
      h.why = "the pair filter would be more than half full";
      return;
    }
    h.ok = true;
    if (getenv("AHA_DEBUG"))
      fprintf(stderr, "aha: hash image: %u pairs in 2^%u slots (%u groups), %u deep entries in 2^%u slots, filter fill %u permille, k1 %#x\n",
              h.n_pairs, h.pair_log2, h.n_groups, h.n_deep, h.deep_log2, h.bloom_fill_permille, h.k1);
    return;
  }
  h.why = "no hash seed places the tables";
}

}  // namespace aha
