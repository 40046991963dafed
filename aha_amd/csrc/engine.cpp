// engine.cpp -- which kernels answer a call: the engines' setup on a device handle, the pipeline of one pass (match_v2:
// regions / full regions / slabs; byte-level, character-level, skip-ahead or prefix-filter traversal), and the sequence of
// passes of a device-resident batch (device_match: the prefix filter's back-off, the repeat with larger regions, the hand-over
// to the two-pass engine).  The C ABI around it is capi.cpp; what they share is handle.hpp.
#include "handle.hpp"

namespace ahai {

// ---- single-traversal engine: sizing, scratch, orchestration ----------------
constexpr size_t kLdsPerCU = 160 * 1024;

// The skip-ahead traversal (scan_skip.hip) can take a handle's plain byte-offset matches when the unit image has 22-bit
// bases and no key is a single unit.  It is OPT-IN (AHA_ENGINE=skip -- the unit image for every eligible key set like "unit" --
// or AHA_SKIP=1 beside the library's own choice): measured on cfg 3 its second kernel is bound by the scattered 16-byte text
// requests of its free-running lanes -- 3.75 ms per GiB against 2.24 for ku_traverse (profiles/r06_skip_engine.txt, DESIGN.md
// section 4.7) --, so no key set gets it by default.
bool skip_eligible(const aha_ac *ac) {
  const UnitImage &u = ac->unit;
  if (!u.ok || u.base_bits != 22 || u.unit_key || u.mark_bloom.empty()) return false;
  const char *eng = getenv("AHA_ENGINE");
  if (eng && strcmp(eng, "skip") != 0) return false;
  // (its walk has the header trip only: images whose states mostly own a header take ku_traverse<.., HB> -- v2_setup's rule)
  const char *hb = getenv("AHA_UNIT_HEADER_BESIDE");
  if (hb ? atoi(hb) != 0 : (!eng && (uint64_t)u.n_nfr * 5 >= u.n_states)) return false;  // (AHA_ENGINE=skip: the header trip)
  const char *sk = getenv("AHA_SKIP");
  if (sk) return atoi(sk) != 0;
  return eng != nullptr;
}

// Host-only plan: how much of the image the traversal kernel keeps in LDS.
// The pair engine (scan_pair.hip) takes a handle's plain byte-offset matches when the unit image has 22-bit bases, no key is a
// single unit, the two-unit paths have a perfect hash table and no trie path holds more than three END states of three
// units or more (the slots its first pass leaves for a deep walk's events).  AHA_ENGINE=pair builds the unit image for every
// eligible key set like "unit" does; AHA_PAIR=0 / 1 overrides the library's own choice.
bool pair_eligible(const aha_ac *ac) {
  const UnitImage &u = ac->unit;
  if (!u.ok || u.base_bits != 22 || u.unit_key || u.pair_tab.empty() || u.mark_bloom.empty() || u.deep_ends_max > kPairMaxDeepEnds ||
      ac->aut.max_key_len > 255)
    return false;
  for (uint32_t k = 0; k < ac->aut.n_keys; k++)
    if (ac->aut.key_cnt[k] >= 255u) return false;  // (an event's hits ride in one byte of the pair table's payload)
  const char *eng = getenv("AHA_ENGINE");
  if (eng && strcmp(eng, "pair") != 0) return false;
  const char *pr = getenv("AHA_PAIR");
  if (pr) return atoi(pr) != 0;
  return eng != nullptr;
}

void plan_engine(aha_ac *ac, const Placement &pl) {
  (void)pl;
  const size_t in_bytes = (size_t)(kV2Threads / 64) * 64 * (kV2Piece + 4);  // padded LDS input rows
  const size_t slot = ac->compact ? 4 : 8;
  // AHA_V2_BPC=2: two workgroups per CU (half the LDS each, twice the waves)
  const char *bpc = getenv("AHA_V2_BPC");
  ac->v2_bpc = (bpc && strcmp(bpc, "2") == 0) ? 2 : 1;
  const size_t budget = kLdsPerCU / ac->v2_bpc - in_bytes;
  // AHA_LDS_SLOTS=n caps the prefix (tests: forces the partial-prefix kernel on small automata)
  const char *cap_s = getenv("AHA_LDS_SLOTS");
  const size_t cap_slots = cap_s ? (size_t)std::max(256, atoi(cap_s)) & ~(size_t)255 : SIZE_MAX;
  if ((size_t)ac->n_slots * slot <= budget && ac->n_slots <= cap_slots) {  // the whole automaton lives in LDS
    ac->v2_lds_slots = ac->n_slots;
    return;
  }
  ac->v2_lds_slots = (uint32_t)std::min<size_t>(std::min<size_t>(budget / slot, cap_slots), ac->n_slots) & ~3u;
}

void v2_setup(aha_ac *ac) {
  const char *eng = getenv("AHA_ENGINE");
  if (eng && strcmp(eng, "v1") == 0) return;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ac->device) != hipSuccess || cus <= 0)
    return;
  if (v2_prepare(ac->compact, v2_lds_bytes(ac->v2_lds_slots, ac->compact)) != 0) return;
  // AHA_RESERVE_CUS=n: leave n CUs without a persistent traversal workgroup so that
  // collective (RCCL) kernels of an overlapped exchange can run beside it
  const char *rs = getenv("AHA_RESERVE_CUS");
  int reserve = rs ? atoi(rs) : 0;
  if (reserve < 0 || reserve >= cus) reserve = 0;
  ac->v2_grid = (uint32_t)(cus - reserve) * ac->v2_bpc;
  ac->v2_ok = true;
  if (ac->pf_d && filter_prepare() == 0 && upload(ac, ac->pf_bloom, &ac->fdev.bloom) == AHA_OK) {
    ac->fdev.d = ac->pf_d;
    ac->fdev.log2 = ac->pf_log2;
    ac->pf_cus = (uint32_t)(cus - reserve);
    ac->pf_ok = true;
  }
  // character-level engine: one step per UTF-8-shaped unit (unit.hpp).  One workgroup per CU: its LDS holds the root's
  // transitions of the whole alphabet.
  if (ac->unit.ok && ac->v2_bpc == 1 && unit_lds_bytes(ac->unit.n_syms) <= kLdsPerCU) {
    const Automaton &a = ac->aut;
    std::vector<uint32_t> info(ac->unit.end_key.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < info.size(); i++) {
      const int32_t k = ac->unit.end_key[i];
      if (k >= 0) info[i] = ac->key_info.empty() ? ((uint32_t)k | (std::min<uint32_t>(a.key_cnt[k], 255u) << 24)) : ac->key_info[k];
    }
    // the fused expansion's table: one gather gives an event's first hit and where the rest of its chain is
    uint32_t max_cnt = 0;
    for (uint32_t k = 0; k < a.n_keys; k++) max_cnt = std::max(max_cnt, a.key_cnt[k]);
    const char *upost = getenv("AHA_UNIT_POST");  // "regroup": the general post passes (tests)
    const bool fused = !ac->key_info.empty() && max_cnt <= kUFusedMaxChain && a.max_key_len < 65536 &&
                       !(upost && strcmp(upost, "regroup") == 0);
    std::vector<uint2> uend, uendc;
    if (fused) {
      uend.assign(ac->unit.end_key.size(), uint2{0, 0});
      uendc = uend;
      for (size_t i = 0; i < uend.size(); i++) {
        const int32_t k = ac->unit.end_key[i];
        if (k < 0) continue;
        const uint32_t co = ac->key_info[k] & 0xFFFFFFu, len = a.key_len[k], kc = a.key_kc[k] + 1u;
        uend[i] = uint2{(uint32_t)k | (len & 0xFFu) << 24, co | (len >> 8) << 24};
        uendc[i] = uint2{(uint32_t)k | (kc & 0xFFu) << 24, co | (kc >> 8) << 24};
      }
    }
    const uint64_t *us = nullptr;
    if (fused && upload(ac, uend, &ac->d_unit_end) == AHA_OK && upload(ac, uendc, &ac->d_unit_end_chars) == AHA_OK)
      ac->unit_fused = true;
    if (unit_prepare(ac->unit.n_syms) == 0 && upload(ac, ac->unit.slots, &us) == AHA_OK &&
        upload(ac, ac->unit.root, &ac->udev.root) == AHA_OK && upload(ac, ac->unit.tables, &ac->udev.tables) == AHA_OK &&
        upload(ac, info, &ac->d_unit_end_info) == AHA_OK) {
      ac->udev.slots = reinterpret_cast<const uint2 *>(us);
      ac->udev.n_slots = ac->unit.n_slots;
      ac->udev.big_lo = ac->unit.n_shared;
      ac->udev.n_low = ac->unit.n_low;
      ac->udev.g0 = ac->unit.g0;
      ac->udev.base_bits = ac->unit.base_bits;
      ac->udev.n_syms = ac->unit.n_syms;
      ac->udev.max_len = a.max_key_len;
      // text falls out of deep matches where many states own a fail header: then the header comes beside the probe (a second
      // load in every trip) instead of in a trip of its own -- -8.5 % on cfg 5, +3.5 % on cfg 3 (profiles/r04_two_walks.txt)
      const char *hb = getenv("AHA_UNIT_HEADER_BESIDE");  // 0 / 1: tests
      const char *eng2 = getenv("AHA_ENGINE");
      ac->udev.hdr_beside = hb ? (uint32_t)(atoi(hb) != 0)
                               : (uint32_t)(!(eng2 && strcmp(eng2, "skip") == 0) && (uint64_t)ac->unit.n_nfr * 5 >= ac->unit.n_states);
      ac->unit_ok = true;
      // the skip-ahead traversal over the same image: its filter over the two-unit paths (unit.hpp, MARKS)
      if (skip_eligible(ac) && !ac->udev.hdr_beside && skip_prepare(ac->unit.n_syms, ac->unit.mark_log2) == 0 &&
          upload(ac, ac->unit.mark_bloom, &ac->sdev.bloom) == AHA_OK) {
        ac->sdev.log2 = ac->unit.mark_log2;
        ac->sdev.k1 = ac->unit.pair_k1;
        ac->skip_ok = true;
      }
      // the pair engine: the same filter, the pair table with the events' payloads (what end_info holds for the state), its
      // displacement bytes
      if (pair_eligible(ac) && pair_prepare(ac->unit.n_syms, ac->unit.mark_log2, ac->unit.pair_groups) == 0) {
        std::vector<uint32_t> tab = ac->unit.pair_tab;
        for (size_t i = 0; i < tab.size(); i += 4) {
          const int32_t k = (int32_t)tab[i + 2];
          tab[i + 2] = (tab[i] == 0u || k < 0) ? 0u : (ac->key_info.empty() ? ((uint32_t)k | (std::min<uint32_t>(a.key_cnt[k], 255u) << 24)) : ac->key_info[k]);
        }
        const uint32_t *dt = nullptr;
        if (upload(ac, ac->unit.mark_bloom, &ac->pdev.bloom) == AHA_OK && upload(ac, tab, &dt) == AHA_OK &&
            upload(ac, ac->unit.pair_disp, &ac->pdev.disp) == AHA_OK) {
          ac->pdev.tab = reinterpret_cast<const uint4 *>(dt);
          ac->pdev.bloom_log2 = ac->unit.mark_log2;
          ac->pdev.k1 = ac->unit.pair_k1;
          ac->pdev.log2 = ac->unit.pair_log2;
          ac->pdev.groups = ac->unit.pair_groups;
          ac->pair_ok = true;
        }
      }
    }
  }
}

int32_t v2_reserve(aha_ac *ac, Scratch *sc, int i, size_t bytes) {
  Buf &b = sc->v2buf[i];
  if (b.bytes >= bytes) return AHA_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  size_t want = bytes + bytes / 8 + 256;
  HIPCHK(ac, hipMalloc(&b.p, want));
  b.bytes = want;
  return AHA_OK;
}

// Pipeline of one call, a function of the call alone (the handle keeps no history):
//   kRegions      per-chunk event regions sized from the caller's capacity (a hit is an event or hangs on one, so the
//                 batch has at most `cap` events the caller can take: twice the average per chunk, plus slack)
//   kFullRegions  regions of one event per input byte (cannot overflow); taken at once when cap says the caller
//                 expects more than one hit per 4 bytes, else after a region overflowed
//   kSlabs        slab + sort pipeline: separator filter, fewer than 16 hits per chunk expected (its cost follows the
//                 events, not the chunks), or regions beyond the temp bound
enum V2Mode { kRegions = 0, kFullRegions = 1, kSlabs = 2 };
// Field widths of the 4-byte exchange stream for this automaton (include/aha_hip.h): the key id needs vb bits, a key's
// length lb bits (in bytes; its length in characters is not longer); when at least 6 bits are left for the step of `end`
// the word carries the length, else only id and a 12-bit step (ids below 2^20) and the receiver looks the length up.
StreamFmt stream_fmt(const aha_ac *ac) {
  auto bits = [](uint32_t x) {
    uint32_t b = 0;
    while (x) {
      b++;
      x >>= 1;
    }
    return b;
  };
  const uint32_t vb = std::max(1u, bits(ac->aut.n_keys ? ac->aut.n_keys - 1 : 0)), lb = std::max(1u, bits(ac->aut.max_key_len));
  // (a step field below 10 bits makes every gap of 1 KiB an exception -- 4 more bytes on the link --, which costs a
  // sparse hit stream more than the key-length lookup on arrival saves: then the word carries id and a 12-bit step only)
  if (vb + lb + 10 <= 32) return StreamFmt{std::min(12u, 32 - vb - lb), lb};
  return StreamFmt{12, 0};
}

constexpr uint64_t kV2MaxRegionBytes = 48ull << 30;

// returns AHA_OK, an error, +1 when the caller must fall back to the two-pass engine, +2 when a region overflowed
// the pinned words a call's verdict and totals come back in, and their device address
int32_t ensure_h_v2(aha_ac *ac, Scratch *sc) {
  if (sc->h_v2) return AHA_OK;
  HIPCHK(ac, hipHostMalloc((void **)&sc->h_v2, 5 * sizeof(unsigned long long), hipHostMallocDefault));
  if (hipHostGetDevicePointer((void **)&sc->h_v2_dev, sc->h_v2, 0) != hipSuccess) {
    (void)hipGetLastError();
    sc->h_v2_dev = nullptr;  // (then the words come back by a copy)
  }
  return AHA_OK;
}

int32_t match_v2(aha_ac *ac, Scratch *sc, MatchArgs &M1, hipStream_t s, uint64_t *n_hits, V2Mode mode) {
  const uint64_t N = M1.n_bytes;
  const uint32_t Lmax = ac->aut.max_key_len;
  uint64_t s_min = std::max<uint64_t>(64, ((8ull * Lmax + 63) / 64) * 64);
  if (s_min > kV2MaxS) return 1;
  const char *de = getenv("AHA_DIRECT");
  uint64_t lanes = (uint64_t)ac->v2_grid * kV2Threads;
  uint64_t S = ((N + lanes - 1) / lanes + 63) / 64 * 64;
  S = std::min<uint64_t>(std::max<uint64_t>(S, s_min), kV2MaxS);
  // the prefix-filter engine: no separator filter, the event regions; a wave takes a chunk of 4, 8, 16 or 32 KiB -- the larger,
  // the fuller its batches of 64 candidates and the fewer chunks the post passes see (cfg 2 at 64 MiB: 0.136 ms with 4 KiB,
  // 0.117 with 16 KiB) -- while every wave of the device still has one, and while the image, if it fits LDS at all, still
  // fits beside the longer candidate lists.
  // (a call with char offsets: kf_filter finds out on its way whether the batch is plain ASCII -- then a character count is a
  // byte count --, otherwise kf_walk counts the continuation bytes of its chunk; the table it keeps for that counts against LDS)
  const bool filt = ac->pf_ok && !ac->unit_ok && !M1.sep && !M1.no_filter && mode != kSlabs && !(de && strcmp(de, "0") == 0);
  if (filt) {
    S = 4096;
    while (S < kV2MaxS && N / (2 * S) >= (uint64_t)ac->pf_cus * 16 &&
           (filter_image_in_lds(ac->n_slots, (uint32_t)(2 * S), M1.chars != 0) || !filter_image_in_lds(ac->n_slots, 4096, M1.chars != 0)))
      S *= 2;
    if (const char *fc = getenv("AHA_FILTER_CHUNK")) {  // the tests' way to the larger chunks without a batch of 128+ MiB
      const long v = atol(fc);
      if (v == 4096 || v == 8192 || v == 16384 || v == 32768) S = (uint64_t)v;
    }
  }
  // the pair engine (scan_pair.hip): byte offsets, no separator filter, the event regions with its tile as the chunk
  const bool want_pair = ac->pair_ok && !M1.chars && !M1.sep && !M1.no_pair && mode != kSlabs && N >= 64 && !(de && strcmp(de, "0") == 0) &&
                         ac->pair_off.load(std::memory_order_relaxed) < 3;
  if (want_pair) S = pair_tile_bytes();
  V2Args M{};
  M.text = M1.text;
  M.doc_off = M1.doc_off;
  M.n_docs = M1.n_docs;
  M.n_bytes = N;
  M.S = (uint32_t)S;
  M.n_chunks = (N + S - 1) / S;
  if (M.n_chunks > 0xFFFFFFFFull) return 1;
  M.lds_slots = ac->v2_lds_slots;
  M.chars = M1.chars;
  M.sep = M1.sep;
  memcpy(M.sep_block, M1.sep_block, sizeof(M.sep_block));
  M.out = M1.out;
  M.cap = M1.cap;
  M.doc_hit_off = M1.doc_hit_off;
  M.unit_bb = ac->unit.base_bits;
  // plain mode (byte offsets or char offsets, no separator filter): per-chunk event regions, no sort
  const bool dense = M1.cap / 4 > N / 16;          // more than one hit per 4 input bytes expected
  const bool sparse = M1.cap < 16ull * M.n_chunks;  // fewer than 16 hits per chunk expected
  if (de && strcmp(de, "0") == 0) mode = kSlabs;
  // (the character-level engine leaves its events wave by wave and expands them group by group: its cost follows the
  // events too, so a handle that has it keeps the regions for sparse batches)
  if (M.sep || (mode == kRegions && sparse && !ac->unit_ok && !filt)) mode = kSlabs;
  if (mode == kRegions && dense) mode = kFullRegions;
  uint64_t stride = S;
  // twice the average the caller allows for, plus a slack of 1/64 of the chunk (64 events at 4 KiB): 16 bytes per hit of
  // capacity + 1/8 byte per input byte
  if (mode == kRegions) stride = std::min<uint64_t>(S, 2 * (M1.cap / M.n_chunks) + std::max<uint64_t>(16, S / 64));
  // bytes per event of the regions: 8 (byte-level engine), 12 (character-level, fused expansion), 12 + 8 (general passes)
  const uint64_t rec_bytes = want_pair ? 8 : (ac->unit_ok ? (ac->unit_fused ? 12 : 20) : 8);
  if (mode != kSlabs && M.n_chunks * stride * rec_bytes > kV2MaxRegionBytes) mode = kSlabs;
  const bool direct = mode != kSlabs;
  const uint64_t waves = (uint64_t)ac->v2_grid * (kV2Threads / 64);
  M.direct = direct ? 1 : 0;
  M.dense_hits = dense ? 1 : 0;
  M.ev_stride = (uint32_t)stride;
  M.ev_cap = direct ? 0 : ((M1.cap + waves * kV2Slab + kV2Slab) / kV2Slab) * kV2Slab;
  const uint64_t n_slabs = M.ev_cap / kV2Slab + 2;
  const uint64_t n_blk = std::max<uint64_t>((M.n_chunks + 255) / 256, (M.ev_cap + 255) / 256) + 2;
  const uint64_t n_reg = direct ? M.n_chunks * M.ev_stride : 0;
  // byte offsets through the event regions: the character-level traversal where the key set has a unit image
  const bool pair = want_pair && direct;
  if (want_pair && !pair) return match_v2(ac, sc, (M1.no_pair = 1, M1), s, n_hits, mode);  // (regions beyond the temp bound: the chunk was the pair engine's)
  const bool unit = ac->unit_ok && direct && !pair;
  // ... started only at the marks of a first, stateless pass where the handle has the filter for it (byte offsets)
  // (not a batch below one piece of the marking pass: its lanes ask for 16-byte windows wherever they stand)
  const bool skip = unit && ac->skip_ok && !M.chars && N >= 64;
  int32_t rc;
  // deep candidates the pair engine has room for (cfg 3: one per ~120 bytes), in segments of its waves: 256 each at least
  const uint64_t cand_cap = pair ? std::max<uint64_t>(N / 48 + 4096, (uint64_t)ac->v2_grid * (kV2Threads / 64) * 256) : 0;
  size_t sizes[26] = {M.ev_cap * 16,      M.ev_cap * 16,      M.ev_cap * 4,     direct ? 0 : n_slabs * 4,
                      M.n_chunks * 4,     (M.n_docs + 1) * 4, M.n_chunks * 8,   n_blk * 8,
                      n_blk * 8,          kCursorBytes,       M.chars ? M.ev_cap * 4 : 0, M.chars ? M.ev_cap * 4 : 0,
                      (M.chars || pair) ? M.n_chunks * 4 : 0, (M.chars || pair) ? M.n_chunks * 4 : 0, M.chars ? (M.n_docs + 1) * 4 : 0,
                      M.chars ? M.n_chunks * 8 : 0,
                      (unit && ac->unit_fused) ? 0 : n_reg * 8, 0 /* [17]: aligned copy of an unaligned corpus */,
                      direct ? M.n_chunks * 4 : 0, direct ? M.n_chunks * 8 : 0,
                      unit ? (M.n_docs + 1) * 4 : 0, unit ? n_reg * 12 : 0,
                      filt ? ((N + 63) / 64 + 2) * 8 : (skip ? skip_bitmap_bytes(N) : (pair ? M.n_chunks * 4 : 0)) /* [22]: candidate bitmap / marks / tile records */,
                      filt ? M.n_chunks * filter_chunk_rec_bytes() : (pair ? pair_cand_bytes(cand_cap) : 0) /* [23] */,
                      pair ? pair_walk_bytes(cand_cap) : 0 /* [24] */, 0};
  for (int i = 0; i < 26; i++) {
    if (!sizes[i]) continue;
    if ((rc = v2_reserve(ac, sc, i, sizes[i]))) {
      // no room for the event regions (someone else holds the HBM): the slab pipeline needs far less temp
      if ((i == 16 || i == 21) && mode != kSlabs) {
        (void)hipGetLastError();
        return match_v2(ac, sc, M1, s, n_hits, kSlabs);
      }
      return rc;
    }
  }
  M.ev = (uint4 *)sc->v2buf[0].p;
  M.sorted_ev = (uint4 *)sc->v2buf[1].p;
  M.sorted_cnt = (uint32_t *)sc->v2buf[2].p;
  M.slab_used = (uint32_t *)sc->v2buf[3].p;
  M.ev_cnt = (uint32_t *)sc->v2buf[4].p;
  M.doc_ev_rank = (uint32_t *)sc->v2buf[5].p;
  M.ev_base = (uint64_t *)sc->v2buf[6].p;
  M.blk_a = (uint64_t *)sc->v2buf[7].p;
  M.blk_b = (uint64_t *)sc->v2buf[8].p;
  if (sc->cursor_buf != sc->v2buf[9].p) {  // (a new buffer: nothing is known about its words)
    sc->cursor_buf = sc->v2buf[9].p;
    sc->cursor_dirty = true;
  }
  M.ev_aux = (uint32_t *)sc->v2buf[10].p;
  M.sorted_aux = (uint32_t *)sc->v2buf[11].p;
  M.lead_cnt = (uint32_t *)sc->v2buf[12].p;
  M.chunk_doc0 = (uint32_t *)sc->v2buf[13].p;
  M.doc_lead_rank = (uint32_t *)sc->v2buf[14].p;
  M.lead_base = (uint64_t *)sc->v2buf[15].p;
  M.evd = (uint2 *)sc->v2buf[16].p;
  M.evg = (uint32_t *)sc->v2buf[21].p;
  M.doc_hit_rank = (uint32_t *)sc->v2buf[20].p;
  M.chunk_hits = (uint32_t *)sc->v2buf[18].p;
  M.hit_base = (uint64_t *)sc->v2buf[19].p;
  if ((rc = ensure_h_v2(ac, sc))) return rc;
  // the region pipelines' last kernel -- the per-document offsets -- leaves the host's five words itself
  M.publish = (direct && M.doc_hit_off && sc->h_v2_dev) ? sc->h_v2_dev : nullptr;
  // the call's counter block; the kernel that publishes also clears the other block for the next call
  static const bool always_clear = getenv("AHA_CURSOR_MEMSET") != nullptr;  // (lab: the memset in front of every call, as before round 6)
  if (sc->cursor_dirty || always_clear) {
    HIPCHK(ac, hipMemsetAsync(sc->v2buf[9].p, 0, kCursorBytes, s));
    sc->cursor_phase = 0;
  }
  M.cursor = (unsigned long long *)sc->v2buf[9].p + 16 * sc->cursor_phase;
  M.totals = (uint64_t *)M.cursor + 2;
  M.clear_next = M.publish ? (unsigned long long *)sc->v2buf[9].p + 16 * (sc->cursor_phase ^ 1u) : nullptr;
  sc->cursor_dirty = true;  // (until this call has run to its end)

  const bool prof = ac->profiling.load() && sc->ev_ready;
  // device-resident offsets nobody has looked at yet: validated here, in front of the traversal; a bad verdict lands in
  // cursor[1], where the traversal and every post pass look first (no read-back before the launch: -30 us per call)
  if (M1.check_docs) launch_check_docs(M.doc_off, M.n_docs, N, nullptr, M.cursor + 1, s);
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[0], s));
  const uint64_t n_tiles = (M.n_chunks + kV2Threads - 1) / kV2Threads;
  DevAut post = ac->dev;
  if (pair) {
    post.end_info = ac->d_unit_end_info;  // its deep events carry bases of the unit image
    post.compact = 1;
    pair_launch(ac->pdev, ac->udev, post, M, sc->v2buf[22].p, sc->v2buf[23].p, sc->v2buf[24].p, cand_cap, ac->v2_grid,
                prof ? (void *)sc->ev[2] : nullptr, s);  // (profiling only: ms_count = the pair pass, ms_scan = the deep walks)
  } else if (unit) {
    post.end_info = ac->d_unit_end_info;  // events carry bases of the unit image
    post.compact = 1;
    if (skip) {
      skip_launch_mark(ac->sdev, M, sc->v2buf[22].p, ac->v2_grid, s);
      if (prof) HIPCHK(ac, hipEventRecord(sc->ev[2], s));  // (profiling only: ms_count = the marks, ms_scan = the walk)
      skip_launch_traverse(ac->udev, M, sc->v2buf[22].p, (uint32_t)std::min<uint64_t>(ac->v2_grid, n_tiles), s);
    } else {
      unit_launch_traverse(ac->udev, M, (uint32_t)std::min<uint64_t>(ac->v2_grid, n_tiles), s);
    }
  } else if (filt) {
    // filter (one bit per byte position), then the candidates' goto walks, a wave per chunk
    unsigned long long *non_ascii = M1.chars ? M.cursor + 6 : nullptr;
    filter_launch_filter(ac->fdev, M, sc->v2buf[22].p, sc->v2buf[23].p, non_ascii, ac->pf_cus, s);
    if (prof) HIPCHK(ac, hipEventRecord(sc->ev[2], s));  // (profiling only: ms_count = the filter, ms_scan = the walks)
    filter_launch_walk(ac->dev, M, sc->v2buf[22].p, sc->v2buf[23].p, non_ascii, ac->pf_cus, s);
  } else {
    v2_launch_traverse(ac->dev, M, (uint32_t)std::min<uint64_t>(ac->v2_grid, n_tiles), s);
  }
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[1], s));
  if (direct) {
    // (no event between the traversal and the post passes of this pipeline: a record costs ~5 us of stream time, and ev[1]
    // stands for ev[2] in the timing below)
    if (unit && ac->unit_fused) {  // the traversal counted the hits: bases, then the expansion straight from the wave-ordered events
      v2_launch_hit_scan(M, s);
      if (M.chars) v2_launch_lead_scan(M, s);  // characters before every chunk (the traversal counted them per chunk)
      if (prof) HIPCHK(ac, hipEventRecord(sc->ev[3], s));
      unit_launch_expand(M.chars ? ac->d_unit_end_chars : ac->d_unit_end, post, M, 2u * ac->v2_grid, s);
    } else {
      if (unit) unit_launch_regroup(post, M, s);  // the wave-ordered events back into the chunks' regions, counted
      v2_launch_direct_post(post, M, s, prof ? (void *)sc->ev[3] : nullptr, unit || filt || pair);  // (kf_walk and kp_pairs count like ku_regroup)
    }
  } else {
    v2_launch_chunk_scan(M, s);
    if (prof) HIPCHK(ac, hipEventRecord(sc->ev[2], s));
    v2_launch_sort(ac->dev, M, M.ev_cap, s);
    if (prof) HIPCHK(ac, hipEventRecord(sc->ev[3], s));
    v2_launch_expand(ac->dev, M, M.ev_cap, s);
  }
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[4], s));
  HIPCHK(ac, hipGetLastError());
  if (M.publish) {
    // (done by k2d_doc_offsets / ku_doc_offsets)
  } else if (sc->h_v2_dev) {
    launch_publish_words((const unsigned long long *)M.cursor, sc->h_v2_dev, 5, s);
  } else {
    HIPCHK(ac, hipMemcpyAsync(sc->h_v2, M.cursor, 5 * 8, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ac, hipStreamSynchronize(s));
  if (M.clear_next) {  // every kernel of the call has run: the other block is clear
    sc->cursor_dirty = false;
    sc->cursor_phase ^= 1u;
  }
#ifdef AHA_EXPAND_CLK
  {  // (lab: the dense expansion's phases, clock cycles summed over the sampled chunks' waves)
    unsigned long long w[16];
    (void)hipMemcpy(w, M.cursor, sizeof(w), hipMemcpyDeviceToHost);
    if (w[15])
      fprintf(stderr, "k2d_expand_dense, %llu sampled chunks, %.1f windows each; cycles per window: A %.0f  B2 %.0f  wait %.0f  fill %.0f  wall-clock ticks of 10 ns per chunk %.0f; clocks per chunk %.0f\n",
              w[15], (double)w[13] / w[15], (double)w[8] / w[13], (double)w[9] / w[13], (double)w[10] / w[13], (double)w[11] / w[13], (double)w[12] / w[15],
              (double)w[14] / w[15]);
    if (w[7]) fprintf(stderr, "  blocks sampled %llu: %.1f chunks each, %.0f clocks from the kernel's first instruction to its stores' acknowledgement\n", w[7], (double)w[6] / w[7], (double)w[5] / w[7]);
  }
#endif
#ifdef AHA_SK_STATS
  if (skip) {
    unsigned long long w[16];
    (void)hipMemcpy(w, M.cursor, sizeof(w), hipMemcpyDeviceToHost);
    fprintf(stderr, "ks_traverse: %llu lane-trips (%.4f per byte), %llu jumps, %llu fresh, %llu wave-trips (%.1f %% of the lane slots used)\n", w[8],
            (double)w[8] / (double)N, w[9], w[11], w[10], 100.0 * (double)w[8] / (64.0 * (double)w[10]));
  }
#endif
  if (sc->h_v2[1] >= 16) {  // the offsets are not what the call says (k_check_docs): nothing was indexed with them
    if (sc->h_v2[1] & 1) {
      tls_err = "doc offsets: need doc_offsets[0] = 0, ascending, doc_offsets[n_docs] = n_bytes";
      return AHA_E_INVALID;
    }
    tls_err = aha_strerror(AHA_E_TOO_LONG);
    return AHA_E_TOO_LONG;
  }
  if (sc->h_v2[1] == 3 && pair) {  // the pair engine gave the batch up (documents of a few bytes, a piece dense with events): engine 4 takes it
    ac->pair_off.fetch_add(1, std::memory_order_relaxed);  // (three times: the handle stops trying)
    M1.no_pair = 1;
  }
  if (sc->h_v2[1] == 3) return 3;  // the prefix-filter engine gave up (candidates too dense, nested keys): the caller repeats without it
                                   // (check_docs stays set: the repeat validates the offsets again -- 5 us -- rather than trust
                                   // that no plain store of a hand-back overwrote a bad verdict)
  M1.check_docs = 0;  // (looked at: a repeated pass or the two-pass engine need not look again)
  if (sc->h_v2[1] == 2) return 2;  // a chunk's event region overflowed: the caller repeats with full-size regions
  if (sc->h_v2[1]) return 1;  // event temp exhausted (cap too small): exact count via the two-pass engine
  *n_hits = sc->h_v2[2];
  if (prof) {
    aha_timing t;
    memset(&t, 0, sizeof(t));
    t.struct_size = sizeof(t);
    t.engine = pair ? 7 : (skip ? 6 : (unit ? 4 : (filt ? 5 : 2)));
    t.chunk_bytes = M.S;
    t.n_kernels = 9;
    (void)hipEventElapsedTime(&t.ms_total, sc->ev[0], sc->ev[4]);
    (void)hipEventElapsedTime(&t.ms_count, sc->ev[0], (filt || skip || pair) ? sc->ev[2] : sc->ev[1]);
    if (filt || skip || pair) (void)hipEventElapsedTime(&t.ms_scan, sc->ev[2], sc->ev[1]);
    if (direct) {
      (void)hipEventElapsedTime(&t.ms_aux, sc->ev[1], sc->ev[3]);
    } else {
      (void)hipEventElapsedTime(&t.ms_scan, sc->ev[1], sc->ev[2]);
      (void)hipEventElapsedTime(&t.ms_aux, sc->ev[2], sc->ev[3]);
    }
    (void)hipEventElapsedTime(&t.ms_write, sc->ev[3], sc->ev[4]);
    t.n_chunks = M.n_chunks;
    t.n_hits = *n_hits;
    publish_timing(ac, t);
  }
  return AHA_OK;
}

// the events of a scratch set are created by the first profiled call that leases it
int32_t ready_events(aha_ac *ac, Scratch *sc) {
  if (!ac->profiling.load() || sc->ev_ready) return AHA_OK;
  // (timing only -- nothing synchronizes with them, the call ends in hipStreamSynchronize --: created without the system-scope
  // fence of a default event, i.e. without a cache write-back and invalidation between the kernels they stand between.  A
  // 64 MiB call of cfg 2: 86 us against 89.5 with the fences (AHA_EVENT_FENCE=1) -- and 65 without any events: the five
  // records themselves cost ~4 us each, profiles/r06_cfg2_fixed_costs.txt)
  static const unsigned ev_flags = getenv("AHA_EVENT_FENCE") ? hipEventDefault : hipEventDisableSystemFence;
  for (auto &e : sc->ev) HIPCHK(ac, hipEventCreateWithFlags(&e, ev_flags));
  sc->ev_ready = true;
  return AHA_OK;
}

int32_t device_match(aha_ac *ac, Scratch *sc, const uint8_t *d_corpus, const uint64_t *d_doc_offsets, uint64_t n_docs,
                     uint64_t n_bytes, const aha_match_params *params, aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets,
                     uint64_t *n_hits, void *stream, bool offsets_checked, const PackOut *pk, bool *packed) {
  if (!ac || !n_hits || !d_doc_offsets) return AHA_E_INVALID;
  if (ac->device < 0) {
    tls_err = aha_strerror(AHA_E_NO_DEVICE);
    return AHA_E_NO_DEVICE;
  }
  if (cap && !d_out) return AHA_E_INVALID;
  DeviceGuard g(ac->device);
  hipStream_t s = (hipStream_t)stream;
  MatchArgs M{};
  int longest = 0;
  int32_t rc = fill_params(ac, params, M, &longest);
  if (rc) return rc;
  if ((rc = ready_events(ac, sc))) return rc;
  *n_hits = 0;
  auto check_now = [&]() -> int32_t {
    // the offsets live in HBM: one small kernel and an 4-byte read-back before anything indexes with them
    int32_t rc2;
    if ((rc2 = v2_reserve(ac, sc, 9, kCursorBytes))) return rc2;
    if ((rc2 = ensure_h_v2(ac, sc))) return rc2;
    uint32_t *flag = (uint32_t *)sc->v2buf[9].p + 64;  // (the third block: odd words)
    HIPCHK(ac, hipMemsetAsync(flag, 0, 4, s));
    launch_check_docs(d_doc_offsets, n_docs, n_bytes, flag, nullptr, s);
    HIPCHK(ac, hipMemcpyAsync(sc->h_v2, flag, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ac, hipStreamSynchronize(s));
    const uint32_t bad = (uint32_t)sc->h_v2[0];
    if (bad & 1u) {
      tls_err = "doc offsets: need doc_offsets[0] = 0, ascending, doc_offsets[n_docs] = n_bytes";
      return AHA_E_INVALID;
    }
    if (bad & 2u) {
      tls_err = aha_strerror(AHA_E_TOO_LONG);
      return AHA_E_TOO_LONG;
    }
    return AHA_OK;
  };
  // The single-traversal pipelines validate on the device in front of their traversal (match_v2); every other path -- an
  // empty batch, match_longest, the two-pass engine -- reads the verdict back first.
  const bool defer_check = !offsets_checked && ac->v2_ok && !longest && n_bytes != 0;
  if (!offsets_checked && !defer_check && (rc = check_now())) return rc;
  M.check_docs = defer_check ? 1 : 0;
  if (n_bytes == 0) {
    if (d_doc_hit_offsets)
      HIPCHK(ac, hipMemsetAsync(d_doc_hit_offsets, 0, (n_docs + 1) * sizeof(uint64_t), s));
    HIPCHK(ac, hipStreamSynchronize(s));
    return AHA_OK;
  }
  if (!d_corpus) return AHA_E_INVALID;
  if (reinterpret_cast<uintptr_t>(d_corpus) % 16 != 0) {
    // the kernels read the corpus in aligned 16-byte pieces: an unaligned view (a slice of a larger buffer) is copied
    // once, device to device, into the handle's scratch (~0.7 ms per GiB: about a fifth of the match itself)
    if ((rc = v2_reserve(ac, sc, 17, n_bytes + 64))) return rc;
    HIPCHK(ac, hipMemcpyAsync(sc->v2buf[17].p, d_corpus, n_bytes, hipMemcpyDeviceToDevice, s));
    d_corpus = (const uint8_t *)sc->v2buf[17].p;
  }
  M.text = d_corpus;
  M.doc_off = d_doc_offsets;
  M.n_docs = n_docs;
  M.n_bytes = n_bytes;
  M.out = d_out;
  M.cap = cap;
  M.doc_hit_off = d_doc_hit_offsets;
  if (longest) {
    // match_longest: count -> scan -> write, like the two-pass engine (kernels.hip)
    int mode = longest == 1 ? 1 : (M.chars ? 3 : 2);
    if ((rc = ensure_scratch(ac, sc, 1, 1, n_docs))) return rc;
    if (mode == 2) {
      // the chunked form is exact only for text without NUL bytes (kernels.hip, k_has_nul): look first
      HIPCHK(ac, hipMemsetAsync(sc->d_totals, 0, 2 * sizeof(uint64_t), s));
      launch_has_nul(d_corpus, n_bytes, sc->d_totals + 1, s);
      HIPCHK(ac, hipMemcpyAsync(sc->h_totals, sc->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      HIPCHK(ac, hipStreamSynchronize(s));
      M.has_nul = sc->h_totals[1] ? 1 : 0;  // the chunks' warm-ups then reach back past the NULs they cross (kernels.hip)
    }
    M.chunk = 1024;
    while (M.chunk < 16ull * ac->aut.max_key_len && M.chunk < (1u << 20)) M.chunk *= 2;  // the warm-up is 2 * Lmax
    M.n_chunks = (n_bytes + M.chunk - 1) / M.chunk;
    const uint64_t units = mode == 2 ? M.n_chunks : n_docs + 1;
    uint64_t n_blocks = (units + kBlock - 1) / kBlock;
    if ((rc = ensure_scratch(ac, sc, units, n_blocks, n_docs))) return rc;
    M.counts = sc->d_counts;
    M.leads = sc->d_leads;
    M.blk_hits = sc->d_blk_hits;
    M.blk_leads = sc->d_blk_leads;
    M.docg = sc->d_docg;
    M.totals = sc->d_totals;
    const int chars = M.chars;
    if ((rc = ensure_stale(ac))) return rc;
    launch_longest(ac->dev_longest, M, mode, false, s);
    if (mode == 2 && M.has_nul) {  // a chunk whose warm-up would not end (a NUL every few bytes) gave up: document by document
      HIPCHK(ac, hipMemcpyAsync(sc->h_totals, sc->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      HIPCHK(ac, hipStreamSynchronize(s));
      if (sc->h_totals[1] == 2) {
        mode = 3;
        const uint64_t units3 = n_docs + 1, blocks3 = (units3 + kBlock - 1) / kBlock;
        if ((rc = ensure_scratch(ac, sc, units3, blocks3, n_docs))) return rc;
        M.counts = sc->d_counts;
        M.leads = sc->d_leads;
        M.blk_hits = sc->d_blk_hits;
        M.blk_leads = sc->d_blk_leads;
        M.docg = sc->d_docg;
        M.totals = sc->d_totals;
        n_blocks = blocks3;
        launch_longest(ac->dev_longest, M, mode, false, s);
      }
    }
    M.chars = 0;  // the block scan has no lead counts to scan here
    launch_scan_blocks(M, n_blocks, s);
    M.chars = chars;
    launch_longest(ac->dev_longest, M, mode, true, s);
    HIPCHK(ac, hipGetLastError());
    HIPCHK(ac, hipMemcpyAsync(sc->h_totals, sc->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(ac, hipStreamSynchronize(s));
    *n_hits = sc->h_totals[0];
    if (*n_hits > cap) {
      tls_err = "output buffer too small";
      return AHA_E_CAPACITY;
    }
    return AHA_OK;
  }
  uint32_t repeats = 0;  // passes thrown away (aha_timing.repeats)
  if (ac->v2_ok) {
    // a handle whose batches keep coming back from the prefix-filter engine (text dense with key starts) skips it for 2, 4,
    // .. 64 calls before it tries again: a batch that is handed back has paid for the filter and part of the walks
    const int pm = M.chars ? 1 : 0;  // (calls with char offsets keep their own count)
    if (ac->pf_ok) {  // (calls on one handle may run side by side: the count goes down by compare-exchange, never below 0)
      uint32_t v = ac->pf_skip[pm].load(std::memory_order_relaxed);
      while (v && !ac->pf_skip[pm].compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {
      }
      if (v) M.no_filter = 1;
    }
    const bool tried = ac->pf_ok && !M.no_filter;
    rc = match_v2(ac, sc, M, s, n_hits, kRegions);
    if (rc == 3) {  // the prefix-filter engine handed the batch back: once more on the byte-level engine
      repeats++;
      M.no_filter = 1;
      const uint32_t streak = std::min(ac->pf_streak[pm].fetch_add(1, std::memory_order_relaxed) + 1, 6u);
      ac->pf_skip[pm].store(1u << streak, std::memory_order_relaxed);
      rc = match_v2(ac, sc, M, s, n_hits, kRegions);
    } else if (tried && rc == AHA_OK) {
      ac->pf_streak[pm].store(0, std::memory_order_relaxed);
    }
    if (rc == 2) {  // denser than cap said: regions of one event per byte
      repeats++;
      rc = match_v2(ac, sc, M, s, n_hits, kFullRegions);
    }
    if (rc == 3) {
      M.no_filter = 1;
      rc = match_v2(ac, sc, M, s, n_hits, kFullRegions);
    }
    if (rc == 2) rc = match_v2(ac, sc, M, s, n_hits, kSlabs);  // (not reached: full-size regions cannot overflow)
    if (rc == AHA_OK && repeats) note_repeats(ac, repeats);
    if (rc < 0) return rc;
    if (rc == AHA_OK) {
      if (*n_hits > cap) {
        tls_err = "output buffer too small";
        return AHA_E_CAPACITY;
      }
      return AHA_OK;
    }
    *n_hits = 0;  // rc == 1: fall through to the two-pass engine
    repeats++;
    if (M.check_docs && (rc = check_now())) return rc;  // (no single-traversal pass has looked at the offsets)
  }
  M.chunk = ac->chunk;
  // warm-up is Lmax-1 bytes per chunk: keep it a small fraction of the chunk
  while (M.chunk < 8ull * ac->aut.max_key_len && M.chunk < (1u << 20)) M.chunk *= 2;
  M.n_chunks = (n_bytes + M.chunk - 1) / M.chunk;
  const uint64_t n_blocks = (M.n_chunks + kBlock - 1) / kBlock;
  if ((rc = ensure_scratch(ac, sc, M.n_chunks, n_blocks, n_docs))) return rc;
  M.counts = sc->d_counts;
  M.leads = sc->d_leads;
  M.blk_hits = sc->d_blk_hits;
  M.blk_leads = sc->d_blk_leads;
  M.docg = sc->d_docg;
  M.totals = sc->d_totals;
  M.out = d_out;
  M.cap = cap;
  M.doc_hit_off = d_doc_hit_offsets;

  const bool prof = ac->profiling.load() && sc->ev_ready;
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[0], s));
  launch_count(ac->dev, M, s);
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[1], s));
  launch_scan_blocks(M, n_blocks, s);
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[2], s));
  if (M.chars) launch_docg(M, s);
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[3], s));
  launch_write(ac->dev, M, s);
  if (prof) HIPCHK(ac, hipEventRecord(sc->ev[4], s));
  HIPCHK(ac, hipGetLastError());
  HIPCHK(ac, hipMemcpyAsync(sc->h_totals, sc->d_totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  HIPCHK(ac, hipStreamSynchronize(s));
  *n_hits = sc->h_totals[0];
  if (prof) {
    aha_timing t;
    memset(&t, 0, sizeof(t));
    t.struct_size = sizeof(t);
    t.n_kernels = M.chars ? 4 : 3;
    (void)hipEventElapsedTime(&t.ms_total, sc->ev[0], sc->ev[4]);
    (void)hipEventElapsedTime(&t.ms_count, sc->ev[0], sc->ev[1]);
    (void)hipEventElapsedTime(&t.ms_scan, sc->ev[1], sc->ev[2]);
    (void)hipEventElapsedTime(&t.ms_aux, sc->ev[2], sc->ev[3]);
    (void)hipEventElapsedTime(&t.ms_write, sc->ev[3], sc->ev[4]);
    t.n_chunks = M.n_chunks;
    t.n_hits = *n_hits;
    t.engine = 1;
    t.chunk_bytes = M.chunk;
    t.repeats = repeats;
    publish_timing(ac, t);
  }
  if (*n_hits > cap) {
    tls_err = "output buffer too small";
    return AHA_E_CAPACITY;
  }
  return AHA_OK;
}

}  // namespace ahai
