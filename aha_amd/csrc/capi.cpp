// capi.cpp -- the C ABI of libaha_hip.so (include/aha_hip.h): host-side
// orchestration of compile (CPU) and match (HIP kernels).  There is NO CPU
// matching fallback in this library: without a usable HIP device every match
// entry point fails with AHA_E_NO_DEVICE.
#include "handle.hpp"

using namespace ahai;

namespace ahai {
// last error text of the calling thread (aha_last_error): calls on one handle may run concurrently
thread_local std::string tls_err;

void free_scratch(Scratch *sc, bool all) {
  for (auto &b : sc->v2buf) {
    if (b.p) (void)hipFree(b.p);
    b = Buf();
  }
  for (auto &b : sc->hostbuf) {
    if (b.p) (void)hipFree(b.p);
    b = Buf();
  }
  void **scratch[] = {(void **)&sc->d_counts, (void **)&sc->d_leads, (void **)&sc->d_blk_hits, (void **)&sc->d_blk_leads,
                      (void **)&sc->d_docg};
  for (void **p : scratch) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  sc->cap_chunks = sc->cap_blocks = sc->cap_docs = 0;
  if (!all) return;
  if (sc->d_totals) (void)hipFree(sc->d_totals);
  if (sc->h_totals) (void)hipHostFree(sc->h_totals);
  if (sc->h_v2) (void)hipHostFree(sc->h_v2);
  sc->d_totals = nullptr;
  sc->h_totals = nullptr;
  sc->h_v2 = nullptr;
  sc->h_v2_dev = nullptr;
  if (sc->ev_ready)
    for (auto &e : sc->ev) (void)hipEventDestroy(e);
  sc->ev_ready = false;
  for (auto &st : sc->hs) {
    if (st) (void)hipStreamDestroy(st);
    st = nullptr;
  }
}

uint64_t scratch_bytes(const Scratch *sc) {
  uint64_t n = 0;
  for (auto &b : sc->v2buf) n += b.bytes;
  for (auto &b : sc->hostbuf) n += b.bytes;
  n += sc->cap_chunks * 8 + sc->cap_blocks * 16 + sc->cap_docs * 8;
  return n;
}

// adds the passes that were thrown away to the timing the last pass published (profiling on)
void note_repeats(aha_ac *ac, uint32_t repeats) {
  if (!ac->profiling.load()) return;
  std::lock_guard<std::mutex> lk(ac->last_mu);
  ac->last.repeats = repeats;
}

void publish_timing(aha_ac *ac, const aha_timing &t) {
  std::lock_guard<std::mutex> lk(ac->last_mu);
  ac->last = t;
}

int32_t upload_image(aha_ac *ac, const Image &img) {
  const Automaton &a = ac->aut;
  DevAut &d = ac->dev;
  d.root = img.root_base;
  d.n_slots = img.n_slots;
  d.max_len = a.max_key_len;
  d.compact = img.compact ? 1u : 0u;
  d.s1_lo = ac->s1_lo;
  d.s2_lo = ac->s2_lo;
  d.s2_hi = ac->s2_hi;
  int32_t rc;
  // flattened output chains (image.hpp): total length = sum of key_cnt; they need 24-bit offsets
  uint64_t total = 0;
  for (uint32_t k = 0; k < a.n_keys; k++) total += a.key_cnt[k];
  d.key_info = nullptr;
  d.chain = nullptr;
  d.chain_chars = nullptr;
  std::vector<uint32_t> kinfo;
  if (a.n_keys && total < (1ull << 24)) {
    std::vector<uint2> kc;  // the same chains for char offsets: {characters of the key, key}
    std::vector<uint2> &ch = ac->chain_host;
    ch.clear();
    kinfo.resize(a.n_keys);
    ch.reserve((size_t)total);
    kc.reserve((size_t)total);
    for (uint32_t k = 0; k < a.n_keys; k++) {
      kinfo[k] = (uint32_t)ch.size() | (std::min<uint32_t>(a.key_cnt[k], 255u) << 24);
      for (int32_t j = (int32_t)k; j >= 0; j = a.key_next[j]) {
        ch.push_back(uint2{a.key_len[j], (uint32_t)j});
        kc.push_back(uint2{a.key_kc[j] + 1u, (uint32_t)j});
      }
    }
    ac->key_info = kinfo;
    if ((rc = upload(ac, kinfo, &d.key_info))) return rc;
    if ((rc = upload(ac, ch, &d.chain))) return rc;
    if ((rc = upload(ac, kc, &d.chain_chars))) return rc;
  }
  if (img.compact) {
    const uint32_t *p = nullptr;
    if ((rc = upload(ac, img.narrow, &p))) return rc;
    d.slots = p;
    if ((rc = upload(ac, img.end_key, &d.end_key))) return rc;
    std::vector<uint32_t> info(img.end_key.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < info.size(); i++) {
      const int32_t k = img.end_key[i];
      if (k >= 0) info[i] = kinfo.empty() ? ((uint32_t)k | (std::min<uint32_t>(a.key_cnt[k], 255u) << 24)) : kinfo[k];
    }
    if ((rc = upload(ac, info, &d.end_info))) return rc;
  } else {
    const uint64_t *p = nullptr;
    if ((rc = upload(ac, img.wide, &p))) return rc;
    d.slots = p;
    d.end_key = nullptr;
    d.end_info = nullptr;
  }
  std::vector<uint2> ln(a.n_keys);
  for (uint32_t k = 0; k < a.n_keys; k++) ln[k] = uint2{a.key_len[k], (uint32_t)a.key_next[k]};
  if ((rc = upload(ac, ln, &d.key_ln))) return rc;
  if ((rc = upload(ac, a.key_cnt, &d.key_cnt))) return rc;
  if ((rc = upload(ac, a.key_kc, &d.key_kc))) return rc;
  return AHA_OK;
}

int32_t ensure_scratch(aha_ac *ac, Scratch *sc, uint64_t n_chunks, uint64_t n_blocks, uint64_t n_docs) {
  if (n_chunks > sc->cap_chunks) {
    if (sc->d_counts) (void)hipFree(sc->d_counts);
    if (sc->d_leads) (void)hipFree(sc->d_leads);
    sc->d_counts = sc->d_leads = nullptr;
    sc->cap_chunks = 0;
    uint64_t n = n_chunks + n_chunks / 8 + 1024;
    HIPCHK(ac, hipMalloc((void **)&sc->d_counts, n * sizeof(uint32_t)));
    HIPCHK(ac, hipMalloc((void **)&sc->d_leads, n * sizeof(uint32_t)));
    sc->cap_chunks = n;
  }
  if (n_blocks > sc->cap_blocks) {
    if (sc->d_blk_hits) (void)hipFree(sc->d_blk_hits);
    if (sc->d_blk_leads) (void)hipFree(sc->d_blk_leads);
    sc->d_blk_hits = sc->d_blk_leads = nullptr;
    sc->cap_blocks = 0;
    uint64_t n = n_blocks + n_blocks / 8 + 64;
    HIPCHK(ac, hipMalloc((void **)&sc->d_blk_hits, n * sizeof(uint64_t)));
    HIPCHK(ac, hipMalloc((void **)&sc->d_blk_leads, n * sizeof(uint64_t)));
    sc->cap_blocks = n;
  }
  if (n_docs + 1 > sc->cap_docs) {
    if (sc->d_docg) (void)hipFree(sc->d_docg);
    sc->d_docg = nullptr;
    sc->cap_docs = 0;
    uint64_t n = n_docs + 1 + n_docs / 8 + 64;
    HIPCHK(ac, hipMalloc((void **)&sc->d_docg, n * sizeof(uint64_t)));
    sc->cap_docs = n;
  }
  if (!sc->d_totals) {
    HIPCHK(ac, hipMalloc((void **)&sc->d_totals, 2 * sizeof(uint64_t)));
    HIPCHK(ac, hipHostMalloc((void **)&sc->h_totals, 2 * sizeof(uint64_t), hipHostMallocDefault));
  }
  return AHA_OK;
}

int32_t fill_params(aha_ac *ac, const aha_match_params *p, MatchArgs &M, int *longest) {
  M.chars = 0;
  M.sep = 0;
  *longest = 0;
  memset(M.sep_block, 0, sizeof(M.sep_block));
  if (!p) return AHA_OK;
  if (p->struct_size >= offsetof(aha_match_params, longest) + sizeof(int32_t)) {
    if (p->longest < 0 || p->longest > 2 || (p->longest && p->sep_size > 0)) {
      tls_err = "match_longest: intersectable is 1 or 2, and there is no separator overload";
      return AHA_E_INVALID;
    }
    *longest = p->longest;
  }
  M.chars = p->char_offsets ? 1 : 0;
  if (p->sep_size > 256) {  // raise "sep BitArray size > 256 is not supported" ac.cr:322
    tls_err = "sep BitArray size > 256 is not supported";
    return AHA_E_SEP_SIZE;
  }
  if (p->sep_size > 0) {
    M.sep = 1;
    // blocked(c) <=> c < sep.size && !sep[c]   (ac.cr:326, 333)
    for (int c = 0; c < p->sep_size; c++)
      if (!((p->sep_bits[c >> 3] >> (c & 7)) & 1)) M.sep_block[c >> 5] |= 1u << (c & 31);
  }
  return AHA_OK;
}

// match_longest asks is_end? like the reference does: "ends a key" OR one of Cedar's stale END flags
// (cedar.cr:642-648, observable at ac.cr:126-128).  The set is derived once per handle, on first use, by replaying
// Cedar's inserts (cedar_replay.cpp); the bitmap (one bit per slot, set at the base of a stale state) lives in HBM
// beside the image and only the match_longest kernels read it.
int32_t ensure_stale(aha_ac *ac) {
  std::call_once(ac->stale_once, [ac]() {
   try {
    cedar_stale_ends(ac->aut, ac->stale_states);
    ac->dev_longest = ac->dev;
    ac->dev_longest.stale_bits = nullptr;
    ac->dev_longest.term_bits = nullptr;
    if (ac->device < 0) return;
    DeviceGuard g(ac->device);
    // states whose Cedar node keeps its value in a label-0 child (they end a key and have children): what a NUL byte
    // of the text reaches there (kernels.hip, kValueNode)
    const Automaton &a = ac->aut;
    std::vector<uint32_t> term(((size_t)ac->n_slots + 31) / 32, 0u);
    bool any_term = false;
    for (uint32_t s = 1; s < a.n_states; s++)
      if (a.key_of[s] >= 0 && a.n_child[s] > 0) {
        term[ac->state_base[s] >> 5] |= 1u << (ac->state_base[s] & 31);
        any_term = true;
      }
    if (any_term && (ac->stale_rc = upload_late(ac, term, &ac->dev_longest.term_bits))) return;
    if (ac->stale_states.empty()) return;
    std::vector<uint32_t> bits(((size_t)ac->n_slots + 31) / 32, 0u);
    for (uint32_t s : ac->stale_states) bits[ac->state_base[s] >> 5] |= 1u << (ac->state_base[s] & 31);
    ac->stale_rc = upload_late(ac, bits, &ac->dev_longest.stale_bits);
   } catch (...) {  // bad_alloc of the replay or of a bitmap: no exception crosses the C boundary
    ac->stale_rc = AHA_E_NOMEM;
   }
  });
  return ac->stale_rc;
}


}  // namespace ahai


extern "C" {

const char *aha_strerror(int32_t code) {
  switch (code) {
    case AHA_OK: return "ok";
    case AHA_E_INVALID: return "invalid argument";
    case AHA_E_EMPTY_KEY: return "Cannot insert empty key";
    case AHA_E_ZERO_BYTE: return "key[pos] is zero";
    case AHA_E_DUP_KEY: return "key appear twice.";
    case AHA_E_SEP_SIZE: return "sep BitArray size > 256 is not supported";
    case AHA_E_CAPACITY: return "output buffer too small";
    case AHA_E_NO_DEVICE: return "no usable HIP device (libaha_hip has no CPU fallback)";
    case AHA_E_HIP: return "HIP runtime error";
    case AHA_E_TOO_LONG: return "sequence longer than Int32 offsets allow";
    case AHA_E_NOT_FOUND: return "not found";
    case AHA_E_TOO_LARGE: return "automaton too large for the device image";
    case AHA_E_NOMEM: return "out of host memory";
  }
  return "unknown error";
}

const char *aha_last_error(const aha_ac *ac) {
  (void)ac;
  return tls_err.c_str();
}
uint32_t aha_abi_version(void) { return AHA_ABI_VERSION; }

int32_t aha_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// uploads the host image of `ac` to `device` (< 0: the current one) and sets the engines up there
static int32_t attach_device(aha_ac *ac, int device) {
  const int n = aha_device_count();
  if (n <= 0) return AHA_E_NO_DEVICE;
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= n) return AHA_E_INVALID;
  ac->device = device;
  DeviceGuard g(device);
  const int32_t rc = upload_image(ac, ac->img);
  if (rc != AHA_OK) {
    fprintf(stderr, "aha_ac_compile: %s\n", tls_err.c_str());
    return rc;
  }
  v2_setup(ac);
  return AHA_OK;
}

int32_t aha_ac_compile(const uint8_t *key_bytes, const uint64_t *key_offsets, uint32_t n_keys,
                       const aha_options *opts, aha_ac **out, uint32_t *err_key) {
  if (!out || !key_offsets || (n_keys && !key_bytes && key_offsets[n_keys] != key_offsets[0]))
    return AHA_E_INVALID;
  *out = nullptr;
  uint32_t flags = opts ? opts->flags : 0;
  int device = opts ? opts->device : -1;
  aha_ac *ac = new aha_ac();
  BuildError be;
  static const uint8_t dummy = 0;
  if (!build_automaton(key_bytes ? key_bytes : &dummy, key_offsets, n_keys, ac->aut, be)) {
    if (err_key) *err_key = be.key_index;
    delete ac;
    return be.code;
  }
  Placement pl;
  Image &img = ac->img;
  // Shadow fail links (default): fail links of depth <= 2 targets are recomputed from the last two input bytes, so
  // only the root and the deep-fail states keep a header slot (automaton.hpp, Placement::headerless).
  // AHA_SHADOW_FAIL=0 keeps a header for every state.
  const char *sf = getenv("AHA_SHADOW_FAIL");
  bool shadow = !(sf && strcmp(sf, "0") == 0);
  // (until round 4 a small automaton that fits LDS kept a header for every state and ran a trip of its own; the shadow
  // fail links take cfg 2 from 1.55 to 1.02 trips per byte, so it gets the same image and trip as everyone else)
  for (;;) {
    place_states(ac->aut, pl, shadow, shadow);
    if (!encode_image(ac->aut, pl, (flags & AHA_OPT_FORCE_WIDE) != 0, img)) {
      delete ac;
      return AHA_E_TOO_LARGE;
    }
    ac->n_slots = img.n_slots;
    ac->compact = img.compact;
    ac->slot_bytes = img.compact ? 4 : 8;
    plan_engine(ac, pl);
    // the traversal probes the depth-1 rows in LDS to keep the shadow state: they must all be resident
    if (!shadow || pl.seg_start[2] <= ac->v2_lds_slots) break;
    shadow = false;
  }
  ac->seg2 = pl.seg_start[2];
  ac->state_base = pl.base;
  ac->s1_lo = shadow ? pl.seg_start[2] : 0;
  ac->s2_lo = shadow ? pl.seg_start[3] : 0;
  ac->s2_hi = shadow ? pl.deep_fail_start : 0;
  {
    // character-level image: built when the keys are UTF-8-shaped and at least 30 % of their bytes lie in multi-byte
    // characters (AHA_ENGINE=unit forces it for every eligible key set, AHA_ENGINE=v2 / v1 never build it)
    const char *eng = getenv("AHA_ENGINE");
    if (!eng || strcmp(eng, "unit") == 0 || strcmp(eng, "skip") == 0 || strcmp(eng, "pair") == 0) build_unit(ac->aut, ac->unit, eng != nullptr);
    if (getenv("AHA_DEBUG") && !ac->unit.ok) fprintf(stderr, "aha: no character-level image: %s\n", ac->unit.why);
    // prefix filter (scan_filter.hip): the keys' first D = min(4, shortest key) bytes in a blocked Bloom filter of 64 KiB.
    // For key sets of at least 3-byte keys (a shorter prefix passes too much text), none beyond 64 bytes (a walk's reach
    // into the next chunk), compact images, and a filter at most a quarter full.  AHA_ENGINE=filter / unset: built;
    // v2 / v1 / unit: not.
    if ((!eng || strcmp(eng, "filter") == 0) && !ac->unit.ok && ac->img.compact && ac->aut.n_keys > 0 && ac->aut.max_key_len <= 64) {
      uint32_t minlen = ~0u;
      for (uint32_t k = 0; k < ac->aut.n_keys; k++) minlen = std::min(minlen, ac->aut.key_len[k]);
      // keys nested in one another along one trie path: a walk of kf_walk keeps four END steps (scan_filter.hip kfMaxEnds)
      // and hands the whole call back when it meets a fifth -- after the filter and part of the walks ran.  The deepest
      // nesting is a property of the key set: with more than four keys on one root-to-leaf path there is no filter.
      uint32_t nest = 0;
      {
        const Automaton &a = ac->aut;
        std::vector<uint8_t> ends(a.n_states, 0);  // BFS numbering: a parent comes before its children
        for (uint32_t st = 0; st < a.n_states; st++)
          for (uint32_t j = 0; j < a.n_child[st]; j++) {
            const uint32_t c = a.first_child[st] + j;
            ends[c] = (uint8_t)std::min<uint32_t>(ends[st] + (a.key_of[c] >= 0 ? 1u : 0u), 255u);
            nest = std::max<uint32_t>(nest, ends[c]);
          }
      }
      if (nest > 4 && getenv("AHA_DEBUG")) fprintf(stderr, "aha: no prefix filter: %u keys nested on one trie path\n", nest);
      if (minlen >= 3 && nest <= 4) {
        const uint32_t D = std::min(4u, minlen);
        // the smallest filter (2^10 .. 2^kFilterLog2 words) that stays under 1/256 full (false candidates: about the square of
        // the fill; the filter's size does not show in kf_filter's time, its false candidates show in kf_walk's), or the largest
        // while it is no more than a quarter full
        const char *fd = getenv("AHA_FILTER_FILL");  // (lab: the fill the size search stops at, as a denominator)
        const uint64_t fill_den = fd && atoi(fd) >= 4 ? (uint64_t)atoi(fd) : 256;
        for (uint32_t lg = 10; lg <= kFilterLog2; lg++) {
          ac->pf_bloom.assign((size_t)1 << lg, 0u);
          for (uint32_t k = 0; k < ac->aut.n_keys; k++) {
            uint32_t w = 0;
            for (uint32_t j = 0; j < D; j++) w |= (uint32_t)ac->aut.blob[ac->aut.offs[k] + j] << (8 * j);
            const uint32_t h = w * kFilterMul;  // (kf_filter: the word from the product's top lg bits, two bits from the ten below)
            ac->pf_bloom[h >> (32 - lg)] |= (1u << ((h >> (32 - lg - 5)) & 31u)) | (1u << ((h >> (32 - lg - 10)) & 31u));
          }
          uint64_t bits = 0;
          for (uint32_t x : ac->pf_bloom) bits += (uint64_t)__builtin_popcount(x);
          if (bits * (lg < kFilterLog2 ? fill_den : 4) <= ((uint64_t)32 << lg)) {
            ac->pf_d = D;
            ac->pf_log2 = lg;
            break;
          }
        }
        if (!ac->pf_d) {
          std::vector<uint32_t>().swap(ac->pf_bloom);
          if (getenv("AHA_DEBUG")) fprintf(stderr, "aha: no prefix filter: it would be more than a quarter full\n");
        }
      }
    }
  }
  if (!(flags & AHA_OPT_HOST_ONLY)) {
    const int32_t rc = attach_device(ac, device);
    if (rc != AHA_OK) {
      aha_ac_free(ac);
      return rc;
    }
  }
  *out = ac;
  return AHA_OK;
}

// A second handle for the same keys on another device: the host side of `src` (automaton, both images, key tables) is
// copied, nothing is compiled again -- compile once, upload per device (SURVEY.md section 8 e; aha_group_compile).
int32_t aha_ac_replicate(const aha_ac *src, int32_t device, aha_ac **out) {
  if (!src || !out) return AHA_E_INVALID;
  *out = nullptr;
  aha_ac *ac = nullptr;
  try {
    ac = new aha_ac();
    ac->aut = src->aut;
    ac->img = src->img;
    ac->n_slots = src->n_slots;
    ac->slot_bytes = src->slot_bytes;
    ac->compact = src->compact;
    ac->chunk = src->chunk;
    ac->v2_lds_slots = src->v2_lds_slots;
    ac->v2_bpc = src->v2_bpc;
    ac->s1_lo = src->s1_lo;
    ac->s2_lo = src->s2_lo;
    ac->s2_hi = src->s2_hi;
    ac->unit = src->unit;
    ac->pf_bloom = src->pf_bloom;
    ac->pf_d = src->pf_d;
    ac->pf_cus = src->pf_cus;
    ac->pf_log2 = src->pf_log2;
    ac->seg2 = src->seg2;
    ac->state_base = src->state_base;
  } catch (...) {
    delete ac;
    return AHA_E_NOMEM;
  }
  const int32_t rc = attach_device(ac, device);
  if (rc != AHA_OK) {
    aha_ac_free(ac);
    return rc;
  }
  *out = ac;
  return AHA_OK;
}

void aha_ac_free(aha_ac *ac) {
  if (!ac) return;
  if (ac->device >= 0) {
    DeviceGuard g(ac->device);
    for (void *p : ac->dev_allocs) (void)hipFree(p);
    for (auto &u : ac->pool) free_scratch(u.get(), true);
  }
  delete ac;
}

int32_t aha_ac_hits_pack_device(aha_ac *ac, const aha_hit *d_hits, uint64_t n, int32_t *d_pairs, void *stream) {
  if (!ac || ac->device < 0 || (n && (!d_hits || !d_pairs))) return AHA_E_INVALID;
  DeviceGuard g(ac->device);
  launch_hits_pack(reinterpret_cast<const int32_t *>(d_hits), n, d_pairs, stream);
  HIPCHK(ac, hipGetLastError());
  return AHA_OK;
}

int32_t aha_ac_hits_unpack_device(aha_ac *ac, const int32_t *d_pairs, uint64_t n, int32_t char_offsets,
                                  aha_hit *d_hits, void *stream) {
  if (!ac || ac->device < 0 || (n && (!d_hits || !d_pairs))) return AHA_E_INVALID;
  DeviceGuard g(ac->device);
  launch_hits_unpack(ac->dev, d_pairs, n, char_offsets ? 1 : 0, reinterpret_cast<int32_t *>(d_hits), stream);
  HIPCHK(ac, hipGetLastError());
  return AHA_OK;
}

int32_t aha_ac_stream_format(const aha_ac *ac, uint32_t *step_bits, uint32_t *len_bits) {
  if (!ac || !step_bits || !len_bits) return AHA_E_INVALID;
  const StreamFmt f = stream_fmt(ac);
  *step_bits = f.step_bits;
  *len_bits = f.len_bits;
  return AHA_OK;
}

int32_t aha_ac_hits_pack4_device(aha_ac *ac, const aha_hit *d_hits, uint64_t n, uint32_t *d_words, uint64_t cap_words,
                                 uint64_t *d_n_words, void *stream) {
  if (!ac || ac->device < 0 || !d_words || !d_n_words || (n && !d_hits)) return AHA_E_INVALID;
  if (stream_fmt(ac).len_bits == 0 && ac->aut.n_keys > (1u << 20)) {
    tls_err = "the 4-byte exchange stream holds key ids below 2^20: use the {end, value} pairs";
    return AHA_E_INVALID;
  }
  if (n >= (1ull << 41)) return AHA_E_INVALID;
  if (cap_words < 2 * n + (n + 1023) / 1024) {
    tls_err = "4-byte exchange stream: capacity below 2 n + ceil(n / 1024) words";
    return AHA_E_CAPACITY;
  }
  DeviceGuard g(ac->device);
  launch_hits_pack4(reinterpret_cast<const int32_t *>(d_hits), n, d_words,
                    reinterpret_cast<unsigned long long *>(d_n_words), stream_fmt(ac), stream);
  HIPCHK(ac, hipGetLastError());
  return AHA_OK;
}

int32_t aha_ac_hits_unpack4_device(aha_ac *ac, const uint32_t *d_words, uint64_t n, int32_t char_offsets,
                                   aha_hit *d_hits, void *stream) {
  if (!ac || ac->device < 0 || (n && (!d_hits || !d_words))) return AHA_E_INVALID;
  DeviceGuard g(ac->device);
  launch_hits_unpack4(ac->dev, d_words, n, char_offsets ? 1 : 0, reinterpret_cast<int32_t *>(d_hits), stream_fmt(ac), stream);
  HIPCHK(ac, hipGetLastError());
  return AHA_OK;
}

int32_t aha_ac_hits_unpack4_segs_device(aha_ac *ac, const uint32_t *d_words, const aha_stream_seg *segs, uint32_t n_segs,
                                        int32_t char_offsets, aha_hit *d_hits, void *stream) {
  if (!ac || ac->device < 0 || n_segs > kMaxSegs || (n_segs && !segs)) return AHA_E_INVALID;
  uint64_t woff[kMaxSegs], nh[kMaxSegs], ooff[kMaxSegs];
  bool any = false;
  for (uint32_t k = 0; k < n_segs; k++) {
    woff[k] = segs[k].word_offset;
    nh[k] = segs[k].n_hits;
    ooff[k] = segs[k].out_offset;
    any = any || nh[k] != 0;
  }
  if (any && (!d_words || !d_hits)) return AHA_E_INVALID;
  DeviceGuard g(ac->device);
  launch_hits_unpack4_segs(ac->dev, d_words, woff, nh, ooff, n_segs, char_offsets ? 1 : 0,
                           reinterpret_cast<int32_t *>(d_hits), stream_fmt(ac), stream);
  HIPCHK(ac, hipGetLastError());
  return AHA_OK;
}

// ---- save / load: the library's own container (see include/aha_hip.h) ----------
namespace {
constexpr char kSaveMagic[8] = {'A', 'H', 'A', 'H', 'I', 'P', '0', '1'};
inline uint64_t fnv1a(const uint8_t *p, uint64_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint64_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}
}  // namespace

int64_t aha_ac_save(const aha_ac *ac, void *buf, uint64_t cap_bytes) {
  if (!ac) return AHA_E_INVALID;
  const uint32_t K = ac->aut.n_keys;
  const uint64_t blob_bytes = ac->aut.blob.size();
  const uint64_t need = 8 + 4 + 4 + 8 + 8ull * (K + 1) + blob_bytes + 8;
  if (!buf || cap_bytes < need) return (int64_t)need;
  uint8_t *w = static_cast<uint8_t *>(buf);
  uint64_t o = 0;
  auto put = [&](const void *src, uint64_t n) {
    if (n) memcpy(w + o, src, n);
    o += n;
  };
  const uint32_t format = 1;
  put(kSaveMagic, 8);
  put(&format, 4);
  put(&K, 4);
  put(&blob_bytes, 8);
  put(ac->aut.offs.data(), 8ull * (K + 1));
  put(ac->aut.blob.data(), blob_bytes);
  const uint64_t h = fnv1a(w, o);
  put(&h, 8);
  return (int64_t)o;
}

int32_t aha_ac_load(const void *buf, uint64_t n_bytes, const aha_options *opts, aha_ac **out) {
  if (!buf || !out) return AHA_E_INVALID;
  *out = nullptr;
  const uint8_t *r = static_cast<const uint8_t *>(buf);
  if (n_bytes < 8 + 4 + 4 + 8 + 8 + 8 || memcmp(r, kSaveMagic, 8) != 0) return AHA_E_INVALID;
  uint32_t format, K;
  uint64_t blob_bytes;
  memcpy(&format, r + 8, 4);
  memcpy(&K, r + 12, 4);
  memcpy(&blob_bytes, r + 16, 8);
  if (format != 1) return AHA_E_INVALID;
  const uint64_t head = 24, offs_bytes = 8ull * ((uint64_t)K + 1);
  if (blob_bytes > n_bytes || offs_bytes > n_bytes || head + offs_bytes + blob_bytes + 8 != n_bytes)
    return AHA_E_INVALID;
  uint64_t h;
  memcpy(&h, r + n_bytes - 8, 8);
  if (h != fnv1a(r, n_bytes - 8)) return AHA_E_INVALID;
  std::vector<uint64_t> offs((size_t)K + 1);
  memcpy(offs.data(), r + head, offs_bytes);
  if (offs[0] != 0 || offs[K] != blob_bytes) return AHA_E_INVALID;
  for (uint32_t k = 0; k < K; k++)
    if (offs[k + 1] < offs[k]) return AHA_E_INVALID;
  return aha_ac_compile(r + head + offs_bytes, offs.data(), K, opts, out, nullptr);
}

// Output structs grow with the ABI: the caller says how many bytes its struct has (struct_size, set before the call) and gets
// no more than that; 0 -- a caller built before the field was read -- means the size the struct had then (ABI 5).
// Output structs that grew with the ABI are filled up to the caller's struct_size -- when that is a size the struct has had
// (sizes[]: ascending, 0-terminated; the first is the size of ABI 5, when the structs were output-only and nobody set the
// field).  Anything else -- 0, or the stack garbage of a caller built against the old header -- gets the ABI-5 size: the
// library never writes more than the oldest struct holds unless the caller says exactly which newer one it has (a caller
// built against a LATER header gets the ABI-5 part and reads from struct_size how much was filled).
static void copy_sized(void *dst, const void *full, uint32_t caller_size, const size_t *sizes) {
  size_t n = sizes[0];
  for (int i = 0; sizes[i]; i++)
    if (caller_size == sizes[i]) n = sizes[i];
  if (n < 4) return;
  memcpy(dst, full, n);
  const uint32_t filled = (uint32_t)n;
  memcpy(dst, &filled, 4);  // struct_size: the bytes that were filled
}

int32_t aha_ac_info(const aha_ac *ac, aha_ac_info_t *caller_info) {
  if (!ac || !caller_info) return AHA_E_INVALID;
  aha_ac_info_t full, *info = &full;
  memset(info, 0, sizeof(*info));
  info->struct_size = sizeof(*info);
  info->n_keys = ac->aut.n_keys;
  info->n_states = ac->aut.n_states;
  info->n_slots = ac->n_slots;
  info->image_bytes = ac->image_bytes;
  info->max_key_len = ac->aut.max_key_len;
  info->slot_bytes = ac->slot_bytes;
  info->lds_slots = ac->v2_lds_slots;
  info->device = ac->device;
  info->fail_s1_lo = ac->s1_lo;
  info->fail_s2_lo = ac->s2_lo;
  info->fail_hdr_lo = ac->s2_hi;
  // (a device handle: the image is uploaded and the kernel's LDS fits -- calls will run it; a host-only handle: it is built)
  info->unit_enabled = (ac->device < 0 ? ac->unit.ok : ac->unit_ok) ? 1u : 0u;
  info->unit_slots = ac->unit.n_slots;
  info->unit_syms = ac->unit.n_syms;
  info->unit_multi_permille = ac->unit.multi_permille;
  info->unit_big_lo = ac->unit.n_shared;
  info->unit_big_block = ac->unit.big_block;
  info->unit_n_low = ac->unit.n_low;
  info->unit_n_big = ac->unit.n_big;
  info->unit_base_bits = ac->unit.ok ? ac->unit.base_bits : 0;
  info->unit_headers = ac->unit.ok ? ac->unit.n_nfr : 0;
  info->unit_header_beside = ac->unit_ok ? ac->udev.hdr_beside : 0;
  const bool pf = ac->device < 0 ? ac->pf_d != 0 : ac->pf_ok;
  info->filter_prefix_bytes = pf ? ac->pf_d : 0;
  info->filter_words = pf ? 1u << ac->pf_log2 : 0;
  const bool sk = ac->device < 0 ? skip_eligible(ac) : ac->skip_ok;
  const bool pr = ac->device < 0 ? pair_eligible(ac) : ac->pair_ok;
  info->skip_filter_words = (sk || pr) ? 1u << ac->unit.mark_log2 : 0;
  info->skip_pairs = (sk || pr) ? ac->unit.n_pairs : 0;
  info->pair_hash_k1 = (sk || pr) ? ac->unit.pair_k1 : 0;
  info->pair_table_log2 = pr ? ac->unit.pair_log2 : 0;
  info->pair_groups = pr ? ac->unit.pair_groups : 0;
  info->pair_engine = pr ? 1u : 0u;
  static const size_t kInfoSizes[] = {offsetof(aha_ac_info_t, unit_big_lo) /* ABI 5 */, offsetof(aha_ac_info_t, filter_prefix_bytes) /* 6 */,
                                      offsetof(aha_ac_info_t, skip_filter_words) /* 7 */, sizeof(aha_ac_info_t), 0};
  copy_sized(caller_info, &full, caller_info->struct_size, kInfoSizes);
  return AHA_OK;
}

int32_t aha_ac_key(const aha_ac *ac, int32_t id, uint8_t *buf, int32_t cap) {
  if (!ac) return AHA_E_INVALID;
  if (id < 0 || (uint32_t)id >= ac->aut.n_keys) return AHA_E_NOT_FOUND;
  uint64_t o = ac->aut.offs[id], n = ac->aut.offs[id + 1] - o;
  if (buf && cap > 0) memcpy(buf, ac->aut.blob.data() + o, std::min<uint64_t>(n, (uint64_t)cap));
  return (int32_t)n;
}

int32_t aha_ac_id(const aha_ac *ac, const uint8_t *key, int32_t len) {
  if (!ac || (!key && len > 0)) return AHA_E_INVALID;
  int32_t k = ac->aut.find_key(key, len);
  return k < 0 ? AHA_E_NOT_FOUND : k;
}

int64_t aha_ac_export(const aha_ac *ac, int32_t which, void *buf, uint64_t cap_bytes) {
  if (!ac) return AHA_E_INVALID;
  const void *src = nullptr;
  uint64_t bytes = 0;
  std::vector<uint32_t> tmp;
  const Automaton &a = ac->aut;
  switch (which) {
    case AHA_IMG_SLOTS:
      if (ac->img.compact) {
        src = ac->img.narrow.data();
        bytes = ac->img.narrow.size() * 4;
      } else {
        src = ac->img.wide.data();
        bytes = ac->img.wide.size() * 8;
      }
      break;
    case AHA_IMG_END_KEY:
      src = ac->img.end_key.data();
      bytes = ac->img.end_key.size() * 4;
      break;
    case AHA_IMG_KEY_LN:
      tmp.resize((size_t)a.n_keys * 2);
      for (uint32_t k = 0; k < a.n_keys; k++) {
        tmp[2 * k] = a.key_len[k];
        tmp[2 * k + 1] = (uint32_t)a.key_next[k];
      }
      src = tmp.data();
      bytes = tmp.size() * 4;
      break;
    case AHA_IMG_KEY_CNT:
      src = a.key_cnt.data();
      bytes = a.key_cnt.size() * 4;
      break;
    case AHA_IMG_KEY_KC:
      src = a.key_kc.data();
      bytes = a.key_kc.size() * 4;
      break;
    case AHA_IMG_UNIT_SLOTS:
      src = ac->unit.slots.data();
      bytes = ac->unit.slots.size() * 8;
      break;
    case AHA_IMG_UNIT_ROOT:
      src = ac->unit.root.data();
      bytes = ac->unit.root.size() * 4;
      break;
    case AHA_IMG_UNIT_END_KEY:
      src = ac->unit.end_key.data();
      bytes = ac->unit.end_key.size() * 4;
      break;
    case AHA_IMG_UNIT_TABLES:
      src = ac->unit.tables.data();
      bytes = ac->unit.tables.size() * 4;
      break;
    case AHA_IMG_UNIT_MARKS:
      src = ac->unit.mark_bloom.data();
      bytes = (skip_eligible(ac) || pair_eligible(ac)) ? ac->unit.mark_bloom.size() * 4 : 0;
      break;
    case AHA_IMG_UNIT_PAIRS:
      src = ac->unit.pair_tab.data();
      bytes = pair_eligible(ac) ? ac->unit.pair_tab.size() * 4 : 0;
      break;
    case AHA_IMG_UNIT_PAIR_DISP:
      src = ac->unit.pair_disp.data();
      bytes = pair_eligible(ac) ? ac->unit.pair_disp.size() : 0;
      break;
    case AHA_IMG_STALE_ENDS: {
      // {key id, prefix length} of every state with a stale END flag: the state is that prefix of that key
      aha_ac *m = const_cast<aha_ac *>(ac);
      if (ensure_stale(m) != AHA_OK) return AHA_E_HIP;
      std::vector<uint8_t> is_stale(a.n_states, 0);
      for (uint32_t st : ac->stale_states) is_stale[st] = 1;
      size_t left = ac->stale_states.size();
      for (uint32_t k = 0; k < a.n_keys && left; k++) {
        uint32_t st = 0;
        for (uint64_t i = a.offs[k]; i < a.offs[k + 1]; i++) {
          st = a.child(st, a.blob[i]);
          if (is_stale[st]) {
            is_stale[st] = 0;
            left--;
            tmp.push_back(k);
            tmp.push_back((uint32_t)(i - a.offs[k] + 1));
          }
        }
      }
      src = tmp.data();
      bytes = tmp.size() * 4;
      break;
    }
    default:
      return AHA_E_INVALID;
  }
  if (buf && cap_bytes >= bytes && bytes) memcpy(buf, src, bytes);
  return (int64_t)bytes;
}

int32_t aha_ac_release_scratch(aha_ac *ac) {
  if (!ac) return AHA_E_INVALID;
  if (ac->device < 0) return AHA_OK;
  std::lock_guard<std::mutex> lk(ac->pool_mu);
  DeviceGuard g(ac->device);
  for (auto &u : ac->pool) {
    std::lock_guard<std::mutex> lk2(u->mu);  // waits for a call that still uses the set
    free_scratch(u.get(), false);
  }
  return AHA_OK;
}

int64_t aha_ac_scratch_bytes(aha_ac *ac) {
  if (!ac) return AHA_E_INVALID;
  std::lock_guard<std::mutex> lk(ac->pool_mu);
  uint64_t n = 0;
  for (auto &u : ac->pool) {
    std::lock_guard<std::mutex> lk2(u->mu);
    n += scratch_bytes(u.get());
  }
  return (int64_t)n;
}

int32_t aha_ac_set_profiling(aha_ac *ac, int32_t enabled) {
  if (!ac) return AHA_E_INVALID;
  if (ac->device < 0) return AHA_E_NO_DEVICE;
  ac->profiling.store(enabled != 0);  // a scratch set creates its events when a profiled call first leases it
  return AHA_OK;
}

int32_t aha_ac_last_timing(const aha_ac *ac, aha_timing *t) {
  if (!ac || !t) return AHA_E_INVALID;
  aha_timing full;
  {
    std::lock_guard<std::mutex> lk(const_cast<aha_ac *>(ac)->last_mu);
    full = ac->last;
  }
  full.struct_size = sizeof(full);
  static const size_t kTimingSizes[] = {offsetof(aha_timing, repeats) /* ABI 5 */, sizeof(aha_timing), 0};
  copy_sized(t, &full, t->struct_size, kTimingSizes);
  return AHA_OK;
}

static int32_t match_batch_device_impl(aha_ac *ac, Scratch *sc, const uint8_t *d_corpus, const uint64_t *d_doc_offsets,
                                       uint64_t n_docs, uint64_t n_bytes, const aha_match_params *params,
                                       aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets, uint64_t *n_hits,
                                       void *stream, bool offsets_checked, const PackOut *pk = nullptr);

int32_t aha_ac_match_batch_device(aha_ac *ac, const uint8_t *d_corpus,
                                  const uint64_t *d_doc_offsets, uint64_t n_docs,
                                  uint64_t n_bytes, const aha_match_params *params,
                                  aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets,
                                  uint64_t *n_hits, void *stream) {
  if (!ac || !n_hits || !d_doc_offsets) return AHA_E_INVALID;
  if (ac->device < 0) {
    tls_err = aha_strerror(AHA_E_NO_DEVICE);
    return AHA_E_NO_DEVICE;
  }
  Lease lease(ac);
  return match_batch_device_impl(ac, lease.get(), d_corpus, d_doc_offsets, n_docs, n_bytes, params, d_out, cap,
                                 d_doc_hit_offsets, n_hits, stream, false);
}

static int32_t match_batch_device_impl(aha_ac *ac, Scratch *sc, const uint8_t *d_corpus, const uint64_t *d_doc_offsets,
                                       uint64_t n_docs, uint64_t n_bytes, const aha_match_params *params,
                                       aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets, uint64_t *n_hits,
                                       void *stream, bool offsets_checked, const PackOut *pk) {
  if (pk) {
    if (!ac || !pk->d_words || !pk->d_n_words) return AHA_E_INVALID;
    if (stream_fmt(ac).len_bits == 0 && ac->aut.n_keys > (1u << 20)) {
      tls_err = "the 4-byte exchange stream holds key ids below 2^20: use the {end, value} pairs";
      return AHA_E_INVALID;
    }
    if (pk->cap_words < 2 * cap + (cap + 1023) / 1024 + 1) {
      tls_err = "4-byte exchange stream: capacity below 2 cap + ceil(cap / 1024) + 1 words";
      return AHA_E_CAPACITY;
    }
  }
  bool packed = false;
  int32_t rc = device_match(ac, sc, d_corpus, d_doc_offsets, n_docs, n_bytes, params, d_out, cap, d_doc_hit_offsets, n_hits,
                                 stream, offsets_checked, pk, &packed);
  if (rc == AHA_OK && pk && !packed) {
    // the three pack kernels over the hits, behind the match on the same stream (the expansion writing the words itself was
    // built and measured in round 5: +0.40 ms on the match against the pack's 0.25, profiles/r05_fused_exchange_stream.txt)
    DeviceGuard g(ac->device);
    launch_hits_pack4(reinterpret_cast<const int32_t *>(d_out), *n_hits, pk->d_words,
                      reinterpret_cast<unsigned long long *>(pk->d_n_words), stream_fmt(ac), stream);
    HIPCHK(ac, hipGetLastError());
    HIPCHK(ac, hipStreamSynchronize((hipStream_t)stream));
  }
  return rc;
}

int32_t aha_ac_match_batch_device_stream(aha_ac *ac, const uint8_t *d_corpus, const uint64_t *d_doc_offsets, uint64_t n_docs,
                                         uint64_t n_bytes, const aha_match_params *params, aha_hit *d_out, uint64_t cap,
                                         uint64_t *d_doc_hit_offsets, uint64_t *n_hits, uint32_t *d_words, uint64_t cap_words,
                                         uint64_t *d_n_words, void *stream) {
  if (!ac) return AHA_E_INVALID;
  if (ac->device < 0) {
    tls_err = aha_strerror(AHA_E_NO_DEVICE);
    return AHA_E_NO_DEVICE;
  }
  Lease lease(ac);
  const PackOut pk{d_words, cap_words, d_n_words};
  return match_batch_device_impl(ac, lease.get(), d_corpus, d_doc_offsets, n_docs, n_bytes, params, d_out, cap, d_doc_hit_offsets,
                                 n_hits, stream, false, &pk);
}

// Host-buffer entry (what `Aha::AC#match(Bytes)`, src/aha/ac.cr:280-286, and a batch of them bind to).  The corpus is
// cut into contiguous document ranges of about kHostRange bytes; three threads run them through a pipeline on three
// private non-blocking streams of the leased scratch set -- upload of range k+1 beside the match of range k beside the
// download of the hits of range k-1 -- so a large batch costs little more than its PCIe transfer, and no call touches
// the NULL stream (concurrent callers on one handle really run side by side).  Documents are independent
// (ac.cr:177), so the ranges are separate device batches whose hit lists concatenate.
namespace {
constexpr uint64_t kHostRange = 64ull << 20;

struct HostPipe {
  std::mutex mu;
  std::condition_variable cv;
  size_t uploaded = 0, matched = 0;
  bool failed = false;
  int32_t rc = AHA_OK;
  std::string err;
  void fail(int32_t code, const std::string &what) {
    std::lock_guard<std::mutex> lk(mu);
    if (!failed) {
      failed = true;
      rc = code;
      err = what;
    }
    cv.notify_all();
  }
};

static bool host_streams(Scratch *sc) {
  for (auto &st : sc->hs)
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
  return true;
}
}  // namespace

// The host entry: upload, match and download pipelined over ranges of whole documents.  d_keep != null: the hits stay
// on the device, in the caller's buffer d_keep[0 .. cap) (aha_ac_match_batch_keep: what the shards of a group call), and
// only the per-document offsets come back to the host.
// (d_keep AND a host copy: both -- the hits stay on the device and the ranges' hits also go to hc->out[0 .. hc->cap) while
// they fit; what a shard of a group calls.  The place may become known while the call runs: the copies wait for hc->ready.)
static int32_t match_batch_host(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                const aha_match_params *params, aha_hit *out, aha_hit *d_keep, uint64_t cap,
                                uint64_t *doc_hit_offsets, uint64_t *n_hits, aha_internal_host_copy *hc = nullptr);

int32_t aha_ac_match_batch(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets,
                           uint64_t n_docs, const aha_match_params *params, aha_hit *out,
                           uint64_t cap, uint64_t *doc_hit_offsets, uint64_t *n_hits) {
  if (cap && !out) return AHA_E_INVALID;
  return match_batch_host(ac, corpus, doc_offsets, n_docs, params, out, nullptr, cap, doc_hit_offsets, n_hits);
}

int32_t aha_ac_match_batch_keep(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                const aha_match_params *params, aha_hit *d_hits, uint64_t cap,
                                uint64_t *doc_hit_offsets, uint64_t *n_hits) {
  if (cap && !d_hits) return AHA_E_INVALID;
  static aha_hit none;  // (cap == 0: counting only; the pointer only says "keep")
  return match_batch_host(ac, corpus, doc_offsets, n_docs, params, nullptr, d_hits ? d_hits : &none, cap, doc_hit_offsets,
                          n_hits);
}

// library-internal (group.cpp): aha_ac_match_batch_keep that also copies the hits to host memory, range by range, as long as
// they fit host_cap hits -- the copy never makes the call fail (a group reports AHA_E_CAPACITY from its total)
int32_t aha_internal_match_batch_keep_copy(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                           const aha_match_params *params, aha_hit *d_hits, uint64_t cap,
                                           aha_internal_host_copy *hc, uint64_t *doc_hit_offsets, uint64_t *n_hits) {
  struct Done {  // whatever way the call ends, whoever waits for its uploads goes on
    aha_internal_host_copy *hc;
    ~Done() {
      if (hc) hc->set_uploads_done();
    }
  } done{hc};
  if (cap && !d_hits) return AHA_E_INVALID;
  static aha_hit none;
  return match_batch_host(ac, corpus, doc_offsets, n_docs, params, nullptr, d_hits ? d_hits : &none, cap, doc_hit_offsets,
                          n_hits, hc);
}

static int32_t match_batch_host(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                const aha_match_params *params, aha_hit *out, aha_hit *d_keep, uint64_t cap,
                                uint64_t *doc_hit_offsets, uint64_t *n_hits, aha_internal_host_copy *hc) {
  if (!ac || !doc_offsets || !n_hits) return AHA_E_INVALID;
  if (ac->device < 0) {
    tls_err = aha_strerror(AHA_E_NO_DEVICE);
    return AHA_E_NO_DEVICE;
  }
  if (doc_offsets[0] != 0) return AHA_E_INVALID;
  for (uint64_t d = 0; d < n_docs; d++) {
    if (doc_offsets[d + 1] < doc_offsets[d]) return AHA_E_INVALID;
    if (doc_offsets[d + 1] - doc_offsets[d] >= 0x7FFFFFFFull) return AHA_E_TOO_LONG;
  }
  const uint64_t n_bytes = doc_offsets[n_docs];
  if (n_bytes && !corpus) return AHA_E_INVALID;
  *n_hits = 0;
  DeviceGuard g(ac->device);
  // device staging buffers are kept in the leased scratch set (grow-only): the reference's
  // usage is one #match per string, so per-call hipMalloc/hipFree would dominate
  Lease lease(ac);
  Scratch *sc = lease.get();
  auto reserve = [&](int i, size_t bytes) -> void * {
    Buf &b = sc->hostbuf[i];
    if (b.bytes < bytes) {
      if (b.p) (void)hipFree(b.p);
      b.p = nullptr;
      b.bytes = 0;
      size_t want = bytes + bytes / 4 + 4096;
      if (hipMalloc(&b.p, want) != hipSuccess) return nullptr;
      b.bytes = want;
    }
    return b.p;
  };
  // ranges: document boundaries nearest to multiples of kHostRange (whole documents only)
  std::vector<uint64_t> bounds;
  try {
    bounds.push_back(0);
    const uint64_t parts = std::max<uint64_t>(1, std::min<uint64_t>((n_bytes + kHostRange - 1) / kHostRange, n_docs));
    for (uint64_t r = 1; r < parts; r++) {
      const uint64_t target = (uint64_t)(((__uint128_t)n_bytes * r) / parts);
      uint64_t d = (uint64_t)(std::lower_bound(doc_offsets, doc_offsets + n_docs + 1, target) - doc_offsets);
      if (d > 0 && (d > n_docs || target - doc_offsets[d - 1] < doc_offsets[d] - target)) d--;
      d = std::min(std::max(d, bounds.back()), n_docs);
      if (d > bounds.back()) bounds.push_back(d);
    }
    if (bounds.back() != n_docs || bounds.size() == 1) bounds.push_back(n_docs);
  } catch (...) {
    return AHA_E_NOMEM;
  }
  const size_t R = bounds.size() - 1;
  // device layout: range k starts 256-byte aligned (the kernels read the corpus in aligned 16-byte pieces)
  std::vector<uint64_t> dev_off(R + 1, 0), rel;
  try {
    for (size_t k = 0; k < R; k++)
      dev_off[k + 1] = (dev_off[k] + (doc_offsets[bounds[k + 1]] - doc_offsets[bounds[k]]) + 255) & ~255ull;
    rel.resize(n_docs + R + 1);  // range k's offsets, relative to its first byte, at rel[bounds[k] + k ..]
    for (size_t k = 0; k < R; k++)
      for (uint64_t d = bounds[k]; d <= bounds[k + 1]; d++) rel[d + k] = doc_offsets[d] - doc_offsets[bounds[k]];
  } catch (...) {
    return AHA_E_NOMEM;
  }
  uint8_t *d_corpus = (uint8_t *)reserve(0, dev_off[R] + 64);
  uint64_t *d_doc = (uint64_t *)reserve(1, (n_docs + R + 1) * sizeof(uint64_t));
  uint64_t *d_dho = (uint64_t *)reserve(2, (n_docs + R + 1) * sizeof(uint64_t));
  aha_hit *d_out = !cap ? nullptr : d_keep ? d_keep : (aha_hit *)reserve(3, cap * sizeof(aha_hit));
  if (!d_corpus || !d_doc || !d_dho || (cap && !d_out) || !host_streams(sc)) {
    tls_err = "hipMalloc / hipStreamCreate failed for the staging buffers";
    return AHA_E_HIP;
  }
  hipStream_t s_up = sc->hs[0], s_match = sc->hs[1], s_down = sc->hs[2];
  if (hipMemcpyAsync(d_doc, rel.data(), (n_docs + R + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s_up) != hipSuccess ||
      hipStreamSynchronize(s_up) != hipSuccess) {
    tls_err = "upload of the document offsets failed";
    return AHA_E_HIP;
  }

  HostPipe P;
  std::vector<uint64_t> base(R + 1, 0), got(R, 0);  // hits before range k; hits of range k that are in d_out
  const int device = ac->device;
  auto uploader = [&]() {
    if (hipSetDevice(device) != hipSuccess) return P.fail(AHA_E_HIP, "hipSetDevice failed");
    for (size_t k = 0; k < R; k++) {
      const uint64_t b0 = doc_offsets[bounds[k]], nb = doc_offsets[bounds[k + 1]] - b0;
      if (nb && (hipMemcpyAsync(d_corpus + dev_off[k], corpus + b0, nb, hipMemcpyHostToDevice, s_up) != hipSuccess ||
                 hipStreamSynchronize(s_up) != hipSuccess))
        return P.fail(AHA_E_HIP, "upload of the corpus failed");
      std::lock_guard<std::mutex> lk(P.mu);
      if (P.failed) return;
      P.uploaded = k + 1;
      P.cv.notify_all();
    }
    if (hc) hc->set_uploads_done();  // (a group starts its next shard's uploads now)
  };
  auto downloader = [&]() {
    if (hipSetDevice(device) != hipSuccess) return P.fail(AHA_E_HIP, "hipSetDevice failed");
    std::vector<uint64_t> tmp;
    for (size_t k = 0; k < R; k++) {
      {
        std::unique_lock<std::mutex> lk(P.mu);
        P.cv.wait(lk, [&] { return P.failed || P.matched > k; });
        if (P.failed) return;
      }
      const uint64_t D = bounds[k + 1] - bounds[k];
      aha_hit *to = out;
      if (d_keep) {  // the keep form: a copy only where the caller of the library-internal entry has said where to
        to = nullptr;
        if (hc && got[k]) {
          hc->wait_ready();  // (set when the shards before have counted)
          if (hc->out && base[k] + got[k] <= hc->cap) to = hc->out;
        }
      }
      if (got[k] && to &&
          hipMemcpyAsync(to + base[k], d_out + base[k], got[k] * sizeof(aha_hit), hipMemcpyDeviceToHost, s_down) != hipSuccess)
        return P.fail(AHA_E_HIP, "download of the hits failed");
      if (doc_hit_offsets) {
        try {
          tmp.resize(D + 1);
        } catch (...) {
          return P.fail(AHA_E_NOMEM, "out of host memory");
        }
        if (hipMemcpyAsync(tmp.data(), d_dho + bounds[k] + k, (D + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost,
                           s_down) != hipSuccess)
          return P.fail(AHA_E_HIP, "download of the document offsets failed");
      }
      if (hipStreamSynchronize(s_down) != hipSuccess) return P.fail(AHA_E_HIP, "download failed");
      if (doc_hit_offsets)
        for (uint64_t d = 0; d <= D; d++) doc_hit_offsets[bounds[k] + d] = base[k] + tmp[d];  // [.. + D] rewritten by k+1
    }
  };
  std::thread t_up, t_down;
  if (R == 1) {
    uploader();  // one range (a single #match, a small batch): nothing to overlap, no threads
  } else {
    try {
      t_up = std::thread(uploader);
      t_down = std::thread(downloader);
    } catch (...) {
      P.fail(AHA_E_NOMEM, "thread creation failed");
    }
  }
  // the matches, in order, on this thread (the error text of a match is this thread's)
  uint64_t total = 0;
  bool overflow = false;
  for (size_t k = 0; k < R; k++) {
    {
      std::unique_lock<std::mutex> lk(P.mu);
      P.cv.wait(lk, [&] { return P.failed || P.uploaded > k; });
      if (P.failed) break;
    }
    const uint64_t D = bounds[k + 1] - bounds[k], nb = doc_offsets[bounds[k + 1]] - doc_offsets[bounds[k]];
    const uint64_t room = (!overflow && cap > total) ? cap - total : 0;
    uint64_t nh = 0;
    int32_t rc = match_batch_device_impl(ac, sc, d_corpus + dev_off[k], d_doc + bounds[k] + k, D, nb, params,
                                         room ? d_out + total : nullptr, room, d_dho + bounds[k] + k, &nh, s_match,
                                         true);  // the offsets were checked on the host above
    if (rc != AHA_OK && rc != AHA_E_CAPACITY) {
      P.fail(rc, tls_err);
      break;
    }
    base[k] = total;
    got[k] = std::min(nh, room);
    total += nh;
    if (rc == AHA_E_CAPACITY) overflow = true;  // later ranges are only counted (their offsets stay valid)
    std::lock_guard<std::mutex> lk(P.mu);
    P.matched = k + 1;
    P.cv.notify_all();
  }
  base[R] = total;
  if (R == 1) downloader();
  if (t_up.joinable()) t_up.join();
  if (t_down.joinable()) t_down.join();
  if (P.failed) {
    tls_err = P.err;
    return P.rc;
  }
  *n_hits = total;
  if (total > cap) {
    tls_err = "output buffer too small";
    return AHA_E_CAPACITY;
  }
  return AHA_OK;
}

// ---- device buffers behind the C ABI (include/aha_hip.h) ---------------------------------------------------------
namespace {
std::mutex g_copy_mu;
static std::vector<hipStream_t> g_copy_streams;  // one private non-blocking stream per device, created on first use

static int32_t copy_stream(int device, hipStream_t *out) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    tls_err = aha_strerror(AHA_E_NO_DEVICE);
    return AHA_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    tls_err = "no such device";
    return AHA_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(g_copy_mu);
  if (g_copy_streams.size() < (size_t)n) g_copy_streams.resize((size_t)n, nullptr);
  if (!g_copy_streams[device] && hipStreamCreateWithFlags(&g_copy_streams[device], hipStreamNonBlocking) != hipSuccess) {
    tls_err = "hipStreamCreate failed";
    return AHA_E_HIP;
  }
  *out = g_copy_streams[device];
  return AHA_OK;
}

static int32_t buffer_copy(int device, void *dst, const void *src, uint64_t bytes, hipMemcpyKind kind) {
  if (bytes && (!dst || !src)) return AHA_E_INVALID;
  DeviceGuard g(device);
  hipStream_t st = nullptr;
  int32_t rc = copy_stream(device, &st);
  if (rc != AHA_OK || !bytes) return rc;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    tls_err = std::string("copy: ") + hipGetErrorString(e);
    return AHA_E_HIP;
  }
  return AHA_OK;
}
}  // namespace

struct aha_corpus {
  int device = -1;
  void *bytes = nullptr;
  void *doc = nullptr;
  uint64_t n_docs = 0, n_bytes = 0;
};

int32_t aha_buffer_alloc(int32_t device, uint64_t bytes, void **d_ptr) {
  if (!d_ptr) return AHA_E_INVALID;
  *d_ptr = nullptr;
  hipStream_t st = nullptr;
  int32_t rc = copy_stream(device, &st);  // validates the device
  if (rc != AHA_OK) return rc;
  DeviceGuard g(device);
  hipError_t e = hipMalloc(d_ptr, std::max<uint64_t>(bytes, 16));
  if (e != hipSuccess) {
    tls_err = std::string("hipMalloc: ") + hipGetErrorString(e);
    *d_ptr = nullptr;
    return AHA_E_HIP;
  }
  return AHA_OK;
}

int32_t aha_buffer_free(int32_t device, void *d_ptr) {
  if (!d_ptr) return AHA_OK;
  DeviceGuard g(device);
  return hipFree(d_ptr) == hipSuccess ? AHA_OK : AHA_E_HIP;
}

int32_t aha_buffer_upload(int32_t device, void *d_dst, const void *src, uint64_t bytes) {
  return buffer_copy(device, d_dst, src, bytes, hipMemcpyHostToDevice);
}

int32_t aha_buffer_download(int32_t device, void *dst, const void *d_src, uint64_t bytes) {
  return buffer_copy(device, dst, d_src, bytes, hipMemcpyDeviceToHost);
}

int32_t aha_corpus_upload(int32_t device, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                          aha_corpus **out) {
  if (!out || !doc_offsets) return AHA_E_INVALID;
  *out = nullptr;
  if (doc_offsets[0] != 0) return AHA_E_INVALID;
  for (uint64_t d = 0; d < n_docs; d++) {
    if (doc_offsets[d + 1] < doc_offsets[d]) return AHA_E_INVALID;
    if (doc_offsets[d + 1] - doc_offsets[d] >= 0x7FFFFFFFull) return AHA_E_TOO_LONG;
  }
  const uint64_t n_bytes = doc_offsets[n_docs];
  if (n_bytes && !corpus) return AHA_E_INVALID;
  aha_corpus *c = new (std::nothrow) aha_corpus();
  if (!c) return AHA_E_NOMEM;
  c->device = device;
  c->n_docs = n_docs;
  c->n_bytes = n_bytes;
  int32_t rc = aha_buffer_alloc(device, n_bytes + 64, &c->bytes);
  if (rc == AHA_OK) rc = aha_buffer_alloc(device, (n_docs + 1) * sizeof(uint64_t), &c->doc);
  if (rc == AHA_OK) rc = aha_buffer_upload(device, c->bytes, corpus, n_bytes);
  if (rc == AHA_OK) rc = aha_buffer_upload(device, c->doc, doc_offsets, (n_docs + 1) * sizeof(uint64_t));
  if (rc != AHA_OK) {
    aha_corpus_free(c);
    return rc;
  }
  *out = c;
  return AHA_OK;
}

void aha_corpus_free(aha_corpus *c) {
  if (!c) return;
  (void)aha_buffer_free(c->device, c->bytes);
  (void)aha_buffer_free(c->device, c->doc);
  delete c;
}

const uint8_t *aha_corpus_bytes(const aha_corpus *c) { return c ? (const uint8_t *)c->bytes : nullptr; }
const uint64_t *aha_corpus_doc_offsets(const aha_corpus *c) { return c ? (const uint64_t *)c->doc : nullptr; }
uint64_t aha_corpus_n_docs(const aha_corpus *c) { return c ? c->n_docs : 0; }
uint64_t aha_corpus_n_bytes(const aha_corpus *c) { return c ? c->n_bytes : 0; }
int32_t aha_corpus_device(const aha_corpus *c) { return c ? c->device : -1; }

int32_t aha_ac_match_bytes(aha_ac *ac, const uint8_t *text, uint64_t n,
                           const aha_match_params *params, aha_hit *out, uint64_t cap,
                           uint64_t *n_hits) {
  uint64_t offs[2] = {0, n};
  return aha_ac_match_batch(ac, text, offs, 1, params, out, cap, nullptr, n_hits);
}

}  // extern "C"
