// synth.cpp -- deterministic synthetic keys / corpora for the BASELINE.json
// configs (SURVEY.md section 8 d).  Bench and test tooling (libaha_synth.so);
// not part of the match path.  PRNG = splitmix64 with the stated seeds.
//
//   cfg 2: 1k ASCII keys (len 4-16, a-z); corpus of space-separated tokens,
//          p=1/64 a key else a random a-z word of 1-12 letters; ~64 KiB docs.
//   cfg 3: 100k UTF-8 keys of 2-8 code points (60% CJK U+4E00-9FA5, 25% a-z,
//          15% Cyrillic U+0430-044F); corpus tokens p=1/32 a key else 1-8
//          random code points, a space after a token with p=1/2; ~1 MiB docs.
//   cfg 4: cfg 3 keys, per-rank corpora with seed+r.
//   cfg 5: 1M keys (cfg 3 generator) whose last keys are replaced by three
//          nested families (runs c..c^16, suffix-closed 16-byte words, broken
//          chains w[0..], w[1..], w[2..]+"#", w[3..]..); hit-dense corpus
//          (p=1/2 key tokens with the families over-sampled 8x, plus runs c^64).
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform in [0, n)
  uint32_t below(uint32_t n) { return (uint32_t)(((unsigned __int128)next() * n) >> 64); }
  // uniform in [lo, hi]
  uint32_t range(uint32_t lo, uint32_t hi) { return lo + below(hi - lo + 1); }
};

inline int put_cp(uint32_t cp, uint8_t *o) {
  if (cp < 0x80) {
    o[0] = (uint8_t)cp;
    return 1;
  }
  if (cp < 0x800) {
    o[0] = (uint8_t)(0xC0 | (cp >> 6));
    o[1] = (uint8_t)(0x80 | (cp & 0x3F));
    return 2;
  }
  o[0] = (uint8_t)(0xE0 | (cp >> 12));
  o[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F));
  o[2] = (uint8_t)(0x80 | (cp & 0x3F));
  return 3;
}

// one code point of the cfg-3 mix
inline uint32_t mix_cp(Rng &r) {
  uint32_t t = r.below(100);
  if (t < 60) return 0x4E00 + r.below(0x9FA5 - 0x4E00 + 1);
  if (t < 85) return 'a' + r.below(26);
  return 0x0430 + r.below(0x044F - 0x0430 + 1);
}

inline void mix_word(Rng &r, uint32_t lo, uint32_t hi, std::string &out) {
  uint32_t n = r.range(lo, hi);
  uint8_t b[4];
  out.clear();
  for (uint32_t i = 0; i < n; i++) out.append((const char *)b, (size_t)put_cp(mix_cp(r), b));
}

inline void ascii_word(Rng &r, uint32_t lo, uint32_t hi, std::string &out) {
  uint32_t n = r.range(lo, hi);
  out.resize(n);
  for (uint32_t i = 0; i < n; i++) out[i] = (char)('a' + r.below(26));
}

struct KeySet {
  std::vector<std::string> keys;
  uint32_t n_family = 0;  // the last n_family keys belong to the cfg-5 families
};

void gen_keys(int cfg, uint64_t seed, uint32_t K, KeySet &ks) {
  Rng r(seed);
  std::unordered_set<std::string> seen;
  seen.reserve((size_t)K * 2);
  std::string w;
  auto add = [&](const std::string &k) {
    if (k.empty() || !seen.insert(k).second) return false;
    ks.keys.push_back(k);
    return true;
  };
  std::vector<std::string> fam;
  if (cfg == 5) {
    uint8_t b[4];
    uint32_t nfam = K >= 96000 ? 1000 : K / 96;  // scaled down for small test sizes
    for (uint32_t i = 0; i < nfam; i++) {  // (i) runs c, cc, ..., c^16
      int n = put_cp(mix_cp(r), b);
      std::string run;
      for (int j = 0; j < 16; j++) {
        run.append((const char *)b, (size_t)n);
        fam.push_back(run);
      }
    }
    for (uint32_t i = 0; i < nfam; i++) {  // (ii) every suffix of a 16-byte word
      ascii_word(r, 16, 16, w);
      for (int j = 0; j < 16; j++) fam.push_back(w.substr((size_t)j));
    }
    for (uint32_t i = 0; i < nfam; i++) {  // (iii) broken chain: w[2..] only as w[2..]+"#"
      ascii_word(r, 16, 16, w);
      for (int j = 0; j < 16; j++) fam.push_back(j == 2 ? w.substr(2) + "#" : w.substr((size_t)j));
    }
  }
  // distinct family keys first (so that they survive de-duplication), then
  // random keys; the families are moved to the tail afterwards.
  std::vector<std::string> famkeys;
  for (auto &f : fam)
    if (famkeys.size() + 1 <= K && seen.insert(f).second) famkeys.push_back(f);
  while (ks.keys.size() + famkeys.size() < K) {
    if (cfg == 2)
      ascii_word(r, 4, 16, w);
    else
      mix_word(r, 2, 8, w);
    add(w);  // duplicates re-drawn
  }
  ks.n_family = (uint32_t)famkeys.size();
  for (auto &f : famkeys) ks.keys.push_back(f);
}

}  // namespace

extern "C" {

// Writes K keys as blob + K+1 offsets.  Returns the blob size in bytes; if it
// exceeds blob_cap nothing is written (call again with a larger buffer).
// *n_family (optional) = number of trailing keys that are cfg-5 family keys.
int64_t aha_synth_keys(int cfg, uint64_t seed, uint32_t K, uint8_t *blob, uint64_t blob_cap,
                       uint64_t *offs, uint32_t *n_family) {
  KeySet ks;
  gen_keys(cfg, seed, K, ks);
  uint64_t total = 0;
  for (auto &k : ks.keys) total += k.size();
  if (n_family) *n_family = ks.n_family;
  if (total > blob_cap || !blob || !offs) return (int64_t)total;
  uint64_t o = 0;
  for (uint32_t i = 0; i < K; i++) {
    offs[i] = o;
    memcpy(blob + o, ks.keys[i].data(), ks.keys[i].size());
    o += ks.keys[i].size();
  }
  offs[K] = o;
  return (int64_t)total;
}

// Fills out[0..n_bytes) and doc_offs (at most doc_cap+1 entries).  Documents
// are cut at token boundaries once they reach doc_bytes.  Returns the number
// of documents (doc_offs[0] = 0, doc_offs[D] = n_bytes).
int64_t aha_synth_corpus(int cfg, uint64_t seed, const uint8_t *key_blob, const uint64_t *key_offs,
                         uint32_t K, uint32_t n_family, uint64_t n_bytes, uint64_t doc_bytes,
                         uint8_t *out, uint64_t *doc_offs, uint64_t doc_cap) {
  Rng r(seed);
  uint64_t pos = 0, D = 0, doc_start = 0;
  doc_offs[0] = 0;
  std::string w;
  uint8_t cpb[4];
  const uint32_t n_plain = K - n_family;
  auto put = [&](const uint8_t *p, uint64_t n) {
    if (pos + n > n_bytes) return false;
    memcpy(out + pos, p, n);
    pos += n;
    return true;
  };
  auto put_key = [&](uint32_t k) { return put(key_blob + key_offs[k], key_offs[k + 1] - key_offs[k]); };
  while (pos < n_bytes) {
    bool ok = true;
    if (cfg == 2) {
      if (r.below(64) == 0) {
        ok = put_key(r.below(K));
      } else {
        ascii_word(r, 1, 12, w);
        ok = put((const uint8_t *)w.data(), w.size());
      }
      if (ok) ok = put((const uint8_t *)" ", 1);
    } else if (cfg == 5) {
      uint32_t t = r.below(16);
      if (t < 8) {  // key token; family keys over-sampled 8x
        uint64_t wt = (uint64_t)n_plain + 8ull * n_family;
        uint64_t x = ((unsigned __int128)r.next() * wt) >> 64;
        uint32_t k = x < n_plain ? (uint32_t)x : n_plain + (uint32_t)((x - n_plain) / 8);
        ok = put_key(k);
      } else if (t == 8 && n_family) {  // a run c^64: first code point of a random family key
        uint32_t k = n_plain + r.below(n_family);
        const uint8_t *kp = key_blob + key_offs[k];
        uint64_t cl = kp[0] < 0x80 ? 1 : (kp[0] < 0xE0 ? 2 : 3);
        for (int i = 0; i < 64 && ok; i++) ok = put(kp, cl);
      } else {
        mix_word(r, 1, 8, w);
        ok = put((const uint8_t *)w.data(), w.size());
      }
      if (ok && r.below(2)) ok = put((const uint8_t *)" ", 1);
    } else {  // cfg 3 / 4
      if (r.below(32) == 0) {
        ok = put_key(r.below(K));
      } else {
        mix_word(r, 1, 8, w);
        ok = put((const uint8_t *)w.data(), w.size());
      }
      if (ok && r.below(2)) ok = put((const uint8_t *)" ", 1);
    }
    if (!ok) {  // the token does not fit: pad the tail with spaces
      memset(out + pos, ' ', n_bytes - pos);
      pos = n_bytes;
    }
    if (pos - doc_start >= doc_bytes && pos < n_bytes && D + 1 < doc_cap) {
      doc_offs[++D] = pos;
      doc_start = pos;
    }
  }
  (void)cpb;
  doc_offs[++D] = n_bytes;
  return (int64_t)D;
}

}  // extern "C"
