// aha/ac.hpp -- C++ host-side mirror of the reference's Crystal API for the
// accelerated path, on top of the C ABI (include/aha_hip.h):
//
//   Aha::AC.compile(keys)                    src/aha/ac.cr:62-69
//   AC#match(seq : Bytes) { |hit| }          src/aha/ac.cr:280-286
//   AC#match(seq : String) { |hit| }         src/aha/matcher.cr:34-39  (char offsets)
//   AC#match(seq, sep : BitArray) { |hit| }  src/aha/ac.cr:321-340, matcher.cr:41-46
//   AC#[](id) / AC#[](key)                   src/aha/ac.cr:41-43
//   Aha::Hit#start/#end/#value               src/aha/matcher.cr:2-11
//
// Errors are thrown as aha::Error carrying the reference's message text.
// Header-only; link with -laha_hip.  No CPU fallback: matching needs a GPU.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../aha_hip.h"

namespace aha {

using Hit = aha_hit;  // {start, end, value : Int32}

struct Error : std::runtime_error {
  int32_t code;
  uint32_t key_index;
  Error(int32_t c, const std::string &msg, uint32_t k = 0) : std::runtime_error(msg), code(c), key_index(k) {}
};

// BitArray as used by match(seq, sep)
class BitArray {
 public:
  explicit BitArray(int size) : size_(size), bits_((size_t)(size > 0 ? (size + 7) / 8 : 1), 0) {}
  int size() const { return size_; }
  void set(int i, bool v = true) {
    if (i < 0 || i >= size_) throw std::out_of_range("BitArray index");
    if (v)
      bits_[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
    else
      bits_[(size_t)i >> 3] &= (uint8_t)~(1u << (i & 7));
  }
  const std::vector<uint8_t> &bytes() const { return bits_; }

 private:
  int size_;
  std::vector<uint8_t> bits_;
};

// A batch resident in HBM (aha_corpus_upload): uploaded once, matched as often as wanted -- by AC::match_corpus.
class Corpus {
 public:
  Corpus(std::string_view bytes, const std::vector<uint64_t> &doc_offsets, int device = 0) {
    if (doc_offsets.empty()) throw Error(AHA_E_INVALID, "doc_offsets holds D + 1 entries");
    int32_t rc = aha_corpus_upload(device, reinterpret_cast<const uint8_t *>(bytes.data()), doc_offsets.data(),
                                   doc_offsets.size() - 1, &c_);
    if (rc != AHA_OK) {
      const char *m = aha_last_error(nullptr);
      throw Error(rc, (m && *m) ? m : aha_strerror(rc));
    }
  }
  Corpus(const Corpus &) = delete;
  Corpus &operator=(const Corpus &) = delete;
  ~Corpus() { aha_corpus_free(c_); }
  const aha_corpus *handle() const { return c_; }

 private:
  aha_corpus *c_ = nullptr;
};

class AC {
 public:
  AC(const AC &) = delete;
  AC &operator=(const AC &) = delete;
  AC(AC &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~AC() { aha_ac_free(h_); }

  // Aha::AC.compile(keys)
  static AC compile(const std::vector<std::string> &keys, int device = -1) {
    std::vector<uint8_t> blob;
    std::vector<uint64_t> offs(keys.size() + 1, 0);
    for (size_t i = 0; i < keys.size(); i++) {
      blob.insert(blob.end(), keys[i].begin(), keys[i].end());
      offs[i + 1] = blob.size();
    }
    aha_options o{};
    o.struct_size = sizeof(o);
    o.device = device;
    aha_ac *h = nullptr;
    uint32_t bad = 0;
    int32_t rc = aha_ac_compile(blob.data(), offs.data(), (uint32_t)keys.size(), &o, &h, &bad);
    if (rc == AHA_E_DUP_KEY) throw Error(rc, "key:" + keys[bad] + " appear twice.", bad);  // ac.cr:66
    if (rc != AHA_OK) throw Error(rc, aha_strerror(rc), bad);
    return AC(h);
  }

  // AC#match(seq : Bytes) -- byte offsets; hits in the reference's order
  template <class F>
  void match(std::string_view seq, F &&block) const {
    run(seq, nullptr, false, std::forward<F>(block));
  }
  // AC#match(seq : String) -- char offsets (valid UTF-8)
  template <class F>
  void match_string(std::string_view seq, F &&block) const {
    run(seq, nullptr, true, std::forward<F>(block));
  }
  // AC#match(seq, sep)
  template <class F>
  void match(std::string_view seq, const BitArray &sep, F &&block, bool chars = false) const {
    run(seq, &sep, chars, std::forward<F>(block));
  }
  std::vector<Hit> match(std::string_view seq, bool chars = false) const {
    std::vector<Hit> v;
    run(seq, nullptr, chars, [&](const Hit &h) { v.push_back(h); });
    return v;
  }
  // AC#match_longest(seq, intersectable = false) -- src/aha/ac.cr:297-319 (String form: chars = true)
  template <class F>
  void match_longest(std::string_view seq, bool intersectable, F &&block, bool chars = false) const {
    run(seq, nullptr, chars, std::forward<F>(block), intersectable ? 2 : 1);
  }
  // D documents in one call (new: the reference is one sequence per call): hits of all documents in document order,
  // doc_hit_offsets[d] .. doc_hit_offsets[d + 1] are document d's
  std::vector<Hit> match_batch(std::string_view corpus, const std::vector<uint64_t> &doc_offsets,
                               std::vector<uint64_t> *doc_hit_offsets = nullptr, bool chars = false) const {
    if (doc_offsets.empty()) throw Error(AHA_E_INVALID, "doc_offsets holds D + 1 entries");
    aha_match_params p{};
    p.struct_size = sizeof(p);
    p.char_offsets = chars ? 1 : 0;
    const uint64_t D = doc_offsets.size() - 1;
    std::vector<uint64_t> dho(D + 1);
    std::vector<Hit> out(corpus.size() / 8 + 64);
    uint64_t n = 0;
    for (;;) {
      int32_t rc = aha_ac_match_batch(h_, reinterpret_cast<const uint8_t *>(corpus.data()), doc_offsets.data(), D, &p,
                                      out.data(), out.size(), dho.data(), &n);
      if (rc == AHA_E_CAPACITY) {
        out.resize(n);
        continue;
      }
      if (rc != AHA_OK) {
        const char *m = aha_last_error(h_);
        throw Error(rc, (m && *m) ? m : aha_strerror(rc));
      }
      break;
    }
    out.resize(n);
    if (doc_hit_offsets) *doc_hit_offsets = std::move(dho);
    return out;
  }

  // The same on a batch that already lives in HBM: the device entry point on the library's own buffers (no copy of
  // the corpus per call; the hits are downloaded at the end).
  std::vector<Hit> match_corpus(const Corpus &c, std::vector<uint64_t> *doc_hit_offsets = nullptr,
                                bool chars = false) const {
    const aha_corpus *h = c.handle();
    const int dev = aha_corpus_device(h);
    const uint64_t D = aha_corpus_n_docs(h);
    aha_match_params p{};
    p.struct_size = sizeof(p);
    p.char_offsets = chars ? 1 : 0;
    void *d_dho = nullptr, *d_out = nullptr;
    uint64_t cap = aha_corpus_n_bytes(h) / 8 + 64, n = 0;
    auto fail = [&](int32_t rc, const char *m) {
      aha_buffer_free(dev, d_dho);
      aha_buffer_free(dev, d_out);
      throw Error(rc, (m && *m) ? m : aha_strerror(rc));
    };
    int32_t rc = aha_buffer_alloc(dev, (D + 1) * sizeof(uint64_t), &d_dho);
    if (rc != AHA_OK) fail(rc, aha_last_error(nullptr));
    for (;;) {
      if ((rc = aha_buffer_alloc(dev, cap * sizeof(Hit), &d_out)) != AHA_OK) fail(rc, aha_last_error(nullptr));
      rc = aha_ac_match_batch_device(h_, aha_corpus_bytes(h), aha_corpus_doc_offsets(h), D, aha_corpus_n_bytes(h), &p,
                                     static_cast<Hit *>(d_out), cap, static_cast<uint64_t *>(d_dho), &n, nullptr);
      if (rc != AHA_E_CAPACITY) break;
      aha_buffer_free(dev, d_out);
      d_out = nullptr;
      cap = n;
    }
    if (rc != AHA_OK) fail(rc, aha_last_error(h_));
    std::vector<Hit> out(n);
    std::vector<uint64_t> dho(D + 1);
    if ((rc = aha_buffer_download(dev, out.data(), d_out, n * sizeof(Hit))) != AHA_OK ||
        (rc = aha_buffer_download(dev, dho.data(), d_dho, (D + 1) * sizeof(uint64_t))) != AHA_OK)
      fail(rc, aha_last_error(nullptr));
    aha_buffer_free(dev, d_dho);
    aha_buffer_free(dev, d_out);
    if (doc_hit_offsets) *doc_hit_offsets = std::move(dho);
    return out;
  }

  // AC#[](sid : Int) : String ;  AC#[](key) : Int (IndexError when absent)
  std::string operator[](int32_t id) const {
    int32_t n = aha_ac_key(h_, id, nullptr, 0);
    if (n < 0) throw Error(n, "Index out of bounds");
    std::string s((size_t)n, '\0');
    aha_ac_key(h_, id, reinterpret_cast<uint8_t *>(&s[0]), n);
    return s;
  }
  int32_t operator[](std::string_view key) const {
    int32_t r = aha_ac_id(h_, reinterpret_cast<const uint8_t *>(key.data()), (int32_t)key.size());
    if (r < 0) throw Error(r, "Index out of bounds");
    return r;
  }
  // AC#save / AC.load (src/aha/ac.cr:45-60): the library's own container, see aha_hip.h
  std::vector<uint8_t> to_bytes() const {
    int64_t n = aha_ac_save(h_, nullptr, 0);
    if (n < 0) throw Error((int32_t)n, aha_strerror((int32_t)n));
    std::vector<uint8_t> v((size_t)n);
    aha_ac_save(h_, v.data(), (uint64_t)n);
    return v;
  }
  static AC from_bytes(const std::vector<uint8_t> &data, int device = -1) {
    aha_options o{};
    o.struct_size = sizeof(o);
    o.device = device;
    aha_ac *h = nullptr;
    int32_t rc = aha_ac_load(data.data(), data.size(), &o, &h);
    if (rc != AHA_OK) throw Error(rc, aha_strerror(rc));
    return AC(h);
  }
  aha_ac *handle() const { return h_; }

 private:
  explicit AC(aha_ac *h) : h_(h) {}
  template <class F>
  void run(std::string_view seq, const BitArray *sep, bool chars, F &&block, int longest = 0) const {
    aha_match_params p{};
    p.struct_size = sizeof(p);
    p.char_offsets = chars ? 1 : 0;
    p.longest = longest;
    if (sep) {
      p.sep_size = sep->size();
      std::memcpy(p.sep_bits, sep->bytes().data(), sep->bytes().size() < 32 ? sep->bytes().size() : 32);
    }
    std::vector<Hit> out(seq.size() / 4 + 64);
    uint64_t n = 0;
    for (;;) {
      int32_t rc = aha_ac_match_bytes(h_, reinterpret_cast<const uint8_t *>(seq.data()), seq.size(), &p,
                                      out.data(), out.size(), &n);
      if (rc == AHA_E_CAPACITY) {
        out.resize(n);
        continue;
      }
      if (rc != AHA_OK) {
        const char *m = aha_last_error(h_);
        throw Error(rc, (m && *m) ? m : aha_strerror(rc));
      }
      break;
    }
    for (uint64_t i = 0; i < n; i++) block(out[i]);
  }
  aha_ac *h_;
};

// Aha::ACBig = ACX(Int64) (src/aha/ac.cr:9): wider node ids, the same Hit with an Int32 value (ac.cr:273)
using ACBig = AC;

// Several GPUs of one node behind one object (aha_group_*): contiguous byte-balanced document ranges, one per device entry,
// all-gatherv of the hit buffers.  match_batch: host buffers in and out; upload + match_resident: the batch stays on the
// devices, the hits too (shard_hits reads one device's copy of the whole ordered stream).
class Group {
 public:
  Group(const Group &) = delete;
  Group &operator=(const Group &) = delete;
  ~Group() { aha_group_free(g_); }

  static Group compile(const std::vector<std::string> &keys, const std::vector<int32_t> &devices) {
    if (aha_abi_version() != AHA_ABI_VERSION) throw Error(AHA_E_INVALID, "libaha_hip.so is not the ABI this header declares");
    std::vector<uint8_t> blob;
    std::vector<uint64_t> offs(keys.size() + 1, 0);
    for (size_t i = 0; i < keys.size(); i++) {
      blob.insert(blob.end(), keys[i].begin(), keys[i].end());
      offs[i + 1] = blob.size();
    }
    aha_group *g = nullptr;
    uint32_t bad = 0;
    int32_t rc = aha_group_compile(blob.data(), offs.data(), (uint32_t)keys.size(), devices.data(), (int32_t)devices.size(), 0, &g,
                                   &bad);
    if (rc == AHA_E_DUP_KEY) throw Error(rc, "key:" + keys[bad] + " appear twice.", bad);
    if (rc != AHA_OK) throw Error(rc, aha_strerror(rc), bad);
    return Group(g);
  }
  Group(Group &&o) noexcept : g_(o.g_) { o.g_ = nullptr; }

  std::vector<Hit> match_batch(std::string_view corpus, const std::vector<uint64_t> &doc_offsets,
                               std::vector<uint64_t> *doc_hit_offsets = nullptr, bool chars = false) const {
    if (doc_offsets.empty()) throw Error(AHA_E_INVALID, "doc_offsets holds D + 1 entries");
    aha_match_params p{};
    p.struct_size = sizeof(p);
    p.char_offsets = chars ? 1 : 0;
    std::vector<uint64_t> dho(doc_offsets.size(), 0);
    std::vector<Hit> out(corpus.size() / 8 + 64);
    uint64_t n = 0;
    for (;;) {
      int32_t rc = aha_group_match_batch(g_, reinterpret_cast<const uint8_t *>(corpus.data()), doc_offsets.data(),
                                         doc_offsets.size() - 1, &p, out.data(), out.size(), dho.data(), &n);
      if (rc == AHA_E_CAPACITY) {
        out.resize(n);
        continue;
      }
      if (rc != AHA_OK) throw Error(rc, aha_group_last_error(g_));
      break;
    }
    out.resize(n);
    if (doc_hit_offsets) *doc_hit_offsets = dho;
    return out;
  }

  // the batch resident on the devices (aha_group_corpus_upload); must not outlive its group
  class Resident {
   public:
    Resident(const Resident &) = delete;
    Resident &operator=(const Resident &) = delete;
    Resident(Resident &&o) noexcept : c_(o.c_), n_docs_(o.n_docs_) { o.c_ = nullptr; }
    ~Resident() { aha_group_corpus_free(c_); }
    uint64_t n_docs() const { return n_docs_; }
    const aha_group_corpus *handle() const { return c_; }

   private:
    friend class Group;
    Resident(aha_group_corpus *c, uint64_t n) : c_(c), n_docs_(n) {}
    aha_group_corpus *c_;
    uint64_t n_docs_;
  };
  Resident upload(std::string_view corpus, const std::vector<uint64_t> &doc_offsets) const {
    if (doc_offsets.empty()) throw Error(AHA_E_INVALID, "doc_offsets holds D + 1 entries");
    aha_group_corpus *c = nullptr;
    int32_t rc = aha_group_corpus_upload(g_, reinterpret_cast<const uint8_t *>(corpus.data()), doc_offsets.data(),
                                         doc_offsets.size() - 1, &c);
    if (rc != AHA_OK) throw Error(rc, aha_group_last_error(g_));
    return Resident(c, doc_offsets.size() - 1);
  }
  // every device matches its resident range, then the all-gatherv: the hit count (the hits stay on the devices)
  uint64_t match_resident(const Resident &c, std::vector<uint64_t> *doc_hit_offsets = nullptr, bool chars = false) const {
    aha_match_params p{};
    p.struct_size = sizeof(p);
    p.char_offsets = chars ? 1 : 0;
    std::vector<uint64_t> dho(c.n_docs() + 1, 0);
    uint64_t n = 0;
    int32_t rc = aha_group_match_batch_device(g_, c.handle(), &p, dho.data(), &n);
    if (rc != AHA_OK) throw Error(rc, aha_group_last_error(g_));
    if (doc_hit_offsets) *doc_hit_offsets = dho;
    return n;
  }
  std::vector<Hit> shard_hits(int32_t shard) const {
    uint64_t n = 0;
    aha_group_download_shard(g_, shard, nullptr, 0, &n);
    std::vector<Hit> out(n + 1);
    int32_t rc = aha_group_download_shard(g_, shard, out.data(), out.size(), &n);
    if (rc != AHA_OK) throw Error(rc, aha_group_last_error(g_));
    out.resize(n);
    return out;
  }
  aha_group *handle() const { return g_; }

 private:
  explicit Group(aha_group *g) : g_(g) {}
  aha_group *g_;
};

}  // namespace aha
