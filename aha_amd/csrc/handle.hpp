// handle.hpp -- what capi.cpp (the C ABI: compile, export, buffers, the host entry points) and engine.cpp (which kernels answer a
// call: engine choice, pipelines, retries, the prefix filter's back-off) share: the handle, a call's scratch set and the helpers
// both sides use.  Library-internal; nothing here is exported (exports.map).
#pragma once
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>
#include "automaton.hpp"
#include "cedar_replay.hpp"
#include "image.hpp"
#include "unit.hpp"
#include "internal.hpp"

namespace ahai {
using namespace aha;
struct Buf {
  void *p = nullptr;
  size_t bytes = 0;
};
// Device scratch of ONE match call (grow-only, reused by later calls that lease the same set).
struct Scratch {
  std::mutex mu;  // held by the call that leased the set
  uint32_t *d_counts = nullptr, *d_leads = nullptr;
  uint64_t *d_blk_hits = nullptr, *d_blk_leads = nullptr, *d_docg = nullptr, *d_totals = nullptr;
  uint64_t cap_chunks = 0, cap_blocks = 0, cap_docs = 0;
  uint64_t *h_totals = nullptr;  // pinned
  hipEvent_t ev[6] = {};
  bool ev_ready = false;
  Buf v2buf[26];
  Buf hostbuf[4];  // device staging of the host-buffer entry points (corpus, doc offsets, doc hit offsets, hits)
  hipStream_t hs[3] = {};  // host-buffer entry: private non-blocking streams for upload, match, download
  unsigned long long *h_v2 = nullptr;  // pinned: cursor[2] + totals[3]
  unsigned long long *h_v2_dev = nullptr;  // the same words as the device addresses them
  // v2buf[9] holds TWO blocks of 16 counter words (and a third of odd words): a call counts in one of them, and its last kernel
  // clears the other for the call behind it -- no memset in front of a call (5 us of a 64 MiB call).  Dirty: the blocks are
  // not known to be clear (new buffer, a call that did not run to its end): the next call clears both itself.
  const void *cursor_buf = nullptr;
  bool cursor_dirty = true;
  uint32_t cursor_phase = 0;
};
constexpr size_t kCursorBytes = 3 * 16 * 8;
constexpr size_t kMaxScratch = 8;
// last error text of the calling thread (aha_last_error): calls on one handle may run concurrently
extern thread_local std::string tls_err;
}  // namespace ahai

using namespace aha;  // (the library's own translation units only)
struct aha_ac {
  Automaton aut;
  Image img;  // host copy of the device image (export / debugging)
  uint32_t n_slots = 0;
  uint32_t slot_bytes = 0;
  bool compact = false;
  uint64_t image_bytes = 0;
  int device = -1;
  DevAut dev{};
  std::vector<void *> dev_allocs;
  // per-call scratch sets: a match call leases one for its duration (Lease below); concurrent calls on one handle
  // get different sets, up to kMaxScratch of them, then wait
  std::mutex pool_mu;
  std::vector<std::unique_ptr<ahai::Scratch>> pool;
  // profiling
  std::atomic<bool> profiling{false};
  std::mutex last_mu;
  aha_timing last{};
  uint32_t chunk = 256;
  // single-traversal engine (scan_v2.hip)
  bool v2_ok = false;
  uint32_t v2_lds_slots = 0;
  uint32_t v2_grid = 0;
  uint32_t v2_bpc = 1;
  // prefix-filter engine (scan_filter.hip): blocked Bloom filter over the keys' first pf_d bytes; usable when the keys are at
  // least 3 bytes long, none longer than 64, the image compact and the filter at most a quarter full
  std::vector<uint32_t> pf_bloom;
  uint32_t pf_d = 0;
  uint32_t pf_cus = 0;
  uint32_t pf_log2 = 0;
  std::atomic<uint32_t> pf_skip[2] = {}, pf_streak[2] = {};  // calls to go without the filter; give-ups in a row ([1]: char offsets)
  bool pf_ok = false;
  FilterDev fdev{};
  uint32_t s1_lo = 0, s2_lo = 0, s2_hi = 0;   // states with base in [s2_lo, s2_hi): depth >= 3 and a fail target of depth <= 2
  // character-level image (unit.hpp, scan_unit.hip): built for key sets of UTF-8-shaped units with mostly multi-byte
  // characters; plain byte-offset matches through the event regions then take one step per character
  UnitImage unit;
  bool unit_ok = false;  // uploaded and usable on the device
  UnitDev udev{};
  const uint32_t *d_unit_end_info = nullptr;
  const uint2 *d_unit_end_chars = nullptr;  // ... with the key's length in characters (char offsets)
  const uint2 *d_unit_end = nullptr;  // fused expansion (scan_unit.hip ku_expand_groups): key, key length, chain offset per END base
  bool unit_fused = false;            // ... usable: flattened chains of at most 15 keys, key lengths below 2^16
  // skip-ahead traversal over the unit image (unit.hpp MARKS, scan_skip.hip): the filter over the two-unit paths on the device
  bool skip_ok = false;
  SkipDev sdev{};
  // pair engine (unit.hpp PAIR TABLE, scan_pair.hip): the two-unit paths as a perfect hash table on the device
  bool pair_ok = false;
  PairDev pdev{};
  std::atomic<uint32_t> pair_off{0};  // batches it gave up (three: the handle stops trying)
  uint32_t seg2 = 0;  // slots below it: the root's and the depth-1 states' rows
  // match_longest only (cedar_replay.cpp): the states that carry one of Cedar's stale END flags, derived on the first
  // match_longest call (it replays every insert: as long again as the rest of compile); dev_longest = dev + the bitmap
  std::vector<uint2> chain_host;     // the flattened output chains {key length, key}
  std::vector<uint32_t> key_info;    // [K] flattened-chain offset | min(chain length, 255) << 24 (empty: no flat chains)
  std::vector<uint32_t> state_base;  // [n_states] base of every state in the image
  std::once_flag stale_once;
  std::vector<uint32_t> stale_states;
  int32_t stale_rc = AHA_OK;
  DevAut dev_longest{};
};

namespace ahai {
#define HIPCHK(ac, call)                                                              \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      tls_err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
      return AHA_E_HIP;                                                               \
    }                                                                                 \
  } while (0)

// Leases one scratch set for the duration of a call: a free one if there is any, a new one while the handle has fewer
// than kMaxScratch, else it waits for the first.
class Lease {
 public:
  explicit Lease(aha_ac *ac) {
    {
      std::lock_guard<std::mutex> lk(ac->pool_mu);
      for (auto &u : ac->pool)
        if (u->mu.try_lock()) {
          sc_ = u.get();
          break;
        }
      if (!sc_ && ac->pool.size() < kMaxScratch) {
        ac->pool.emplace_back(new Scratch());
        sc_ = ac->pool.back().get();
        sc_->mu.lock();
      }
      if (!sc_) wait_ = ac->pool[0].get();
    }
    if (!sc_) {
      wait_->mu.lock();
      sc_ = wait_;
    }
  }
  ~Lease() { sc_->mu.unlock(); }
  Lease(const Lease &) = delete;
  Lease &operator=(const Lease &) = delete;
  Scratch *get() const { return sc_; }

 private:
  Scratch *sc_ = nullptr;
  Scratch *wait_ = nullptr;
};

void free_scratch(Scratch *sc, bool all);
uint64_t scratch_bytes(const Scratch *sc);
// adds the passes that were thrown away to the timing the last pass published (profiling on)
void note_repeats(aha_ac *ac, uint32_t repeats);
void publish_timing(aha_ac *ac, const aha_timing &t);

template <typename T>
int32_t upload(aha_ac *ac, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
  HIPCHK(ac, hipMalloc(&d, bytes));
  ac->dev_allocs.push_back(d);
  if (!v.empty()) HIPCHK(ac, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  ac->image_bytes += v.size() * sizeof(T);
  *out = reinterpret_cast<const T *>(d);
  return AHA_OK;
}

// upload() for tables that appear after compile (the first match_longest call), possibly while other threads match on the
// handle: the copy goes over a private non-blocking stream (no call of the library touches the NULL stream), the handle's
// allocation list is touched under the pool mutex.
template <class T>
int32_t upload_late(aha_ac *ac, const std::vector<T> &v, const T **out) {
  void *d = nullptr;
  const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
  HIPCHK(ac, hipMalloc(&d, bytes));
  {
    std::lock_guard<std::mutex> lk(ac->pool_mu);
    ac->dev_allocs.push_back(d);
    ac->image_bytes += v.size() * sizeof(T);
  }
  if (!v.empty()) {
    hipStream_t st = nullptr;
    HIPCHK(ac, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    HIPCHK(ac, e);
  }
  *out = reinterpret_cast<const T *>(d);
  return AHA_OK;
}

int32_t upload_image(aha_ac *ac, const Image &img);
int32_t ensure_scratch(aha_ac *ac, Scratch *sc, uint64_t n_chunks, uint64_t n_blocks, uint64_t n_docs);
int32_t fill_params(aha_ac *ac, const aha_match_params *p, MatchArgs &M, int *longest);

struct DeviceGuard {
  int prev = -1;
  bool active = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
      active = hipSetDevice(dev) == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (active) (void)hipSetDevice(prev);
  }
};

int32_t ensure_stale(aha_ac *ac);  // match_longest: the states with one of Cedar's stale END flags, derived on first use (capi.cpp)

// ---- engine.cpp
bool skip_eligible(const aha_ac *ac);
bool pair_eligible(const aha_ac *ac);
void plan_engine(aha_ac *ac, const Placement &pl);  // host-only plan: how much of the image the byte-level traversal keeps in LDS
void v2_setup(aha_ac *ac);                          // once per device handle: kernels' LDS limits, grids, the engines' device tables
StreamFmt stream_fmt(const aha_ac *ac);             // field widths of the 4-byte exchange stream for this automaton
int32_t ready_events(aha_ac *ac, Scratch *sc);      // the events of a scratch set are created by the first profiled call that leases it
// the hits as the 4-byte exchange stream as well (aha_ac_match_batch_device_stream); null: not asked for
struct PackOut {
  uint32_t *d_words;
  uint64_t cap_words;
  uint64_t *d_n_words;
};
// one device-resident batch through whichever engine takes it (retries, hand-backs, the two-pass engine as the last resort)
int32_t device_match(aha_ac *ac, Scratch *sc, const uint8_t *d_corpus, const uint64_t *d_doc_offsets, uint64_t n_docs,
                     uint64_t n_bytes, const aha_match_params *params, aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets,
                     uint64_t *n_hits, void *stream, bool offsets_checked, const PackOut *pk, bool *packed);
}  // namespace ahai
