// internal.hpp -- what capi.cpp and group.cpp share beside the C ABI (library-internal, not exported).
#pragma once
#include <condition_variable>
#include <cstdint>
#include <mutex>

#include "../../include/aha_hip.h"

// A shard of a group on a device it shares with other shards (group.cpp): where its hits go in the caller's host buffer
// becomes known while its call runs -- when the shards before it have counted (`ready`) -- and the next shard starts when
// this one's text is on the device (`uploads_done`).  Waiters sleep on the condition variable (they used to spin on two
// volatile ints, a core each for the whole phase).
struct aha_internal_host_copy {
  aha_hit *out = nullptr;  // where this call's hits go in host memory (null: nowhere)
  uint64_t cap = 0;        // room there, in hits
  std::mutex mu;
  std::condition_variable cv;
  bool ready = false;         // out / cap are final
  bool uploads_done = false;  // the call's text is on the device (or the call is over)
  void set_ready(aha_hit *o, uint64_t c) {
    {
      std::lock_guard<std::mutex> lk(mu);
      out = o;
      cap = c;
      ready = true;
    }
    cv.notify_all();
  }
  void wait_ready() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return ready; });
  }
  void set_uploads_done() {
    {
      std::lock_guard<std::mutex> lk(mu);
      uploads_done = true;
    }
    cv.notify_all();
  }
  void wait_uploads_done() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return uploads_done; });
  }
};

// aha_ac_match_batch_keep that also copies the ranges' hits to host memory while they fit (hc may be null)
extern "C" int32_t aha_internal_match_batch_keep_copy(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets,
                                                      uint64_t n_docs, const aha_match_params *params, aha_hit *d_hits,
                                                      uint64_t cap, aha_internal_host_copy *hc, uint64_t *doc_hit_offsets,
                                                      uint64_t *n_hits);
