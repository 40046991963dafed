// group.cpp -- aha_group_*: one batch over several GPUs of one node, behind the C ABI (include/aha_hip.h).
//
// SURVEY.md section 8 e: documents are independent (src/aha/ac.cr:177, the state is per sequence), so the batch is
// cut into contiguous, byte-balanced document ranges, one per device (global hit order = range order: no merge);
// the automaton is replicated; every device matches its range with the single-device path; the hit buffers are
// exchanged with an all-gatherv so that every device ends up with the whole ordered hit stream.  Between distinct
// devices the exchange is RCCL (ncclGroupStart / all-pairs ncclSend + ncclRecv / ncclGroupEnd: every device drives
// its xGMI links at once -- a ring would be bound by one link); RCCL is loaded with dlopen on first use so that
// single-GPU users do not pay for it.  Entries of the device list that name the SAME device (several shards on one
// GPU, each with its own handle and stream) exchange with device-to-device copies instead: that is the form the
// 1-GPU development box can execute.  AHA_GROUP_RCCL=self (read by aha_group_compile) makes such a group send every
// shard's OWN stream to itself through RCCL -- one communicator of one rank per shard, a grouped ncclSend/ncclRecv
// pair through the same payload()/landing() code -- so that the library loading, the symbol signatures, the datatype
// constant, ncclCommInitAll and the stream ordering behind the pack kernels are executed on a 1-GPU box; the peers'
// streams still arrive by copies.  What only a node with several GPUs can exercise is the topology (n > 1 ranks).
// Calls on one group are serialised (a mutex); a worker thread never lets an exception cross the C boundary.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aha_hip.h"

#include "internal.hpp"

namespace {

// the few RCCL entry points used (rccl.h declares them; resolved at run time)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclInt32 = 2;  // ncclDataType_t::ncclInt32
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  bool load(std::string &err) {
    if (lib) return true;
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      err = std::string("librccl.so: ") + dlerror();
      return false;
    }
#define AHA_SYM(field, name)                                  \
  field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
  if (!field) {                                               \
    err = std::string("librccl.so lacks ") + name;            \
    return false;                                             \
  }
    AHA_SYM(CommInitAll, "ncclCommInitAll")
    AHA_SYM(CommDestroy, "ncclCommDestroy")
    AHA_SYM(GroupStart, "ncclGroupStart")
    AHA_SYM(GroupEnd, "ncclGroupEnd")
    AHA_SYM(Send, "ncclSend")
    AHA_SYM(Recv, "ncclRecv")
#undef AHA_SYM
    return true;
  }
};

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  bool reserve(size_t n) {
    if (bytes >= n) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    const size_t want = n + n / 8 + 256;
    if (hipMalloc(&p, want) != hipSuccess) return false;
    bytes = want;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

struct Shard {
  aha_ac *ac = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t dstream = nullptr;  // the shard's download of its own hits (beside the exchange on `stream`)
  DevBuf corpus, doc, dho, out, all;
  DevBuf pk, land, nw;  // 4-byte exchange stream: packed own hits, landing area of the peers' streams, stream length
  // per call
  uint64_t d0 = 0, d1 = 0, n_hits = 0, n_words = 0;
  std::string err;
  std::vector<uint64_t> h_dho;
  int32_t rc = AHA_OK;
  double ms_match = 0;
  aha_internal_host_copy *hc = nullptr;  // shards in turn on a shared device: this shard's place in the caller's buffer
};

double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

struct aha_group {
  std::vector<Shard> shards;
  bool distinct = true;   // no device named twice: the exchange is RCCL
  bool self_rccl = false;  // AHA_GROUP_RCCL=self: every shard's own stream travels through RCCL (rehearsal on one GPU)
  Rccl rccl;
  std::vector<ncclComm_t> comms;  // distinct: rank r of ONE communicator of n ranks; self: n communicators of one rank
  aha_group_timing last{};
  std::string err;
  std::mutex mu;  // calls on one group are serialised
  uint64_t gathered = 0;  // hits every shard's `all` buffer holds since the last successful exchange
};

// The all-gatherv of the hit buffers (SURVEY.md section 8 e): every shard holds its own hits in `out` (and, words: their
// 4-byte stream in `pk`, its length in `nw`, packed behind its match); afterwards every shard's `all` holds the whole ordered
// stream.  base[r] = hits before shard r, total = all of them.  Fills T.ms_exchange / exchange / packed / wire_bytes.
static int32_t group_exchange(aha_group *g, const aha_match_params *params, bool words, int via_i, const std::vector<uint64_t> &base,
                              uint64_t total) {
  enum Transport { kCopies = 0, kRccl = 1, kSelfRccl = 2 };
  const Transport via = (Transport)via_i;
  const size_t n = g->shards.size();
  aha_group_timing &T = g->last;
  auto fail = [g](int32_t rc, const std::string &what) {
    g->err = what;
    return rc;
  };
  const auto t_x = std::chrono::steady_clock::now();
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    if (hipSetDevice(s.device) != hipSuccess || !s.all.reserve(std::max<uint64_t>(total, 1) * sizeof(aha_hit)))
      return fail(AHA_E_HIP, "hipMalloc failed for the gathered hits");
  }
  const int chars = (params && params->char_offsets) ? 1 : 0;
  std::vector<uint64_t> ebase(n + 1, 0);  // in 32-bit elements
  if (words) {  // (every shard has packed its hits behind its match: work() above)
    for (size_t r = 0; r < n; r++) {
      Shard &s = g->shards[r];
      if (hipSetDevice(s.device) != hipSuccess ||
          hipMemcpyAsync(&s.n_words, s.nw.p, 8, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
          hipStreamSynchronize(s.stream) != hipSuccess)
        return fail(AHA_E_HIP, "reading the stream length failed");
    }
  } else {
    for (size_t r = 0; r < n; r++) g->shards[r].n_words = g->shards[r].n_hits * 3;
  }
  for (size_t r = 0; r < n; r++) ebase[r + 1] = ebase[r] + g->shards[r].n_words;
  T.wire_bytes = 4 * ebase[n];
  if (words)
    for (size_t r = 0; r < n; r++) {
      Shard &s = g->shards[r];
      if (hipSetDevice(s.device) != hipSuccess || !s.land.reserve(std::max<uint64_t>(ebase[n], 1) * 4))
        return fail(AHA_E_HIP, "hipMalloc failed for the landing area");
    }
  // payload of shard p as shard r sees it arrive: triples land in place, words in the landing area
  auto landing = [&](Shard &s, size_t p) -> void * {
    return words ? (void *)((uint32_t *)s.land.p + ebase[p]) : (void *)((aha_hit *)s.all.p + base[p]);
  };
  auto payload = [&](const Shard &q) -> const void * { return words ? q.pk.p : q.out.p; };
  if (via != kCopies) {
    if (g->comms.empty()) {
      if (!g->rccl.load(g->err)) return AHA_E_HIP;
      g->comms.assign(n, nullptr);
      bool ok = true;
      if (via == kRccl) {  // one communicator, rank r on devices[r]
        std::vector<int> devs;
        for (const Shard &s : g->shards) devs.push_back(s.device);
        ok = g->rccl.CommInitAll(g->comms.data(), (int)n, devs.data()) == 0;
      } else {  // rehearsal: one communicator of ONE rank per shard
        for (size_t r = 0; r < n && ok; r++) {
          const int dev = g->shards[r].device;
          ok = g->rccl.CommInitAll(&g->comms[r], 1, &dev) == 0;
        }
      }
      if (!ok) {
        for (ncclComm_t c : g->comms)
          if (c) (void)g->rccl.CommDestroy(c);
        g->comms.clear();
        return fail(AHA_E_HIP, "ncclCommInitAll failed");
      }
    }
    bool ok = g->rccl.GroupStart() == 0;
    for (size_t r = 0; r < n && ok; r++) {
      Shard &s = g->shards[r];
      ok = hipSetDevice(s.device) == hipSuccess;  // the calls of rank r are made with its device current
      for (size_t p = 0; p < n && ok; p++) {
        if (via == kRccl ? p == r : p != r) continue;  // real exchange: every peer; rehearsal: the own stream only
        const int peer = via == kRccl ? (int)p : 0;
        if (s.n_words) ok = ok && g->rccl.Send(payload(s), s.n_words, kNcclInt32, peer, g->comms[r], s.stream) == 0;
        if (g->shards[p].n_words)
          ok = ok && g->rccl.Recv(landing(s, p), g->shards[p].n_words, kNcclInt32, peer, g->comms[r], s.stream) == 0;
      }
    }
    ok = (g->rccl.GroupEnd() == 0) && ok;
    if (!ok) return fail(AHA_E_HIP, "RCCL send/recv failed");
  }
  if (via != kRccl) {  // shards on one device: the peers' payloads by device-to-device copies
    for (size_t r = 0; r < n; r++) {
      Shard &s = g->shards[r];
      if (hipSetDevice(s.device) != hipSuccess) return fail(AHA_E_HIP, "hipSetDevice failed");
      for (size_t p = 0; p < n; p++) {
        const Shard &q = g->shards[p];
        if (p != r && q.n_words &&
            hipMemcpyAsync(landing(s, p), payload(q), q.n_words * 4, hipMemcpyDeviceToDevice, s.stream) != hipSuccess)
          return fail(AHA_E_HIP, "device-to-device copy failed");
      }
    }
  }
  // the own part as it is (unless it came back through RCCL: then it is rebuilt like a peer's, so the test of the
  // rehearsal checks the bytes RCCL delivered); all streams that arrived are rebuilt into triples by ONE launch
  const bool own_via_rccl = via == kSelfRccl;
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    if (hipSetDevice(s.device) != hipSuccess) return fail(AHA_E_HIP, "hipSetDevice failed");
    if (s.n_hits && !own_via_rccl &&
        hipMemcpyAsync((aha_hit *)s.all.p + base[r], s.out.p, s.n_hits * sizeof(aha_hit), hipMemcpyDeviceToDevice,
                       s.stream) != hipSuccess)
      return fail(AHA_E_HIP, "device-to-device copy failed");
    if (!words) continue;  // triples landed in place
    std::vector<aha_stream_seg> segs;
    for (size_t p = 0; p < n; p++) {
      if ((p == r && !own_via_rccl) || !g->shards[p].n_hits) continue;
      segs.push_back(aha_stream_seg{ebase[p], g->shards[p].n_hits, base[p]});
    }
    for (size_t k = 0; k < segs.size(); k += 64) {
      int32_t rc = aha_ac_hits_unpack4_segs_device(s.ac, (const uint32_t *)s.land.p, segs.data() + k,
                                                   (uint32_t)std::min<size_t>(64, segs.size() - k), chars,
                                                   (aha_hit *)s.all.p, s.stream);
      if (rc != AHA_OK) return fail(rc, std::string("rebuild: ") + aha_last_error(s.ac));
    }
  }
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    if (hipSetDevice(s.device) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess)
      return fail(AHA_E_HIP, "exchange failed");
  }
  T.ms_exchange = (float)ms_since(t_x);
  T.exchange = (uint32_t)via;
  T.packed = words ? 1u : 0u;
  g->gathered = total;
  return AHA_OK;
}

extern "C" {

int32_t aha_group_compile(const uint8_t *key_bytes, const uint64_t *key_offsets, uint32_t n_keys, const int32_t *devices,
                          int32_t n_devices, uint32_t flags, aha_group **out, uint32_t *err_key) {
  if (!out || !devices || n_devices <= 0 || n_devices > 64) return AHA_E_INVALID;
  *out = nullptr;
  aha_group *g = new aha_group();
  g->shards.resize((size_t)n_devices);
  const char *rc_env = getenv("AHA_GROUP_RCCL");
  g->self_rccl = rc_env && strcmp(rc_env, "self") == 0;
  for (int32_t r = 0; r < n_devices; r++) {
    for (int32_t q = 0; q < r; q++)
      if (devices[q] == devices[r]) g->distinct = false;
    aha_options o{};
    o.struct_size = sizeof(o);
    o.device = devices[r];
    o.flags = flags;
    Shard &s = g->shards[(size_t)r];
    s.device = devices[r];
    // the automaton is replicated: compiled ONCE (the first shard), the others copy its host image and upload it
    int32_t rc = r == 0 || (flags & AHA_OPT_HOST_ONLY) ? aha_ac_compile(key_bytes, key_offsets, n_keys, &o, &s.ac, err_key)
                                                        : aha_ac_replicate(g->shards[0].ac, devices[r], &s.ac);
    if (rc != AHA_OK) {
      aha_group_free(g);
      return rc;
    }
    if (!(flags & AHA_OPT_HOST_ONLY)) {
      if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
        aha_group_free(g);
        return AHA_E_HIP;
      }
    }
  }
  *out = g;
  return AHA_OK;
}

void aha_group_free(aha_group *g) {
  if (!g) return;
  for (ncclComm_t c : g->comms)
    if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
  for (Shard &s : g->shards) {
    if (s.stream || s.corpus.p || s.out.p || s.all.p) (void)hipSetDevice(s.device);
    s.corpus.release();
    s.doc.release();
    s.dho.release();
    s.out.release();
    s.all.release();
    s.pk.release();
    s.land.release();
    s.nw.release();
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.dstream) (void)hipStreamDestroy(s.dstream);
    if (s.ac) aha_ac_free(s.ac);
  }
  delete g;
}

int32_t aha_group_size(const aha_group *g) { return g ? (int32_t)g->shards.size() : AHA_E_INVALID; }
const char *aha_group_last_error(const aha_group *g) { return g ? g->err.c_str() : ""; }

int32_t aha_group_partition(const uint64_t *doc_offsets, uint64_t n_docs, int32_t n_parts, uint64_t *bounds) {
  // contiguous document ranges balanced by bytes: bounds[r] = the document boundary nearest to r * N / n_parts
  if (!doc_offsets || !bounds || n_parts <= 0) return AHA_E_INVALID;
  const uint64_t N = doc_offsets[n_docs];
  bounds[0] = 0;
  for (int32_t r = 1; r < n_parts; r++) {
    const uint64_t target = (uint64_t)(((__uint128_t)N * (uint64_t)r) / (uint64_t)n_parts);
    const uint64_t *lo = std::lower_bound(doc_offsets, doc_offsets + n_docs + 1, target);
    uint64_t d = (uint64_t)(lo - doc_offsets);
    if (d > 0 && (d > n_docs || target - doc_offsets[d - 1] < doc_offsets[d] - target)) d--;
    bounds[r] = std::min(std::max(d, bounds[r - 1]), n_docs);
  }
  bounds[n_parts] = n_docs;
  return AHA_OK;
}

int32_t aha_group_match_batch(aha_group *g, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                              const aha_match_params *params, aha_hit *out, uint64_t cap, uint64_t *doc_hit_offsets,
                              uint64_t *n_hits) {
  if (!g || !doc_offsets || !n_hits) return AHA_E_INVALID;
  if (doc_offsets[0] != 0) return AHA_E_INVALID;
  for (uint64_t d = 0; d < n_docs; d++) {
    if (doc_offsets[d + 1] < doc_offsets[d]) return AHA_E_INVALID;
    if (doc_offsets[d + 1] - doc_offsets[d] >= 0x7FFFFFFFull) return AHA_E_TOO_LONG;
  }
  const uint64_t N = doc_offsets[n_docs];
  if ((N && !corpus) || (cap && !out)) return AHA_E_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  auto fail = [g](int32_t rc, const std::string &what) {
    g->err = what;
    return rc;
  };
  g->gathered = 0;
  const size_t n = g->shards.size();
  *n_hits = 0;
  std::vector<uint64_t> bounds(n + 1);
  (void)aha_group_partition(doc_offsets, n_docs, (int32_t)n, bounds.data());
  aha_group_timing &T = g->last;
  memset(&T, 0, sizeof(T));
  T.struct_size = sizeof(T);
  T.n_devices = (uint32_t)n;

  enum Transport { kCopies = 0, kRccl = 1, kSelfRccl = 2 };
  const Transport via = (g->distinct && n > 1) ? kRccl : (g->self_rccl ? kSelfRccl : kCopies);
  // What travels: the 4-byte stream of include/aha_hip.h (aha_ac_hits_pack4_device) when the key ids fit 20 bits,
  // else the 12-byte triples.  Same code for every transport, so the packed path runs on a one-GPU box too.
  aha_ac_info_t info{};
  info.struct_size = sizeof(info);
  (void)aha_ac_info(g->shards[0].ac, &info);
  // (the 4-byte stream's own condition, aha_ac_hits_pack4_device: ids beyond 2^20 need the length in the word)
  uint32_t sf_step = 0, sf_len = 0;
  const bool fits = aha_ac_stream_format(g->shards[0].ac, &sf_step, &sf_len) == AHA_OK && (sf_len != 0 || info.n_keys <= (1u << 20));
  const bool words = (n > 1 || via == kSelfRccl) && fits;

  // ---- every shard: upload its range, match on its device (one host thread per shard: the calls block)
  const auto t_all = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  auto work = [&](size_t r) {
    Shard &s = g->shards[r];
    s.rc = AHA_OK;
    s.n_hits = 0;
    s.err.clear();
    const uint64_t D = s.d1 - s.d0, b0 = doc_offsets[s.d0], nb = doc_offsets[s.d1] - b0;
    s.h_dho.assign(D + 1, 0);
    if (hipSetDevice(s.device) != hipSuccess) {
      s.rc = AHA_E_HIP;
      s.err = "hipSetDevice failed";
      return;
    }
    std::vector<uint64_t> rel(D + 1);
    for (uint64_t d = 0; d <= D; d++) rel[d] = doc_offsets[s.d0 + d] - b0;
    // room for the shard's share of the caller's capacity plus half; never more than one hit per 4 input bytes up
    // front (a denser shard reports its exact count and is matched once more)
    uint64_t want = N ? (uint64_t)((__uint128_t)cap * nb / N) * 3 / 2 + 1024 : 1024;
    want = std::max<uint64_t>(1024, std::min<uint64_t>(want, nb / 4 + 4096));
    if (!s.out.reserve(want * sizeof(aha_hit))) {
      s.rc = AHA_E_HIP;
      s.err = "hipMalloc failed for the shard's buffers";
      return;
    }
    // the shard's range goes through the handle's pipelined host entry (upload, match and offsets of ~64 MiB document
    // ranges on three private streams), the hits stay on the device for the exchange
    const auto t0 = std::chrono::steady_clock::now();
    for (int attempt = 0; attempt < 2; attempt++) {
      uint64_t nh = 0;
      // (s.host_out: shards that run one after the other know their place in the caller's buffer -- the hits of a range
      // go there from inside the shard's own pipeline, beside the upload of the next range)
      s.rc = aha_internal_match_batch_keep_copy(s.ac, corpus + b0, rel.data(), D, params, (aha_hit *)s.out.p,
                                                s.out.bytes / sizeof(aha_hit), s.hc, s.h_dho.data(), &nh);
      s.n_hits = nh;
      if (s.rc != AHA_E_CAPACITY) break;
      if (!s.out.reserve(nh * sizeof(aha_hit))) {  // the call told the exact count: once more with room
        s.rc = AHA_E_HIP;
        s.err = "hipMalloc failed for the shard's hits";
        return;
      }
    }
    s.ms_match = ms_since(t0);
    if (s.rc != AHA_OK) {
      s.err = aha_last_error(s.ac);  // the text is the calling thread's: take it along
      return;
    }
    if (words) {  // the shard's exchange stream, launched behind its match (it does not wait for the other shards)
      const uint64_t capw = 2 * s.n_hits + (s.n_hits + 1023) / 1024 + 16;
      if (!s.pk.reserve(capw * 4) || !s.nw.reserve(8)) {
        s.rc = AHA_E_HIP;
        s.err = "hipMalloc failed for the packed stream";
        return;
      }
      s.rc = aha_ac_hits_pack4_device(s.ac, (const aha_hit *)s.out.p, s.n_hits, (uint32_t *)s.pk.p, capw, (uint64_t *)s.nw.p,
                                      s.stream);
      if (s.rc != AHA_OK) s.err = std::string("pack: ") + aha_last_error(s.ac);
    }
  };
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    s.d0 = bounds[r];
    s.d1 = bounds[r + 1];
    s.rc = AHA_E_NOMEM;
    s.err = "worker thread not started";
  }
  // the caller's copy of a shard's hits (a private stream per shard, a thread per copy: the source is pageable memory's peer)
  std::vector<std::thread> dl;
  std::vector<int> dl_rc(n, 0);
  std::vector<char> dl_started(n, 0);
  struct JoinAll {
    std::vector<std::thread> &t;
    ~JoinAll() {
      for (auto &x : t)
        if (x.joinable()) x.join();
    }
  } join_dl{dl};
  auto t_d = std::chrono::steady_clock::now();
  auto start_download = [&](size_t r, uint64_t at) {
    if (!out || !g->shards[r].n_hits) return;
    if (dl.empty()) t_d = std::chrono::steady_clock::now();
    dl_started[r] = 1;
    try {
      dl.emplace_back([&, r, at]() {
        Shard &s = g->shards[r];
        if (hipSetDevice(s.device) != hipSuccess ||
            (!s.dstream && hipStreamCreateWithFlags(&s.dstream, hipStreamNonBlocking) != hipSuccess) ||
            hipMemcpyAsync(out + at, s.out.p, s.n_hits * sizeof(aha_hit), hipMemcpyDeviceToHost, s.dstream) != hipSuccess ||
            hipStreamSynchronize(s.dstream) != hipSuccess)
          dl_rc[r] = 1;
      });
    } catch (...) {
      dl_rc[r] = 1;
    }
  };
  if (g->distinct || n == 1) {
    for (size_t r = 0; r < n; r++) {
      try {
        th.emplace_back([&, r]() {
          try {
            work(r);
          } catch (...) {  // bad_alloc of a host vector: no exception crosses the C boundary (or a thread's top frame)
            g->shards[r].rc = AHA_E_NOMEM;
            g->shards[r].err = "out of host memory";
          }
        });
      } catch (...) {
        break;  // thread creation failed: the shards started so far are joined below, the rest keep AHA_E_NOMEM
      }
    }
    for (auto &t : th) t.join();
  } else {
    // Shards that share a device share its PCIe link: side by side they would only take turns on it, and every shard's
    // place in the caller's buffer would be known when all of them have counted -- the hits' way back would follow the whole
    // match phase.  One after the other, a shard's place is known when it is done, and its hits go to the host beside the
    // next shard's upload (the link is full duplex), as in the single handle's pipeline.
    // A shard starts when the one before it has its text on the device (its last match and the copy of its last hits run
    // beside the next shard's first upload); where its hits go is told when the shards before it have counted (`ready`) --
    // its pipeline only needs that for the copies to the host, which wait for it.
    std::vector<aha_internal_host_copy> hc(n);
    for (size_t r = 0; r < n; r++) g->shards[r].hc = &hc[r];
    for (size_t r = 0; r < n; r++) {
      try {
        th.emplace_back([&, r]() {
          if (r) hc[r - 1].wait_uploads_done();  // (set by the shard's call, however it ends, and once more below)
          try {
            work(r);
          } catch (...) {
            g->shards[r].rc = AHA_E_NOMEM;
            g->shards[r].err = "out of host memory";
          }
          hc[r].set_uploads_done();
        });
      } catch (...) {
        for (size_t q = r; q < n; q++) hc[q].set_uploads_done();
        break;  // thread creation failed: the shards started so far are joined below, the rest keep AHA_E_NOMEM
      }
    }
    uint64_t at = 0;
    bool dead = false;
    for (size_t r = 0; r < th.size(); r++) {
      Shard &s = g->shards[r];
      if (!dead && out && at < cap)
        hc[r].set_ready(out + at, cap - at);
      else
        hc[r].set_ready(nullptr, 0);
      th[r].join();
      if (s.rc != AHA_OK) dead = true;  // (the later shards still run to their end; the call reports this one)
      // (a shard whose hits did not all fit the rest of the caller's buffer: AHA_E_CAPACITY below, nothing is written beyond cap)
      dl_started[r] = 1;
      at += s.n_hits;
    }
    for (size_t r = 0; r < n; r++) g->shards[r].hc = nullptr;
  }
  T.ms_match = (float)ms_since(t_all);
  uint64_t total = 0;
  std::vector<uint64_t> base(n + 1, 0);
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    if (s.rc != AHA_OK) return fail(s.rc, std::string("shard ") + std::to_string(r) + ": " + s.err);
    T.ms_match_max_shard = std::max(T.ms_match_max_shard, (float)s.ms_match);
    base[r] = total;
    total += s.n_hits;
  }
  base[n] = total;
  *n_hits = total;
  T.n_hits = total;
  if (doc_hit_offsets) {
    for (size_t r = 0; r < n; r++) {
      const Shard &s = g->shards[r];
      for (uint64_t d = s.d0; d < s.d1; d++) doc_hit_offsets[d] = base[r] + s.h_dho[d - s.d0];
    }
    doc_hit_offsets[n_docs] = total;
  }
  // the caller's buffer is too small: say so before anything is exchanged (count and offsets are already final)
  if (total > cap) return fail(AHA_E_CAPACITY, "output buffer too small");

  // ---- the caller's copy: every device sends ITS hits to the host over its own PCIe link (a private stream per shard),
  // beside the exchange between the devices -- not one device the whole gathered stream behind it.  (Shards on one device:
  // started above, behind each shard's match.)
  for (size_t r = 0; r < n; r++)
    if (!dl_started[r]) start_download(r, base[r]);

  // ---- all-gatherv of the hit buffers: every device gets the whole ordered stream
  {
    const int32_t xrc = group_exchange(g, params, words, (int)via, base, total);
    if (xrc != AHA_OK) return xrc;
  }
  for (auto &x : dl)
    if (x.joinable()) x.join();
  for (size_t r = 0; r < n; r++)
    if (dl_rc[r]) return fail(AHA_E_HIP, "download of a shard's hits failed");
  T.ms_download = (float)ms_since(t_d);  // (from the first download's start: it runs beside the exchange)
  return AHA_OK;
}

// ---- the batch resident on the devices: what a caller that keeps its corpus in HBM (the metric's "corpus resident") calls

struct aha_group_corpus {
  aha_group *g = nullptr;
  std::vector<aha_corpus *> parts;  // shard r's documents on shard r's device
  std::vector<uint64_t> bounds;     // document bounds of the shards
  uint64_t n_docs = 0, n_bytes = 0;
};

int32_t aha_group_corpus_upload(aha_group *g, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                aha_group_corpus **out) {
  if (!g || !doc_offsets || !out) return AHA_E_INVALID;
  *out = nullptr;
  if (doc_offsets[0] != 0) return AHA_E_INVALID;
  for (uint64_t d = 0; d < n_docs; d++) {
    if (doc_offsets[d + 1] < doc_offsets[d]) return AHA_E_INVALID;
    if (doc_offsets[d + 1] - doc_offsets[d] >= 0x7FFFFFFFull) return AHA_E_TOO_LONG;
  }
  const uint64_t N = doc_offsets[n_docs];
  if (N && !corpus) return AHA_E_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  const size_t n = g->shards.size();
  aha_group_corpus *c = nullptr;
  try {
    c = new aha_group_corpus();
    c->g = g;
    c->n_docs = n_docs;
    c->n_bytes = N;
    c->parts.assign(n, nullptr);
    c->bounds.assign(n + 1, 0);
    (void)aha_group_partition(doc_offsets, n_docs, (int32_t)n, c->bounds.data());
    // every device takes its range over its own PCIe link: a thread per shard (shards that share a device take turns on it)
    std::vector<int32_t> rcs(n, AHA_OK);
    std::vector<std::thread> th;
    auto up = [&](size_t r) {
      const uint64_t d0 = c->bounds[r], d1 = c->bounds[r + 1], b0 = doc_offsets[d0];
      try {
        std::vector<uint64_t> rel(d1 - d0 + 1);
        for (uint64_t d = d0; d <= d1; d++) rel[d - d0] = doc_offsets[d] - b0;
        rcs[r] = aha_corpus_upload(g->shards[r].device, corpus ? corpus + b0 : nullptr, rel.data(), d1 - d0, &c->parts[r]);
      } catch (...) {
        rcs[r] = AHA_E_NOMEM;
      }
    };
    if (g->distinct && n > 1) {
      for (size_t r = 0; r < n; r++) th.emplace_back(up, r);
      for (auto &t : th) t.join();
    } else {
      for (size_t r = 0; r < n; r++) up(r);
    }
    for (size_t r = 0; r < n; r++)
      if (rcs[r] != AHA_OK) {
        g->err = "shard " + std::to_string(r) + ": upload of its documents failed";
        const int32_t rc = rcs[r];
        for (aha_corpus *p : c->parts) aha_corpus_free(p);
        delete c;
        return rc;
      }
  } catch (...) {
    if (c) {
      for (aha_corpus *p : c->parts) aha_corpus_free(p);
      delete c;
    }
    return AHA_E_NOMEM;
  }
  *out = c;
  return AHA_OK;
}

void aha_group_corpus_free(aha_group_corpus *c) {
  if (!c) return;
  for (aha_corpus *p : c->parts) aha_corpus_free(p);
  delete c;
}

int32_t aha_group_match_batch_device(aha_group *g, const aha_group_corpus *c, const aha_match_params *params,
                                     uint64_t *doc_hit_offsets, uint64_t *n_hits) {
  if (!g || !c || !n_hits || c->g != g) return AHA_E_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  auto fail = [g](int32_t rc, const std::string &what) {
    g->err = what;
    return rc;
  };
  g->gathered = 0;
  *n_hits = 0;
  const size_t n = g->shards.size();
  if (c->parts.size() != n) return AHA_E_INVALID;
  aha_group_timing &T = g->last;
  memset(&T, 0, sizeof(T));
  T.struct_size = sizeof(T);
  T.n_devices = (uint32_t)n;
  enum Transport { kCopies = 0, kRccl = 1, kSelfRccl = 2 };
  const Transport via = (g->distinct && n > 1) ? kRccl : (g->self_rccl ? kSelfRccl : kCopies);
  aha_ac_info_t info{};
  info.struct_size = sizeof(info);
  (void)aha_ac_info(g->shards[0].ac, &info);
  uint32_t sf_step = 0, sf_len = 0;
  const bool fits = aha_ac_stream_format(g->shards[0].ac, &sf_step, &sf_len) == AHA_OK && (sf_len != 0 || info.n_keys <= (1u << 20));
  const bool words = (n > 1 || via == kSelfRccl) && fits;

  // ---- every device matches its resident range (a host thread per shard: the calls block), packs its exchange stream
  const auto t_all = std::chrono::steady_clock::now();
  auto work = [&](size_t r) {
    Shard &s = g->shards[r];
    const aha_corpus *part = c->parts[r];
    const uint64_t D = aha_corpus_n_docs(part), nb = aha_corpus_n_bytes(part);
    s.d0 = c->bounds[r];
    s.d1 = c->bounds[r + 1];
    s.rc = AHA_OK;
    s.n_hits = 0;
    s.err.clear();
    if (hipSetDevice(s.device) != hipSuccess) {
      s.rc = AHA_E_HIP;
      s.err = "hipSetDevice failed";
      return;
    }
    // room for one hit per 16 input bytes up front (what is there from earlier calls stays); a denser shard reports its
    // exact count and is matched once more
    const uint64_t want = std::max<uint64_t>(s.out.bytes / sizeof(aha_hit), nb / 16 + 4096);
    if (!s.out.reserve(want * sizeof(aha_hit)) || !s.dho.reserve((D + 1) * 8)) {
      s.rc = AHA_E_HIP;
      s.err = "hipMalloc failed for the shard's buffers";
      return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int attempt = 0; attempt < 2; attempt++) {
      uint64_t nh = 0;
      const uint64_t capo = s.out.bytes / sizeof(aha_hit);
      if (words) {  // the match leaves the shard's exchange stream as well (written by the expansion, no pack pass)
        const uint64_t capw = 2 * capo + (capo + 1023) / 1024 + 16;
        if (!s.pk.reserve(capw * 4) || !s.nw.reserve(8)) {
          s.rc = AHA_E_HIP;
          s.err = "hipMalloc failed for the packed stream";
          return;
        }
        s.rc = aha_ac_match_batch_device_stream(s.ac, aha_corpus_bytes(part), aha_corpus_doc_offsets(part), D, nb, params,
                                                (aha_hit *)s.out.p, capo, (uint64_t *)s.dho.p, &nh, (uint32_t *)s.pk.p, capw,
                                                (uint64_t *)s.nw.p, s.stream);
      } else {
        s.rc = aha_ac_match_batch_device(s.ac, aha_corpus_bytes(part), aha_corpus_doc_offsets(part), D, nb, params, (aha_hit *)s.out.p,
                                         capo, (uint64_t *)s.dho.p, &nh, s.stream);
      }
      s.n_hits = nh;
      if (s.rc != AHA_E_CAPACITY) break;
      if (!s.out.reserve(nh * sizeof(aha_hit))) {
        s.rc = AHA_E_HIP;
        s.err = "hipMalloc failed for the shard's hits";
        return;
      }
    }
    s.ms_match = ms_since(t0);
    if (s.rc != AHA_OK) {
      s.err = aha_last_error(s.ac);
      return;
    }
    if (doc_hit_offsets) {
      s.h_dho.assign(D + 1, 0);
      if (hipMemcpyAsync(s.h_dho.data(), s.dho.p, (D + 1) * 8, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
          hipStreamSynchronize(s.stream) != hipSuccess) {
        s.rc = AHA_E_HIP;
        s.err = "download of the per-document offsets failed";
        return;
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (size_t r = 0; r < n; r++) {
      g->shards[r].rc = AHA_E_NOMEM;
      g->shards[r].err = "worker thread not started";
    }
    for (size_t r = 0; r < n; r++) {
      try {
        th.emplace_back([&, r]() {
          try {
            work(r);
          } catch (...) {
            g->shards[r].rc = AHA_E_NOMEM;
            g->shards[r].err = "out of host memory";
          }
        });
      } catch (...) {
        break;
      }
    }
    for (auto &t : th) t.join();
  }
  T.ms_match = (float)ms_since(t_all);
  uint64_t total = 0;
  std::vector<uint64_t> base(n + 1, 0);
  for (size_t r = 0; r < n; r++) {
    Shard &s = g->shards[r];
    if (s.rc != AHA_OK) return fail(s.rc, std::string("shard ") + std::to_string(r) + ": " + s.err);
    T.ms_match_max_shard = std::max(T.ms_match_max_shard, (float)s.ms_match);
    base[r] = total;
    total += s.n_hits;
  }
  base[n] = total;
  *n_hits = total;
  T.n_hits = total;
  if (doc_hit_offsets) {
    for (size_t r = 0; r < n; r++) {
      const Shard &s = g->shards[r];
      for (uint64_t d = s.d0; d < s.d1; d++) doc_hit_offsets[d] = base[r] + s.h_dho[d - s.d0];
    }
    doc_hit_offsets[c->n_docs] = total;
  }
  return group_exchange(g, params, words, (int)via, base, total);
}

int32_t aha_group_download_shard(aha_group *g, int32_t shard, aha_hit *out, uint64_t cap, uint64_t *n_hits) {
  if (!g || shard < 0 || (size_t)shard >= g->shards.size() || !n_hits) return AHA_E_INVALID;
  std::lock_guard<std::mutex> lk(g->mu);
  *n_hits = g->gathered;
  if (g->gathered > cap) return AHA_E_CAPACITY;
  Shard &s = g->shards[(size_t)shard];
  if (g->gathered && (!out || !s.all.p || hipSetDevice(s.device) != hipSuccess ||
                      hipMemcpy(out, s.all.p, g->gathered * sizeof(aha_hit), hipMemcpyDeviceToHost) != hipSuccess)) {
    g->err = "download of a shard's gathered hits failed";
    return AHA_E_HIP;
  }
  return AHA_OK;
}

int32_t aha_group_last_timing(const aha_group *g, aha_group_timing *t) {
  if (!g || !t) return AHA_E_INVALID;
  *t = g->last;
  t->struct_size = sizeof(*t);
  return AHA_OK;
}

}  // extern "C"
