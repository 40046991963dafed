// image.hpp -- structures shared by the host C ABI (capi.cpp) and the HIP
// kernels (kernels.hip).  Plain data, passed to kernels by value.
#pragma once
#include <cstdint>

#include "../../include/aha_hip.h"

namespace aha {

// Automaton image resident in HBM (see automaton.hpp for the slot formats).
struct DevAut {
  const void *slots;         // uint2[n_slots] (wide) or uint32[n_slots] (compact)
  const int32_t *end_key;    // compact only: key id at header slots of end states
  const uint2 *key_ln;       // [K] {len, next}: ac.cr key_lens / output.next chain
  const uint32_t *key_cnt;   // [K] hits emitted when the key's state is reached
  const uint32_t *key_kc;    // [K] lead bytes in key[1..len)
  uint32_t root;
  uint32_t n_slots;
  uint32_t max_len;
  uint32_t compact;
};

struct MatchArgs {
  const uint8_t *text;
  const uint64_t *doc_off;   // [D+1]
  uint64_t n_docs;
  uint64_t n_bytes;
  uint64_t n_chunks;
  uint32_t chunk;            // bytes per chunk (multiple of 16)
  int32_t chars;             // 1: String overload, char offsets (matcher.cr:34-39)
  int32_t sep;               // 1: match(seq, sep) (ac.cr:321-340)
  uint32_t sep_block[8];     // bit c set <=> (c < sep.size && !sep[c])
  // scratch
  uint32_t *counts;          // [n_chunks] hits per chunk
  uint32_t *leads;           // [n_chunks] UTF-8 lead bytes per chunk (chars mode)
  uint64_t *blk_hits;        // [n_blocks] per-block sums, then exclusive bases
  uint64_t *blk_leads;       // [n_blocks]
  uint64_t *docg;            // [D+1] absolute lead-byte count at each doc start
  uint64_t *totals;          // [2] total hits, total leads
  // outputs
  aha_hit *out;
  uint64_t cap;
  uint64_t *doc_hit_off;     // [D+1] or null
};

constexpr int kBlock = 256;  // threads per block in the traversal kernels

struct LaunchCfg {
  void *stream;
};

// launchers (kernels.hip)
void launch_count(const DevAut &A, const MatchArgs &M, void *stream);
void launch_scan_blocks(const MatchArgs &M, uint64_t n_blocks, void *stream);
void launch_docg(const MatchArgs &M, void *stream);
void launch_write(const DevAut &A, const MatchArgs &M, void *stream);

}  // namespace aha
