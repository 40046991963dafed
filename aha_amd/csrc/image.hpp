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
  const uint32_t *end_info;  // compact only: end_key (or, with flattened chains, its chain offset) | min(key_cnt, 255) << 24
  const uint2 *key_ln;       // [K] {len, next}: ac.cr key_lens / output.next chain
  const uint32_t *key_cnt;   // [K] hits emitted when the key's state is reached
  const uint32_t *key_kc;    // [K] lead bytes in key[1..len)
  // the output chains once more, flattened (null when their total length is unreasonable): key k's chain is
  // chain[chain_off[k] .. + key_cnt[k]) = {len, key} of the key itself and of every key after it on the chain, so the
  // expansion reads consecutive entries instead of chasing key_ln.next (cfg 5: 947 M dependent gathers otherwise)
  // With the flattened chains an event record carries `chain offset | min(chain length, 255) << 24` instead of the key
  // id (k2d_count takes it from end_info / key_info with the one gather it makes anyway); chains need 24-bit offsets.
  const uint32_t *key_info;   // [K] chain_off[k] | min(key_cnt[k], 255) << 24 (wide format: gathered by the count pass)
  const uint2 *chain;         // {len, key}
  const uint2 *chain_chars;   // {lead bytes in key[1..len) + 1 = the key's length in characters, key}, same index (char
                              // offsets: one gather per hit like the byte offsets, not a second one for the length)
  uint32_t root;
  uint32_t n_slots;
  uint32_t max_len;
  uint32_t compact;
  // shadow fail: a state with base in [s2_lo, s2_hi) has depth >= 3 and a fail target of depth <= 2, i.e. its fail
  // target is the depth<=2 state of the last two input bytes, which k2_traverse keeps in registers (0,0 = off)
  uint32_t s1_lo;            // [s1_lo, s2_lo): depth-2 states; their fail target is the depth-1 state of the last byte
  uint32_t s2_lo;
  uint32_t s2_hi;
  // match_longest only: bit B set <=> the state with base B carries one of Cedar's stale END flags (cedar_replay.cpp);
  // null when there is none
  const uint32_t *stale_bits;
  const uint32_t *term_bits;   // match_longest: one bit per slot, set at the base of a state that ends a key AND has children
                               // (its Cedar node keeps the value in a label-0 child, which a NUL byte reaches; kernels.hip)
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
  int32_t has_nul;           // match_longest, chunked form: the batch holds NUL bytes (the warm-ups look for them)
  int32_t no_filter;         // host side: this call has been handed back by the prefix-filter engine (scan_filter.hip)
  int32_t no_pair;           // ... by the pair engine (scan_pair.hip)
  int32_t check_docs;        // host side: the device-resident doc offsets have not been validated yet -- the single-traversal
                             // pipelines do it on the device, in front of the traversal, without a round trip to the host
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

// ---- single-traversal engine (scan_v2.hip) ---------------------------------
constexpr int kV2Threads = 1024;      // one persistent workgroup per CU
constexpr int kV2Piece = 32;          // input bytes staged per lane per round (multiple of 16)
constexpr uint32_t kV2Slab = 512;     // event records a wave reserves per atomic
constexpr uint32_t kV2MaxS = 32768;   // super-chunk bytes per lane (rel/seq fit 15/16 bits)

// One record per position whose state ends a key (fetch is expanded later):
//   x = chunk id, y = seq << 16 | last_byte_of_doc << 15 | rel (pos - chunk start),
//   z = end offset inside the document (bytes), w = key id (wide) / state base (compact)
struct V2Args {
  const uint8_t *text;
  const uint64_t *doc_off;
  uint64_t n_docs;
  uint64_t n_bytes;
  uint64_t n_chunks;
  uint32_t S;                // super-chunk bytes per lane, multiple of 64
  uint32_t lds_slots;        // automaton slots cached in LDS (prefix of the image)
  int32_t chars;
  int32_t sep;
  uint32_t sep_block[8];
  // traversal outputs
  uint4 *ev;                 // [ev_cap] event records in arrival order
  uint32_t *ev_aux;          // [ev_cap] chars mode: lead count << 1 | exact
  uint64_t ev_cap;
  unsigned long long *cursor;  // [0] next free record, [1] overflow flag
  uint32_t *slab_used;       // [ev_cap / kV2Slab + 1] valid records per slab
  uint32_t *ev_cnt;          // [n_chunks] events per chunk
  uint32_t *lead_cnt;        // [n_chunks] chars: lead bytes per chunk
  uint32_t *chunk_doc0;      // [n_chunks] chars: document containing the chunk start
  uint32_t *doc_ev_rank;     // [D+1] events of the chunk before the document start
  uint32_t *doc_lead_rank;   // [D+1] chars: lead bytes of the chunk before the document start
  // post passes
  uint64_t *ev_base;         // [n_chunks] exclusive scan of ev_cnt
  uint64_t *lead_base;       // [n_chunks] exclusive scan of lead_cnt
  uint64_t *blk_a;           // block sums / bases scratch
  uint64_t *blk_b;
  uint4 *sorted_ev;          // [ev_cap] {key, end_b, chunk, y} in final order
  uint32_t *sorted_aux;      // [ev_cap]
  uint32_t *sorted_cnt;      // [ev_cap] hits per event
  uint64_t *totals;          // [0] hits [1] leads [2] events
  // direct event regions (plain mode): chunk c stores its events in order at evd[c * ev_stride + seq], so the
  // post passes need no sort: count (wave per chunk) -> scan -> expand (wave per chunk)
  int32_t direct;
  int32_t dense_hits;        // THIS call's capacity allows for more than one hit per 4 input bytes (capi.cpp match_v2)
  uint32_t ev_stride;        // events a chunk may store (S / 4); more -> overflow flag, slab pipeline instead
  uint2 *evd;                // [n_chunks * ev_stride] {state base (compact) or key id, end offset in the document}
  uint32_t *evg;             // [n_chunks * ev_stride * 3] character-level traversal: the events of 64 chunks (a wave) together,
                             // in the order of the wave's trips: {END state base | lane | min(hits, 255) (unit.hpp u_rec_x / u_rec_z), end offset
                             // in the document, hits of the chunk before the event}
  uint32_t *doc_hit_rank;    // [D+1] character-level traversal: hits of the chunk before the document start (exact when no
                             // event stands for more than 15 hits: the record carries the count in bits 28..31 then)
  uint32_t unit_bb;          // character-level traversal: base width of the image (event records: base | lane << bb | hits << bb + 6)
  uint32_t *chunk_hits;      // [n_chunks]
  uint64_t *hit_base;        // [n_chunks] exclusive scan of chunk_hits
  aha_hit *out;
  uint64_t cap;
  uint64_t *doc_hit_off;
  // region pipelines: where the call's last kernel -- the per-document offsets -- leaves the five words the host reads
  // (cursor[0..1], totals[0..2]: k_publish_words as a launch of its own costs ~5 us of a 64 MiB call); null: nobody publishes
  unsigned long long *publish;
  // the 16 counter words of the NEXT call on this scratch, cleared by the same kernel (engine.cpp: Scratch::cursor_phase); null: nobody clears
  unsigned long long *clear_next;
  // the region pipeline's expansion launch: its blocks from doc_from on compute the documents' hit offsets (0: a launch of their own)
  uint32_t doc_from, doc_lanes16;
};

// ---- character-level engine (scan_unit.hip, unit.hpp) ---------------------------
struct UnitDev {
  const uint2 *slots;      // [n_slots] {lo, hi} entries
  const uint32_t *root;    // [n_syms] the root's transitions by symbol
  const uint32_t *tables;  // [kUTabWords] decode tables
  uint32_t n_slots;
  uint32_t big_lo;         // bases from here on are big states: a private block each (unit.hpp, BIG STATES)
  uint32_t n_low;          // symbols below it index a big state's block directly, the others a group record
  uint32_t g0;             // group record of symbol s: base + g0 + (s >> 5)
  uint32_t base_bits;      // 22 or 23 (unit.hpp, BASE WIDTH)
  uint32_t n_syms;
  uint32_t max_len;        // longest key, bytes
  uint32_t hdr_beside;     // 1: the traversal requests a state's header beside its probe (scan_unit.hip ku_traverse<.., HB>)
};
size_t unit_lds_bytes(uint32_t n_syms);
int unit_prepare(uint32_t n_syms);  // raises the dynamic-LDS limit; hipError_t as int
void unit_launch_traverse(const UnitDev &U, const V2Args &M, uint32_t grid, void *stream);
void unit_launch_regroup(const DevAut &A, const V2Args &M, void *stream);  // evg -> evd + chunk_hits (replaces k2d_count)
// evg + hit_base -> out, doc_hit_off: the whole expansion in one pass over the wave-ordered events.  uend[base] of an END
// state = {key | (key length & 255) << 24, offset of its flattened output chain | (key length >> 8) << 24}
void unit_launch_expand(const uint2 *uend, const DevAut &A, const V2Args &M, uint32_t workgroups, void *stream);
void v2_launch_hit_scan(const V2Args &M, void *stream);   // chunk_hits -> hit_base, totals[0]
void v2_launch_lead_scan(const V2Args &M, void *stream);  // lead_cnt -> lead_base, totals[1] (char offsets)

// ---- skip-ahead traversal over the unit image (scan_skip.hip; unit.hpp, MARKS)
struct SkipDev {
  const uint32_t *bloom;  // [1 << log2] blocked Bloom filter over the two-unit paths, keyed by raw bytes
  uint32_t log2;
  uint32_t k1;            // the second unit's multiplier of the pair hash (UnitImage::pair_k1)
};
int skip_prepare(uint32_t n_syms, uint32_t log2_words);  // raises the dynamic-LDS limits; hipError_t as int
size_t skip_bitmap_bytes(uint64_t n_bytes);             // scratch of a call: one bit per byte position, padded
// ks_mark: text -> bitmap (bit p: a two-unit path may start at p).  ks_traverse: ku_traverse's walk and outputs (evg, ev_cnt,
// chunk_hits, doc_*_rank), started only at marked positions.
void skip_launch_mark(const SkipDev &K, const V2Args &M, void *bitmap, uint32_t grid, void *stream);
void skip_launch_traverse(const UnitDev &U, const V2Args &M, const void *bitmap, uint32_t grid, void *stream);

// ---- pair engine (scan_pair.hip; unit.hpp, PAIR TABLE)
struct PairDev {
  const uint32_t *bloom;   // the marks' filter
  const uint4 *tab;        // [1 << log2] {raw0 | hits << 24, raw1, event payload (0: none), child filter}
  const uint8_t *disp;     // [groups]
  uint32_t bloom_log2, k1, log2, groups;
};

int pair_prepare(uint32_t n_syms, uint32_t bloom_log2, uint32_t groups);  // raises the dynamic-LDS limits; hipError_t as int
uint32_t pair_tile_bytes();                 // a tile of kp_pairs = a chunk of the event regions (V2Args.S)
size_t pair_cand_bytes(uint64_t cap);       // scratch: the deep candidates' list and what kp_walk finds for each
size_t pair_walk_bytes(uint64_t cap);
// tile_dn: n_chunks words; V2Args.lead_cnt / chunk_doc0 (free with byte offsets): a tile's first candidate and their number;
// cursor[7]: the list's length.  mid_event (nullable): recorded behind kp_pairs.
void pair_launch(const PairDev &P, const UnitDev &U, const DevAut &A, const V2Args &M, void *tile_dn, void *cand, void *wres,
                 uint64_t cand_cap, uint32_t grid, void *mid_event, void *stream);

// ---- prefix-filter engine (scan_filter.hip): byte-level, for batches where few positions can start a key
constexpr uint32_t kFilterMul = 0x9E3779B1u;  // an entry of the filter: scan_filter.hip kf_filter, capi.cpp filter_entry
constexpr uint32_t kFilterLog2 = 14;  // 2^14 words = 64 KiB: blocked Bloom filter over the keys' first D bytes
struct FilterDev {
  const uint32_t *bloom;
  uint32_t d;     // bytes of a key the filter looks at: min(4, shortest key)
  uint32_t log2;  // the filter has 2^log2 words (10 .. kFilterLog2)
};
// kf_walk keeps the image in LDS, beside its 16 waves' candidate lists, when both fit (chunks of 32 KiB: 57 KiB of image, 4 KiB: 115)
bool filter_image_in_lds(uint32_t n_slots, uint32_t chunk_bytes, bool chars = false);
int filter_prepare();  // once per process, before the first launch (LDS beyond 64 KiB is opt-in)
// filter_launch_filter: bitmap (one bit per byte position of M.text) and -- on the same launch -- the chunk records of chunks
// of M.S bytes (4, 8, 16 or 32 KiB; chunk_rec: n_chunks * filter_chunk_rec_bytes() of scratch).  non_ascii (nullable): set
// to 1 when the batch holds a byte >= 0x80; kf_walk then hands the call back (cursor[1] = 3).
// filter_launch_walk: bitmap + records -> evd / ev_cnt / doc_ev_rank.
size_t filter_chunk_rec_bytes();
void filter_launch_filter(const FilterDev &F, const V2Args &M, void *bitmap, void *chunk_rec, unsigned long long *non_ascii,
                          uint32_t cus, void *stream);
void filter_launch_walk(const DevAut &A, const V2Args &M, const void *bitmap, const void *chunk_rec, const unsigned long long *non_ascii,
                        uint32_t cus, void *stream);

size_t v2_lds_bytes(uint32_t lds_slots, bool compact);
int v2_prepare(bool compact, size_t lds_bytes);  // raises the dynamic-LDS limit; hipError_t as int
void v2_launch_traverse(const DevAut &A, const V2Args &M, uint32_t grid, void *stream);
// scans ev_cnt (and lead_cnt) into ev_base / lead_base; totals[2] = events, totals[1] = leads
void v2_launch_chunk_scan(const V2Args &M, void *stream);
void v2_launch_sort(const DevAut &A, const V2Args &M, uint64_t n_records, void *stream);
void v2_launch_expand(const DevAut &A, const V2Args &M, uint64_t n_events, void *stream);
// direct pipeline (plain mode): per-chunk hit counts, their scan, expansion and document offsets
void v2_launch_direct_post(const DevAut &A, const V2Args &M, void *stream, void *ev_mid, bool counted = false);

// exchange format of the multi-GPU all-gatherv (kernels.hip): {end, value} pairs <-> Hit triples
void launch_hits_pack(const int32_t *hits, uint64_t n, int32_t *pairs, void *stream);
void launch_hits_unpack(const DevAut &A, const int32_t *pairs, uint64_t n, int chars, int32_t *hits, void *stream);
// 4-byte exchange stream (kernels.hip): stream_words holds n + ceil(n/1024) + exceptions words (capacity 2n + ceil(n/1024))
// field widths of the 4-byte exchange stream's words (kernels.hip): value << (step_bits + len_bits) | len << step_bits | step
struct StreamFmt {
  uint32_t step_bits;  // 6..12; 2^step_bits - 1 = exception
  uint32_t len_bits;   // 0: the key's length is not carried (looked up on arrival)
};
void launch_hits_pack4(const int32_t *hits, uint64_t n, uint32_t *stream_words, unsigned long long *n_words, StreamFmt F,
                       void *stream);
void launch_hits_unpack4(const DevAut &A, const uint32_t *stream_words, uint64_t n, int chars, int32_t *hits, StreamFmt F,
                         void *stream);
// several streams in one launch: stream k starts at word word_off[k] of `land`, holds n_hits[k] hits, goes to hits[out_off[k]..]
constexpr uint32_t kMaxSegs = 64;
void launch_hits_unpack4_segs(const DevAut &A, const uint32_t *land, const uint64_t *word_off, const uint64_t *n_hits,
                              const uint64_t *out_off, uint32_t n_segs, int chars, int32_t *hits, StreamFmt F, void *stream);

// flag[0] |= 1: not the offsets of n_docs documents over n_bytes; |= 2: a document of 2^31 bytes or more
void launch_publish_words(const unsigned long long *src, unsigned long long *host_dst, int n, void *stream);
void launch_check_docs(const uint64_t *doc_off, uint64_t n_docs, uint64_t n_bytes, uint32_t *flag, unsigned long long *abort_word,
                       void *stream);

// launchers (kernels.hip)
void launch_count(const DevAut &A, const MatchArgs &M, void *stream);
void launch_scan_blocks(const MatchArgs &M, uint64_t n_blocks, void *stream);
void launch_docg(const MatchArgs &M, void *stream);
void launch_write(const DevAut &A, const MatchArgs &M, void *stream);
// match_longest (ac.cr:118-143): mode 1 = intersectable false (a thread per document), 2 = true (a thread per chunk,
// byte offsets), 3 = true with a thread per document (char offsets)
void launch_has_nul(const uint8_t *text, uint64_t n, uint64_t *flag, void *stream);  // *flag = 1 when the text holds a NUL byte
void launch_longest(const DevAut &A, const MatchArgs &M, int mode, bool write, void *stream);

}  // namespace aha
