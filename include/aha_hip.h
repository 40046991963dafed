/*
 * aha_hip.h -- C ABI of libaha_hip.so: MI355X-native Aha::AC#match.
 *
 * Drop-in boundary for ONE path of chenkovsky/aha: batch Aho-Corasick
 * traversal (Aha::AC.compile / #match / Aha::Hit).  The reference has no FFI
 * of its own (pure Crystal); these entry points are what a Crystal
 * `lib LibAhaHip` binding calls (bindings/crystal/aha_hip.cr).
 * Citations are file:line relative to the reference repository root.
 *
 * Conventions: extern "C", plain pointers and sizes, no exceptions across the
 * boundary.  Every function returns an int32 status (AHA_OK = 0, <0 error)
 * unless stated.  The caller owns every buffer it passes; the library never
 * retains caller pointers past return.  The AUTOMATON of a handle is immutable after compile and which pipeline a
 * match call takes is a function of that call alone (its size, its params and its `cap`), never of earlier calls.
 * What a handle does keep between calls is grow-only device scratch (aha_ac_release_scratch frees it,
 * aha_ac_scratch_bytes reports it), organised in sets: a match call leases one set for its duration, so concurrent
 * calls on one handle (one host thread and one stream each) run side by side on up to 8 sets and only then wait.
 * aha_last_error returns the text of the calling thread's last failure.
 *
 * Device scratch of one match call on N input bytes with output capacity `cap` hits (single-traversal engine): the
 * event records.  Region pipeline: 8 bytes each, per chunk (N / 2^18 bytes, at least 192) twice the average the
 * capacity allows for plus 1/64 of the chunk, i.e. 16 bytes per hit of capacity + N/8.  Slab pipeline (capacity
 * below 16 hits per chunk, or a separator filter): 41 bytes per hit of capacity + 80 MB.  One event per input byte
 * (8 N bytes, bounded at 48 GiB) only when `cap` announces more than one hit per 4 input bytes or a chunk
 * overflowed its region (aha_timing.repeats then says that the call ran twice).  Plus ~24 bytes per chunk and 4 bytes per
 * document.  A device corpus that is not 16-byte aligned is first copied into scratch (N bytes).
 * Character-level engine (aha_ac_info_t.unit_enabled; aha_timing.engine = 4): its records are 12 bytes and go straight to
 * the expansion -- 24 bytes per hit of capacity + 3N/16 with the fused expansion (output chains of at most 15 keys), + the
 * 8-byte regions above with the general post passes.
 * Prefix-filter engine (aha_ac_info_t.filter_prefix_bytes; aha_timing.engine = 5): the region pipeline's records in chunks
 * of 4 .. 32 KiB (16 bytes per hit of capacity + N/8 as above) + one bit per input byte (N/8) + 16 bytes per chunk.
 */
#ifndef AHA_HIP_H
#define AHA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: what this header declares is all it exports. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define AHA_ABI_VERSION 8

/* Aha::Hit -- src/aha/matcher.cr:2-11.  Half-open [start,end) offsets
 * relative to the start of the sequence (document); value = key index in
 * compile order (src/aha/ac.cr:64-67). */
typedef struct {
  int32_t start, end, value;
} aha_hit;

typedef struct aha_ac aha_ac;

enum {
  AHA_OK = 0,
  AHA_E_INVALID = -1,     /* bad argument */
  AHA_E_EMPTY_KEY = -2,   /* raise "Cannot insert empty key"        src/aha/cedar.cr:756 */
  AHA_E_ZERO_BYTE = -3,   /* raise "key[pos] is zero"               src/aha/cedar.cr:235 */
  AHA_E_DUP_KEY = -4,     /* raise "key:... appear twice."          src/aha/ac.cr:66     */
  AHA_E_SEP_SIZE = -5,    /* raise "sep BitArray size > 256 ..."    src/aha/ac.cr:322    */
  AHA_E_CAPACITY = -6,    /* out buffer too small; *n_hits = required count */
  AHA_E_NO_DEVICE = -7,   /* no usable HIP device: the product path has NO CPU fallback */
  AHA_E_HIP = -8,         /* a HIP runtime call failed; see aha_last_error */
  AHA_E_TOO_LONG = -9,    /* a sequence is >= 2^31 bytes (Int32 offsets, src/aha/matcher.cr:3-5) */
  AHA_E_NOT_FOUND = -10,  /* key / id lookup miss (IndexError in the reference, src/aha/cedar.cr:830-834) */
  AHA_E_TOO_LARGE = -11,  /* automaton exceeds the device image limits */
  AHA_E_NOMEM = -12       /* host memory exhausted (no exception crosses the boundary) */
};

/* Compile-time options.  Zero-initialise and set struct_size. */
typedef struct {
  uint32_t struct_size;
  int32_t device;        /* HIP device ordinal; -1 = current device */
  uint32_t flags;        /* AHA_OPT_* */
  uint32_t reserved;
} aha_options;

#define AHA_OPT_HOST_ONLY 1u /* build the automaton image but do not touch the GPU (tests of host logic) */
#define AHA_OPT_FORCE_WIDE 2u /* always use the 8-byte slot format (default: compact 4-byte when it fits) */

/* Per-call options mirroring the reference's overloads:
 *   char_offsets = 0: match(seq : Bytes)        src/aha/ac.cr:280-286  (byte offsets)
 *   char_offsets = 1: match(seq : String)       src/aha/matcher.cr:34-39 (char offsets; valid UTF-8)
 *   sep_size > 0    : match(seq, sep : BitArray) src/aha/ac.cr:321-340, matcher.cr:41-46;
 *                     sep_bits is the BitArray, LSB-first, sep_size <= 256. */
typedef struct {
  uint32_t struct_size;
  int32_t char_offsets;
  int32_t sep_size;
  uint8_t sep_bits[32];
  /* match_longest(seq, intersectable) -- src/aha/ac.cr:118-143, 249-263, 297-319: 0 = plain match (default; also when
   * struct_size stops before this field), 1 = match_longest with intersectable = false, 2 = with intersectable = true.
   * Not combinable with a separator filter (the reference has no such overload): AHA_E_INVALID.
   * intersectable = true is chunk-parallel (a 2 * Lmax warm-up makes the pending-end register exact);
   * intersectable = false resets the state after every yield, so a document is walked in order by one thread
   * (documents in parallel) -- correct at any size, fast only for batches of many documents. */
  int32_t longest;
} aha_match_params;

typedef struct {
  uint32_t struct_size;
  uint32_t n_keys;          /* K */
  uint64_t n_states;        /* trie nodes incl. root */
  uint64_t n_slots;         /* double-array slots in the device image */
  uint64_t image_bytes;     /* bytes resident in HBM for the automaton */
  uint32_t max_key_len;     /* Lmax, bytes */
  uint32_t slot_bytes;      /* 4 (compact) or 8 (wide) */
  uint32_t lds_slots;       /* slots of the image cached in LDS by the match kernel */
  int32_t device;           /* device the image lives on, -1 if host only */
  /* Shadow fail links (all 0 = every state has a fail header at slot[base]).  Otherwise only the root and the
   * states with base >= fail_hdr_lo own one; for the others the fail target follows from the last input bytes:
   * base < fail_s1_lo: root; base < fail_s2_lo: the depth-1 state of the last byte; base < fail_hdr_lo: the
   * deepest state of depth <= 2 spelled by the last two bytes. */
  uint32_t fail_s1_lo;
  uint32_t fail_s2_lo;
  uint32_t fail_hdr_lo;
  uint32_t unit_header_beside;  /* ABI 7 (was reserved): 1 = the character-level traversal requests a state's fail header beside
                                 * its probe instead of in a trip of its own -- chosen when the image is compiled, for key sets
                                 * where at least a fifth of the states own a header (text then falls out of deep matches often:
                                 * -8.5 % on BASELINE config 5, +3.5 % on config 3, which keeps the header trip) */
  /* Character-level image (aha_amd/csrc/unit.hpp), built when every key is a sequence of UTF-8-shaped units, at least
   * 30 % of the key bytes lie in multi-byte characters and the keys' characters fit the symbol table (AHA_ENGINE=unit:
   * for every eligible key set).  1 = this handle's matches without a separator filter -- byte or char offsets -- take
   * one step per character instead of one per byte (bit-exact; aha_timing.engine = 4); on a host-only handle: the image
   * was built (aha_ac_export). */
  uint32_t unit_enabled;
  uint32_t unit_slots;          /* 8-byte slots of its double array */
  uint32_t unit_syms;           /* symbols of its dense alphabet (the root's transitions: 4 bytes each, in LDS) */
  uint32_t unit_multi_permille; /* key bytes in multi-byte units, per 1000 */
  /* ABI 6: how the image's big states are laid out (aha_amd/csrc/unit.hpp, BIG STATES) -- what a reader of
   * AHA_IMG_UNIT_SLOTS needs beside the slots: a state with base >= unit_big_lo owns unit_big_block slots; the symbols
   * below unit_n_low index them directly, the others select a group record at base + unit_n_low - unit_n_low / 32 +
   * (symbol >> 5). */
  uint32_t unit_big_lo;
  uint32_t unit_big_block;
  uint32_t unit_n_low;
  uint32_t unit_n_big;
  uint32_t unit_base_bits;      /* 22, or 23 for an image beyond 2^22 slots: width of the base field of a state word; the
                                 * filter takes the bits from there up to bit 28 (7 or 6 of them) */
  uint32_t unit_headers;        /* ABI 7 (was reserved): states of the character-level image that own a fail header (their fail
                                 * state is neither the root nor a one-character state) */
  /* ABI 7: the prefix-filter engine (aha_amd/csrc/scan_filter.hip; aha_timing.engine = 5).  Built for a key set without a
   * character-level image whose keys are 3 .. 64 bytes long (a keyword list): a blocked Bloom filter over the keys' first
   * filter_prefix_bytes bytes (min(4, shortest key); 0 = this handle has none) of filter_words 32-bit words.  Matches without
   * a separator filter -- byte or char offsets -- then look at every text position
   * through the filter and walk the automaton only from the positions it lets through; a batch whose text is dense with
   * such positions (more than about one in twenty) is handed to the single-traversal engine by the call itself
   * (aha_timing.repeats counts it). */
  uint32_t filter_prefix_bytes;
  uint32_t filter_words;
  /* ABI 8: the skip-ahead traversal (aha_amd/csrc/scan_skip.hip; aha_timing.engine = 6) over the character-level image.
   * While the state of src/aha/ac.cr:176-192 is the root or a one-character state, its next state depends on the next two
   * characters alone and -- when no key is a single character -- nothing can be reported, so the walk may jump to the next
   * position where a two-character trie path starts.  A stateless first kernel marks those positions through a blocked Bloom
   * filter over the image's skip_pairs two-character paths (skip_filter_words 32-bit words, keyed by the characters' raw bytes;
   * 0 = this handle has none: not asked for, a one-character key, 23-bit bases, the header requested beside the probe); the
   * second kernel is the character-level traversal, started only at marked positions.  Byte offsets, no separator filter;
   * every other call of the handle keeps engine 4.  OPT-IN (AHA_ENGINE=skip or AHA_SKIP=1 when the handle is compiled): on
   * BASELINE config 3 it is slower than engine 4 (its lanes' scattered text requests), so no key set gets it by default. */
  uint32_t skip_filter_words;
  uint32_t skip_pairs;
  /* ABI 8: the pair engine (aha_amd/csrc/scan_pair.hip; aha_timing.engine = 7).  The same two-character paths behind the same
   * filter, once more as a perfect hash table keyed by the characters' raw bytes (pair_table_log2: log2 of its 16-byte slots, 0 =
   * none; pair_groups displacement bytes; pair_hash_k1 the multiplier of the second character in the pair hash -- chosen so that
   * no two pairs share the 32-bit value; marks and table use the same hash).  A stateless pass over every character resolves
   * the two-character states itself -- one 16-byte load per filter-positive pair -- and writes their events in position order;
   * only the positions where a THIRD character can continue a path (a few per cent) are walked, by a second kernel that voids
   * the events its walks cover and fills its own into slots the first pass left for them.  pair_engine = 1: this handle's
   * matches with byte offsets and without a separator filter run it (no one-character key, at most three END states of three
   * characters or more on one trie path); every other call keeps the engine it had. */
  uint32_t pair_hash_k1;
  uint32_t pair_table_log2;
  uint32_t pair_groups;
  uint32_t pair_engine;
} aha_ac_info_t;

/* Timing of the most recent device match on this handle (HIP events recorded
 * on the launch stream).  Only filled when profiling is enabled. */
typedef struct {
  uint32_t struct_size;
  uint32_t n_kernels;
  float ms_total;           /* first launch -> hits and offsets final in HBM */
  float ms_count;           /* engines 4, 2: the traversal kernel; engine 5: the filter kernel; engine 6: the marking kernel; engine 1:
                             * traversal pass 1 */
  float ms_scan;            /* scans of per-chunk counts (slab pipeline; the region pipelines have them in ms_aux: no event in
                             * between); engine 5: the candidates' walks (chunk records + kf_walk); engine 6: the traversal */
  float ms_write;           /* engine 2: chain expansion + doc offsets; engine 1: traversal pass 2 */
  float ms_aux;             /* engine 2: hits per chunk + scan (regions) or event sort (slabs); engine 1: char-offset prefix pass */
  uint64_t n_chunks;
  uint64_t n_hits;
  uint32_t engine;          /* 7 = pair engine (stateless pair pass + deep walks), 6 = marks + skip-ahead character-level traversal,
                             * 5 = prefix filter + candidate walks,
                             * 4 = character-level traversal, 2 = single-traversal engine, 1 = two-pass engine */
  uint32_t chunk_bytes;     /* bytes per lane chunk */
  /* ABI 6: passes over the batch that were thrown away before this one: 1 when a chunk's event region overflowed -- the
   * batch was denser than `cap` said -- and the match ran once more with full-size regions (the call took about twice
   * the time: give a capacity nearer to the hits to avoid it); +1 when the event temp overflowed and the two-pass engine
   * took over. */
  uint32_t repeats;
  uint32_t reserved;
} aha_timing;

const char *aha_strerror(int32_t code);
/* Message of the last error of the CALLING THREAD (thread-local: calls on one handle may run concurrently; `ac` may be
 * NULL, e.g. after aha_buffer_* / aha_corpus_upload). */
const char *aha_last_error(const aha_ac *ac);
uint32_t aha_abi_version(void);
/* Number of visible HIP devices (0 when there is none / no driver). */
int32_t aha_device_count(void);

/* Aha::AC.compile(keys) -- src/aha/ac.cr:62-112.  Keys are one blob plus K+1
 * offsets.  On AHA_E_EMPTY_KEY / ZERO_BYTE / DUP_KEY *err_key (optional) is
 * the index of the offending key (the index at which the reference raises). */
int32_t aha_ac_compile(const uint8_t *key_bytes, const uint64_t *key_offsets, uint32_t n_keys,
                       const aha_options *opts, aha_ac **out, uint32_t *err_key);
void aha_ac_free(aha_ac *ac);
/* A second handle for the same keys on `device` (< 0: the current one): the host side of `ac` is copied and uploaded,
 * nothing is compiled again.  What aha_group_compile does for every device after the first -- the reference's compile
 * is one call (src/aha/ac.cr:62-69), so n devices must not cost n compiles. */
int32_t aha_ac_replicate(const aha_ac *ac, int32_t device, aha_ac **out);
int32_t aha_ac_info(const aha_ac *ac, aha_ac_info_t *info);

/* AC#[](sid : Int) : String and AC#[](key) : Int -- delegated to the trie in
 * the reference (src/aha/ac.cr:41-43, src/aha/cedar.cr:747-749, 817-834).
 * aha_ac_key returns the key length (copies min(len,cap) bytes) or <0. */
int32_t aha_ac_key(const aha_ac *ac, int32_t id, uint8_t *buf, int32_t cap);
int32_t aha_ac_id(const aha_ac *ac, const uint8_t *key, int32_t len);

/* The batch entry below with the hits left on the device: host corpus and offsets in (uploaded range by range beside
 * the matches, like aha_ac_match_batch), hits into d_hits[0 .. cap) -- device memory of the handle's device --, per-document
 * offsets (optional) and the count to the host.  For callers that go on working with the hits on the GPU; each shard of
 * aha_group_match_batch is one such call.  AHA_E_CAPACITY: *n_hits is the required count. */
int32_t aha_ac_match_batch_keep(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                const aha_match_params *params, aha_hit *d_hits, uint64_t cap,
                                uint64_t *doc_hit_offsets, uint64_t *n_hits);

/* AC#match on ONE sequence held in host memory (uploads, matches on the GPU,
 * downloads).  Hits come back in the reference's order: ascending end
 * position, and per position own key first, then the output chain
 * (src/aha/ac.cr:176-192, 265-278).  params may be NULL (plain byte match). */
int32_t aha_ac_match_bytes(aha_ac *ac, const uint8_t *text, uint64_t n,
                           const aha_match_params *params, aha_hit *out, uint64_t cap,
                           uint64_t *n_hits);

/* Batch entry (new; the reference is one-sequence-per-call): D documents,
 * document d = corpus[doc_offsets[d] .. doc_offsets[d+1]); doc_offsets[0]
 * must be 0.  Equivalent to D independent #match calls concatenated;
 * doc_hit_offsets (D+1 entries, optional) delimits each document's hits. */
int32_t aha_ac_match_batch(aha_ac *ac, const uint8_t *corpus, const uint64_t *doc_offsets,
                           uint64_t n_docs, const aha_match_params *params, aha_hit *out,
                           uint64_t cap, uint64_t *doc_hit_offsets, uint64_t *n_hits);

/* Device-resident variant: every pointer prefixed d_ is HBM on the handle's
 * device; nothing crosses PCIe except the 8-byte hit count.  `stream` is a
 * hipStream_t (NULL = default stream).  Blocks until the hits are final.
 * On AHA_E_CAPACITY the first `cap` hits and all offsets are still valid.
 * d_doc_offsets must hold n_docs + 1 ascending offsets with [0] = 0 and [n_docs] = n_bytes, every document shorter
 * than 2^31 bytes: checked on the device before anything is indexed with them (AHA_E_INVALID / AHA_E_TOO_LONG) -- by a
 * small kernel in front of the traversal whose verdict every later kernel of the call looks at first, so a valid call
 * pays no read-back for it (match_longest and the two-pass engine read the verdict back before they start). */
int32_t aha_ac_match_batch_device(aha_ac *ac, const uint8_t *d_corpus,
                                  const uint64_t *d_doc_offsets, uint64_t n_docs,
                                  uint64_t n_bytes, const aha_match_params *params,
                                  aha_hit *d_out, uint64_t cap, uint64_t *d_doc_hit_offsets,
                                  uint64_t *n_hits, void *stream);

/* ABI 7 -- aha_ac_match_batch_device that leaves the hits ALSO as the 4-byte exchange stream (below: the format of
 * aha_ac_hits_pack4_device, bit-compatible with aha_ac_hits_unpack4_*): d_words[0 .. *d_n_words) on the device, capacity
 * cap_words >= 2 cap + ceil(cap / 1024) + 1 words: the pack kernels run behind the match on the same stream, one call instead
 * of two (what every resident shard of a group calls).  Final when the call returns. */
int32_t aha_ac_match_batch_device_stream(aha_ac *ac, const uint8_t *d_corpus, const uint64_t *d_doc_offsets, uint64_t n_docs,
                                         uint64_t n_bytes, const aha_match_params *params, aha_hit *d_out, uint64_t cap,
                                         uint64_t *d_doc_hit_offsets, uint64_t *n_hits, uint32_t *d_words, uint64_t cap_words,
                                         uint64_t *d_n_words, void *stream);

/* Exchange format for the multi-GPU all-gatherv of hit buffers (SURVEY.md
 * section 8 e): Hit#start = Hit#end - length of key[value] (src/aha/ac.cr:270-272;
 * with char offsets: - number of chars of the key), so ranks send {end, value}
 * pairs (8 B per hit instead of 12 on the xGMI links) and rebuild the triples
 * on arrival.  All pointers are HBM on the handle's device; the calls are
 * asynchronous on `stream`.  d_pairs holds 2*n int32, d_hits n triples. */
int32_t aha_ac_hits_pack_device(aha_ac *ac, const aha_hit *d_hits, uint64_t n, int32_t *d_pairs,
                                void *stream);
int32_t aha_ac_hits_unpack_device(aha_ac *ac, const int32_t *d_pairs, uint64_t n, int32_t char_offsets,
                                  aha_hit *d_hits, void *stream);

/* The 4-byte form of the same exchange.  Hits of a batch come in per-document order with ascending `end`, so the stream
 * carries one word per hit, value << (step_bits + len_bits) | len << step_bits | step, with step = end - previous end
 * and len = end - start (the key's length in the batch's offsets), and the absolute `end` only for the first hit of
 * every 1024, for a document change and for a gap of 2^step_bits - 1 or more:
 *   d_words = words[n] . first_exception[ceil(n/1024)] . exception_end[...]
 * The widths follow from the automaton (aha_ac_stream_format; every rank holds the same one): key ids take
 * bit_width(n_keys - 1) bits, lengths bit_width(longest key in bytes); with at least 10 bits left the step gets what is
 * left (at most 12) and the receiver rebuilds Hit#start without a table lookup; else len_bits = 0, step_bits = 12
 * (key ids below 2^20) and the length is looked up on arrival (a narrower step would turn every gap of a few hundred
 * bytes into an exception, 4 more bytes on the link).
 * pack4 needs cap_words >= 2 n + ceil(n/1024) (the worst case) and writes the real length -- what has to travel --
 * into *d_n_words (device memory); unpack4 takes the stream and n.  Asynchronous on `stream`. */
int32_t aha_ac_stream_format(const aha_ac *ac, uint32_t *step_bits, uint32_t *len_bits);
int32_t aha_ac_hits_pack4_device(aha_ac *ac, const aha_hit *d_hits, uint64_t n, uint32_t *d_words,
                                 uint64_t cap_words, uint64_t *d_n_words, void *stream);
int32_t aha_ac_hits_unpack4_device(aha_ac *ac, const uint32_t *d_words, uint64_t n, int32_t char_offsets,
                                   aha_hit *d_hits, void *stream);
/* Several streams rebuilt by ONE launch (an 8-GPU step receives seven peers' streams): stream k starts at word
 * word_offset of d_words, holds n_hits hits and is written to d_hits[out_offset ..].  At most 64 segments. */
typedef struct {
  uint64_t word_offset;
  uint64_t n_hits;
  uint64_t out_offset;
} aha_stream_seg;
int32_t aha_ac_hits_unpack4_segs_device(aha_ac *ac, const uint32_t *d_words, const aha_stream_seg *segs,
                                        uint32_t n_segs, int32_t char_offsets, aha_hit *d_hits, void *stream);

/* Copies one array of the automaton image (as uploaded to HBM) into buf;
 * returns its size in bytes (call with cap_bytes = 0 to size the buffer).
 * Data only -- used by host-logic tests and debugging tools. */
enum {
  AHA_IMG_SLOTS = 0,   /* uint32[n_slots] (compact) or uint64[n_slots] (wide) */
  AHA_IMG_END_KEY = 1, /* int32[n_slots], compact only */
  AHA_IMG_KEY_LN = 2,  /* {uint32 len, int32 next}[K] */
  AHA_IMG_KEY_CNT = 3, /* uint32[K] */
  AHA_IMG_KEY_KC = 4,  /* uint32[K] */
  AHA_IMG_UNIT_SLOTS = 6,     /* uint64[unit_slots]: a transition is lo = child base (22 bits) | filter (7) << 22 | F1 << 29 |
                                 NFR << 30 | END << 31, hi = symbol (16 bits) | min(hits, 255) << 16; slots[base] of a state with
                                 NFR and without F1 is its header {word of the fail state, 0}; group records and child runs of
                                 the big states: aha_amd/csrc/unit.hpp, IMAGE and BIG STATES */
  AHA_IMG_UNIT_ROOT = 7,      /* uint32[unit_syms]: the root's transitions by symbol */
  AHA_IMG_UNIT_END_KEY = 8,   /* int32[unit_slots]: key id at the base of an END state, else -1 */
  AHA_IMG_UNIT_TABLES = 9,    /* uint32[2816]: the decode tables (unit.hpp, SYMBOLS) */
  AHA_IMG_UNIT_MARKS = 10,    /* uint32[skip_filter_words]: the Bloom filter over the two-character paths (unit.hpp, MARKS) */
  AHA_IMG_UNIT_PAIRS = 11,    /* uint32[4 << pair_table_log2]: the pair table {raw0 | hits << 24, raw1, event payload, child filter} */
  AHA_IMG_UNIT_PAIR_DISP = 12, /* uint8[pair_groups]: its displacement bytes (unit.hpp, PAIR TABLE) */
  AHA_IMG_STALE_ENDS = 5     /* {uint32 key id, uint32 prefix length}[]: the states (a prefix of a key each) whose node in
                                 the reference's Cedar keeps a stale END flag (src/aha/cedar.cr:642-648); match_longest
                                 treats them as ends that yield nothing (src/aha/ac.cr:126-128, 249-263) */
};
int64_t aha_ac_export(const aha_ac *ac, int32_t which, void *buf, uint64_t cap_bytes);

/* AC#to_io / AC.from_io -- src/aha/ac.cr:45-60 (`save`/`load` in the reference's
 * README).  The reference's on-disk framing goes through the un-vendored
 * `super_io` shard and no reference test reads a loaded automaton back
 * (parity unpinned), so this is the library's OWN
 * container, not the Crystal byte stream: little endian
 *   "AHAHIP01" | u32 format=1 | u32 K | u64 blob_bytes | u64 offs[K+1] | blob |
 *   u64 FNV-1a of everything before it.
 * It stores the keys in `compile` order; aha_ac_load re-derives the automaton
 * (deterministic, same key ids), so a file written by one build loads in any
 * later one.  aha_ac_save returns the size in bytes (cap_bytes = 0 sizes the
 * buffer) or <0; aha_ac_load returns AHA_E_INVALID for a truncated/corrupt
 * buffer, else whatever aha_ac_compile returns. */
int64_t aha_ac_save(const aha_ac *ac, void *buf, uint64_t cap_bytes);
int32_t aha_ac_load(const void *buf, uint64_t n_bytes, const aha_options *opts, aha_ac **out);

/* Frees the handle's device scratch (it grows with the largest batch seen and is otherwise kept for reuse). */
int32_t aha_ac_release_scratch(aha_ac *ac);
/* Device bytes currently held as scratch by the handle (all sets); waits for running calls. */
int64_t aha_ac_scratch_bytes(aha_ac *ac);

/* Enable/disable HIP-event timing of device matches on this handle.  aha_timing.engine tells which engine answered the
 * last call: keys longer than 4096 bytes, a NULL-capacity sizing call and event-temp overflow take the two-pass
 * engine (1), which is an order of magnitude slower than the single-traversal engine (2). */
int32_t aha_ac_set_profiling(aha_ac *ac, int32_t enabled);
int32_t aha_ac_last_timing(const aha_ac *ac, aha_timing *t);

/* ---- device memory for callers without a GPU framework (SURVEY.md section 8 b: aha_corpus_upload / _free) ----------
 * aha_ac_match_batch_device takes HBM pointers; these give a C, C++ or Crystal caller a way to obtain them, so that a
 * corpus is uploaded once and matched many times (or by several automata) and the hits stay on the device until they
 * are wanted.  Copies run on a private stream of the library and block the calling thread only.  Buffers are 256-byte
 * aligned.  Errors: AHA_E_HIP / AHA_E_INVALID / AHA_E_NO_DEVICE, text through aha_last_error(NULL). */
int32_t aha_buffer_alloc(int32_t device, uint64_t bytes, void **d_ptr);
int32_t aha_buffer_free(int32_t device, void *d_ptr);
int32_t aha_buffer_upload(int32_t device, void *d_dst, const void *src, uint64_t bytes);
int32_t aha_buffer_download(int32_t device, void *dst, const void *d_src, uint64_t bytes);
/* A batch resident in HBM: the bytes of all documents and their D + 1 offsets (validated like aha_ac_match_batch
 * validates them: ascending from 0, every document below 2^31 bytes). */
typedef struct aha_corpus aha_corpus;
int32_t aha_corpus_upload(int32_t device, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                          aha_corpus **out);
void aha_corpus_free(aha_corpus *c);
const uint8_t *aha_corpus_bytes(const aha_corpus *c);         /* device pointer */
const uint64_t *aha_corpus_doc_offsets(const aha_corpus *c);  /* device pointer, n_docs + 1 entries */
uint64_t aha_corpus_n_docs(const aha_corpus *c);
uint64_t aha_corpus_n_bytes(const aha_corpus *c);
int32_t aha_corpus_device(const aha_corpus *c);

/* ---- several GPUs of one node behind one handle (SURVEY.md section 8 b / e) -------------------------------------
 * The batch is cut into contiguous, byte-balanced document ranges, one per entry of `devices` (documents are
 * independent: src/aha/ac.cr:177; contiguous ranges make the global hit order the range order); the automaton is
 * replicated; every device matches its range; the hit buffers are exchanged with an all-gatherv, so every device
 * holds the whole ordered hit stream, and come back to the caller exactly as aha_ac_match_batch would return them.
 * Between distinct devices the exchange is RCCL over xGMI (all-pairs ncclSend/ncclRecv in one group; librccl.so is
 * loaded on first use); entries that name the same device twice -- several shards on one GPU, the form a 1-GPU box
 * can run -- exchange with device-to-device copies.  Buffers are host memory; the calls block; calls on one group
 * are serialised inside the library.  The capacity is checked before the exchange: on AHA_E_CAPACITY the devices
 * hold their own shards' hits only. */
typedef struct aha_group aha_group;

typedef struct {
  uint32_t struct_size;
  uint32_t n_devices;
  float ms_match;            /* upload + match of all shards, wall clock (shards run concurrently) */
  float ms_match_max_shard;  /* the slowest shard's device match alone */
  float ms_exchange;         /* the all-gatherv of the hit buffers */
  float ms_download;         /* gathered hits -> caller's buffer */
  uint64_t n_hits;
  uint32_t exchange;         /* 1 = RCCL between distinct devices, 0 = device-to-device copies on one device,
                                2 = the rehearsal AHA_GROUP_RCCL=self: every shard's own stream through RCCL (one
                                communicator of one rank per shard), the peers' by copies */
  uint32_t packed;           /* 1 = the 4-byte exchange stream travelled, 0 = the 12-byte triples (2^20 keys or more) */
  uint64_t wire_bytes;       /* payload bytes of all shards together (each goes to every other shard) */
} aha_group_timing;

int32_t aha_group_compile(const uint8_t *key_bytes, const uint64_t *key_offsets, uint32_t n_keys,
                          const int32_t *devices, int32_t n_devices, uint32_t flags, aha_group **out,
                          uint32_t *err_key);
void aha_group_free(aha_group *g);
int32_t aha_group_size(const aha_group *g);
const char *aha_group_last_error(const aha_group *g);
/* bounds[0..n_parts]: shard r holds documents bounds[r] .. bounds[r+1]-1 (the boundary nearest to r * N / n_parts). */
int32_t aha_group_partition(const uint64_t *doc_offsets, uint64_t n_docs, int32_t n_parts, uint64_t *bounds);
/* Same contract as aha_ac_match_batch (AHA_E_CAPACITY: *n_hits = required count, offsets valid). */
int32_t aha_group_match_batch(aha_group *g, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                              const aha_match_params *params, aha_hit *out, uint64_t cap,
                              uint64_t *doc_hit_offsets, uint64_t *n_hits);
/* The gathered hit stream as device `shard` holds it after the last aha_group_match_batch (every device holds the
 * whole ordered stream: this is how a test, or a caller that wants a particular device's copy, reads it back). */
int32_t aha_group_download_shard(aha_group *g, int32_t shard, aha_hit *out, uint64_t cap, uint64_t *n_hits);
int32_t aha_group_last_timing(const aha_group *g, aha_group_timing *t);
/* ABI 7 -- the batch RESIDENT on the group's devices (the group counterpart of aha_corpus_upload + aha_ac_match_batch_device;
 * keeps the reference's one call per batch, src/aha/ac.cr:280-286): aha_group_corpus_upload partitions the documents like
 * aha_group_match_batch does and leaves every shard's range on its device (each over its own PCIe link);
 * aha_group_match_batch_device matches the resident ranges, every device its own, and runs the same all-gatherv -- nothing of
 * the batch crosses PCIe any more.  The hits stay on the devices: every device holds the whole ordered stream
 * (aha_group_download_shard reads one device's copy); *n_hits and doc_hit_offsets (n_docs + 1 entries, optional) are host
 * memory.  A corpus belongs to the group it was uploaded for and must be freed before that group. */
typedef struct aha_group_corpus aha_group_corpus;
int32_t aha_group_corpus_upload(aha_group *g, const uint8_t *corpus, const uint64_t *doc_offsets, uint64_t n_docs,
                                aha_group_corpus **out);
void aha_group_corpus_free(aha_group_corpus *c);
int32_t aha_group_match_batch_device(aha_group *g, const aha_group_corpus *c, const aha_match_params *params,
                                     uint64_t *doc_hit_offsets, uint64_t *n_hits);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* AHA_HIP_H */
