/*
 * aha_oracle.h -- CPU restatement of chenkovsky/aha's Aha::AC#match path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.  The product path (libaha_hip.so) never links,
 * loads or calls it.
 *
 * Parity status: PINNED by the reference's own known-answer specs
 * (spec/ac_spec.cr, spec/ac_longest_match_spec.cr, spec/cedar_spec.cr,
 * README.md:29-38); see tests/test_oracle_kats.py.  The reference is Crystal
 * and cannot be compiled here (no crystal toolchain, deps un-vendored), so
 * there is no oracle/_ref build.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * that it follows.
 */
#ifndef AHA_ORACLE_H
#define AHA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Aha::Hit, src/aha/matcher.cr:2-11 */
typedef struct {
  int32_t start, end, value;
} orc_hit;

typedef struct orc_cedar orc_cedar;
typedef struct orc_ac orc_ac;

/* error codes (the reference raises Strings; messages in orc_strerror) */
enum {
  ORC_OK = 0,
  ORC_E_EMPTY_KEY = -2,   /* "Cannot insert empty key"      cedar.cr:756 */
  ORC_E_ZERO_BYTE = -3,   /* "key[pos] is zero"             cedar.cr:235 */
  ORC_E_DUP_KEY = -4,     /* "key:... appear twice."        ac.cr:66     */
  ORC_E_SEP_SIZE = -5,    /* "sep BitArray size > 256 ..."  ac.cr:322    */
  ORC_E_ORDERED = -6,     /* ordered=true unsupported (cedar.cr:409-411 loops forever) */
  ORC_E_NOT_FOUND = -7
};
const char *orc_strerror(int code);

/* ---- CedarX(Int32), src/aha/cedar.cr ---- */
orc_cedar *orc_cedar_new(void);                                           /* :198-221 */
void orc_cedar_free(orc_cedar *c);
int32_t orc_cedar_insert(orc_cedar *c, const uint8_t *key, int32_t len);  /* :755-778 -> id or ORC_E_* */
int32_t orc_cedar_get(const orc_cedar *c, const uint8_t *key, int32_t len); /* []? :822-828 -> id or -1 */
int32_t orc_cedar_delete(orc_cedar *c, const uint8_t *key, int32_t len);  /* :785-815 */
int32_t orc_cedar_key(const orc_cedar *c, int32_t id, uint8_t *buf, int32_t cap); /* [](sid) :747-749 -> len or ORC_E_* */
int32_t orc_cedar_array_size(const orc_cedar *c);
int32_t orc_cedar_key_num(const orc_cedar *c);
int32_t orc_cedar_leaf_size(const orc_cedar *c);

/* ---- ACX(Int32), src/aha/ac.cr ---- */
/* AC.compile(keys) :62-69.  Keys are a concatenated blob + K+1 offsets.
 * On error returns NULL and sets *err (ORC_E_*), *err_key = offending index. */
orc_ac *orc_ac_compile_keys(const uint8_t *blob, const uint64_t *offs, uint32_t K,
                            int *err, uint32_t *err_key);
/* AC.compile(da) :71-112 (takes ownership of the trie). */
orc_ac *orc_ac_compile_cedar(orc_cedar *c);
void orc_ac_free(orc_ac *a);
int32_t orc_ac_slots(const orc_ac *a);      /* da.array_size */
int32_t orc_ac_keys(const orc_ac *a);       /* da.leaf_size  */
uint32_t orc_ac_max_key_len(const orc_ac *a);
int32_t orc_ac_key(const orc_ac *a, int32_t id, uint8_t *buf, int32_t cap); /* delegate [] ac.cr:42 */
int32_t orc_ac_id(const orc_ac *a, const uint8_t *key, int32_t len);        /* delegate []? ac.cr:43 */

/* match(Bytes) :280-286 (match_ :176-192 + fetch :265-278).
 * Writes at most cap hits, returns the total number the reference yields.
 * sep_bits==NULL: plain match.  Otherwise match(seq, sep) :321-340 with a
 * BitArray of sep_size bits (LSB-first in sep_bits), sep_size<=256.
 * char_offsets!=0: the String overload, matcher.cr:34-46 (valid UTF-8 only). */
int64_t orc_ac_match(const orc_ac *a, const uint8_t *text, int64_t n, int char_offsets,
                     const uint8_t *sep_bits, int32_t sep_size, orc_hit *out, int64_t cap);
/* match_longest(Bytes, intersectable) :297-303 (+ String overload :305-310). */
int32_t orc_ac_stale_ends(const orc_ac *a); /* nodes with a stale END flag (cedar.cr:642-648) */
/* the same nodes as [u32 length][path bytes] records; returns the bytes needed, writes when they fit cap */
int64_t orc_ac_stale_paths(const orc_ac *a, uint8_t *buf, int64_t cap);
int64_t orc_ac_match_longest(const orc_ac *a, const uint8_t *text, int64_t n, int intersectable,
                             int char_offsets, orc_hit *out, int64_t cap);
/* One `match` call per document (the reference has no batch entry; this is
 * the loop a caller would write).  doc_hit_offsets has D+1 entries. */
int64_t orc_ac_match_batch(const orc_ac *a, const uint8_t *corpus, const uint64_t *doc_offsets,
                           uint64_t D, int char_offsets, orc_hit *out, int64_t cap,
                           uint64_t *doc_hit_offsets);

#ifdef __cplusplus
}
#endif
#endif
