/*
 * aha_oracle.c -- CPU restatement (plain C) of the reference's Aha::AC#match
 * path: Cedar double-array trie builder + traversal, AC compile, match_,
 * fetch, char_map.  TEST INFRASTRUCTURE ONLY -- see aha_oracle.h.
 *
 * Written from the cited reference lines; the data structures (12-byte AoS
 * node, separate fails, {next,value} output pairs, key_lens) keep the
 * reference's layout so that timing this code is a fair stand-in for the
 * reference CPU path ("C restatement of the reference CPU path").
 *
 * Integer semantics: the reference is Crystal 0.23 where Int32 arithmetic
 * wraps; base() relies on that for value == Int32::MAX (cedar.cr:81-83,355).
 */
#include "aha_oracle.h"

#include <stdlib.h>
#include <string.h>

#define VALUE_LIMIT INT32_MAX /* CedarX.value_limit cedar.cr:10-12 */
#define VALUE_MIN INT32_MIN   /* CedarX.value_min   cedar.cr:14-16 */
#define CHILD_NUM_MASK 0x1FF  /* cedar.cr:62 */
#define END_MASK 0x200        /* cedar.cr:63 */

/* Node(Int32) cedar.cr:29-34: 12 bytes */
typedef struct {
  int32_t value;
  int32_t check;
  uint8_t sibling;
  uint8_t child;
  uint16_t flags;
} node_t;

/* Block(Int32) cedar.cr:98-104 */
typedef struct {
  int32_t prev, next;
  int32_t num, reject, trial;
  int32_t ehead;
} block_t;

struct orc_cedar {
  int32_t key_num, key_capacity;
  node_t *array;
  block_t *blocks;
  int32_t reject[257];
  int32_t bheadF, bheadC, bheadO;
  int32_t array_size, capacity;
  int32_t max_trial;
  int32_t *leafs;
  int32_t leaf_size;
  int32_t free_leaf_slot;
};

/* OutNode ac.cr:15-33 */
typedef struct {
  int32_t next, value;
} outnode_t;

struct orc_ac {
  orc_cedar *da;
  outnode_t *output;
  int32_t *fails;
  uint32_t *key_lens;
  uint32_t max_len;
};

const char *orc_strerror(int code) {
  switch (code) {
    case ORC_OK: return "ok";
    case ORC_E_EMPTY_KEY: return "Cannot insert empty key";
    case ORC_E_ZERO_BYTE: return "key[pos] is zero";
    case ORC_E_DUP_KEY: return "key appear twice.";
    case ORC_E_SEP_SIZE: return "sep BitArray size > 256 is not supported";
    case ORC_E_ORDERED: return "ordered Cedar is not supported";
    case ORC_E_NOT_FOUND: return "not found";
  }
  return "unknown";
}

/* Node#base cedar.cr:81-83, with Crystal-0.23 wrap-around */
static inline int32_t nbase(const node_t *n) {
  return (int32_t)(0u - ((uint32_t)n->value + 1u));
}
static inline node_t mknode(int32_t value, int32_t check) { /* Node#initialize cedar.cr:56-60 */
  node_t n;
  n.value = value;
  n.check = check;
  n.sibling = 0;
  n.child = 0;
  n.flags = 0;
  return n;
}
static inline uint16_t child_num(const node_t *n) { return n->flags & CHILD_NUM_MASK; }
static inline void set_child_num(node_t *n, uint16_t v) { /* cedar.cr:69-71 */
  n->flags = (uint16_t)((n->flags & (uint16_t)~CHILD_NUM_MASK) | v);
}

/* CedarX#initialize cedar.cr:198-221 (ordered=false) */
orc_cedar *orc_cedar_new(void) {
  orc_cedar *c = (orc_cedar *)calloc(1, sizeof(*c));
  c->capacity = 256;
  c->key_num = 0;
  c->leaf_size = 0;
  c->key_capacity = c->capacity;
  c->leafs = (int32_t *)calloc((size_t)c->key_capacity, sizeof(int32_t));
  c->array = (node_t *)calloc((size_t)c->capacity, sizeof(node_t));
  c->array_size = c->capacity;
  c->blocks = (block_t *)calloc((size_t)(c->capacity >> 8), sizeof(block_t));
  c->max_trial = 1;
  c->array[0] = mknode(-1, -1);
  for (int i = 1; i < 256; i++) c->array[i] = mknode(-(i - 1), -(i + 1));
  c->array[1].value = -255;
  c->array[255].check = -1;
  /* Block.new(prev,next,trial,ehead,num=256,reject=257) cedar.cr:127,215 */
  c->blocks[0].prev = 0;
  c->blocks[0].next = 0;
  c->blocks[0].trial = 0;
  c->blocks[0].ehead = 1;
  c->blocks[0].num = 256;
  c->blocks[0].reject = 257;
  for (int i = 0; i < 257; i++) c->reject[i] = i + 1;
  c->bheadF = c->bheadC = c->bheadO = 0;
  c->free_leaf_slot = VALUE_MIN;
  return c;
}

void orc_cedar_free(orc_cedar *c) {
  if (!c) return;
  free(c->leafs);
  free(c->array);
  free(c->blocks);
  free(c);
}

int32_t orc_cedar_array_size(const orc_cedar *c) { return c->array_size; }
int32_t orc_cedar_key_num(const orc_cedar *c) { return c->key_num; }
int32_t orc_cedar_leaf_size(const orc_cedar *c) { return c->leaf_size; }

/* pop_block cedar.cr:266-278 */
static void pop_block(orc_cedar *c, int32_t bi, int32_t *head_in, int last) {
  if (last) {
    *head_in = 0;
  } else {
    block_t *b = &c->blocks[bi];
    c->blocks[b->prev].next = b->next;
    c->blocks[b->next].prev = b->prev;
    if (bi == *head_in) *head_in = b->next;
  }
}

/* push_block cedar.cr:285-296 */
static void push_block(orc_cedar *c, int32_t bi, int32_t *head_out, int empty) {
  block_t *b = &c->blocks[bi];
  if (empty) {
    *head_out = bi;
    b->prev = bi;
    b->next = bi;
  } else {
    block_t *tail_out = &c->blocks[*head_out];
    b->prev = tail_out->prev;
    b->next = *head_out;
    c->blocks[tail_out->prev].next = bi;
    *head_out = bi;
    tail_out->prev = bi;
  }
}

/* add_block cedar.cr:299-313 */
static int32_t add_block(orc_cedar *c) {
  if (c->array_size == c->capacity) {
    c->capacity *= 2;
    c->blocks = (block_t *)realloc(c->blocks, (size_t)(c->capacity >> 8) * sizeof(block_t));
    c->array = (node_t *)realloc(c->array, (size_t)c->capacity * sizeof(node_t));
  }
  block_t *nb = &c->blocks[c->array_size >> 8];
  nb->prev = 0;
  nb->next = 0;
  nb->trial = 0;
  nb->ehead = c->array_size;
  nb->num = 256;
  nb->reject = 257;
  for (int i = 0; i < 256; i++)
    c->array[c->array_size + i] =
        mknode(-(((i + 255) & 255) + c->array_size), -(((i + 1) & 255) + c->array_size));
  push_block(c, c->array_size >> 8, &c->bheadO, c->bheadO == 0);
  c->array_size += 256;
  return (c->array_size >> 8) - 1;
}

/* transfer_block cedar.cr:320-323 */
static void transfer_block(orc_cedar *c, int32_t bi, int32_t *head_in, int32_t *head_out) {
  pop_block(c, bi, head_in, bi == c->blocks[bi].next);
  push_block(c, bi, head_out, *head_out == 0 && c->blocks[bi].num != 0);
}

/* find_place cedar.cr:526-530 */
static int32_t find_place(orc_cedar *c) {
  if (c->bheadC != 0) return c->blocks[c->bheadC].ehead;
  if (c->bheadO != 0) return c->blocks[c->bheadO].ehead;
  return add_block(c) << 8;
}

/* find_places cedar.cr:533-580 */
static int32_t find_places(orc_cedar *c, const uint8_t *children, int children_size) {
  int32_t bi = c->bheadO;
  if (bi != 0) {
    int32_t bz = c->blocks[c->bheadO].prev;
    int32_t nc = children_size;
    for (;;) {
      block_t *b = &c->blocks[bi];
      if (b->num >= nc && nc < b->reject) {
        int32_t e = b->ehead;
        for (;;) {
          int32_t base = e ^ children[0];
          for (int i = 0; i < children_size; i++) {
            uint8_t ch = children[i];
            if (!(c->array[base ^ ch].check < 0)) break;
            if (i == children_size - 1) {
              b->ehead = e;
              return e;
            }
          }
          e = -c->array[e].check;
          if (e == b->ehead) break;
        }
      }
      b->reject = nc;
      if (b->reject < c->reject[b->num]) c->reject[b->num] = b->reject;
      int32_t bin = b->next;
      b->trial = b->trial + 1;
      if (b->trial == c->max_trial) transfer_block(c, bi, &c->bheadO, &c->bheadC);
      if (bi == bz) break;
      bi = bin;
    }
  }
  return add_block(c) << 8;
}

/* pop_enode cedar.cr:329-359 */
static int32_t pop_enode(orc_cedar *c, int32_t base, uint8_t label, int32_t from) {
  int32_t e = base < 0 ? find_place(c) : (base ^ label);
  int32_t bi = e >> 8;
  node_t *n = &c->array[e];
  block_t *b = &c->blocks[bi];
  b->num = b->num - 1;
  if (b->num == 0) {
    if (bi != 0) transfer_block(c, bi, &c->bheadC, &c->bheadF);
  } else {
    c->array[-n->value].check = n->check;
    c->array[-n->check].value = n->value;
    if (e == b->ehead) b->ehead = -n->check;
    if (bi != 0 && b->num == 1 && b->trial != c->max_trial)
      transfer_block(c, bi, &c->bheadO, &c->bheadC);
  }
  n->value = VALUE_LIMIT;
  n->check = from;
  if (base < 0) c->array[from].value = -(e ^ label) - 1;
  return e;
}

/* push_enode cedar.cr:361-397 */
static void push_enode(orc_cedar *c, int32_t e) {
  node_t *ep = &c->array[e];
  int32_t bi = e >> 8;
  block_t *b = &c->blocks[bi];
  b->num = b->num + 1;
  if (b->num == 1) {
    b->ehead = e;
    ep->value = -e;
    ep->check = -e;
    if (bi != 0) transfer_block(c, bi, &c->bheadF, &c->bheadC);
  } else {
    int32_t prev = b->ehead;
    node_t *pp = &c->array[prev];
    int32_t next_ = -pp->check;
    ep->value = -prev;
    ep->check = -next_;
    pp->check = -e;
    c->array[next_].value = -e;
    if (b->num == 2 || b->trial == c->max_trial) {
      if (bi != 0) transfer_block(c, bi, &c->bheadC, &c->bheadO);
    }
    b->trial = 0;
  }
  if (b->reject < c->reject[b->num]) b->reject = c->reject[b->num];
  ep->child = 0;
  ep->sibling = 0;
  ep->flags = 0;
}

/* push_sibling cedar.cr:403-417 (ordered=false) */
static void push_sibling(orc_cedar *c, int32_t from, int32_t base, uint8_t label, int has_child) {
  node_t *fp = &c->array[from];
  uint8_t *child_ptr = &fp->child;
  int keep_order = (*child_ptr == 0);
  if (has_child && keep_order) child_ptr = &c->array[base ^ *child_ptr].sibling;
  c->array[base ^ label].sibling = *child_ptr;
  *child_ptr = label;
  set_child_num(fp, (uint16_t)(child_num(fp) + 1));
}

/* pop_sibling cedar.cr:420-429 */
static void pop_sibling(orc_cedar *c, int32_t from, uint8_t label) {
  node_t *fp = &c->array[from];
  int32_t base = nbase(fp);
  uint8_t *child_ptr = &fp->child;
  while (*child_ptr != label) child_ptr = &c->array[base ^ *child_ptr].sibling;
  *child_ptr = c->array[base ^ *child_ptr].sibling;
  set_child_num(fp, (uint16_t)(child_num(fp) - 1));
}

/* set_child cedar.cr:498-523 (ordered=false) */
static int set_child(orc_cedar *c, int32_t base, uint8_t ch, uint8_t label, int append_label,
                     uint8_t *children) {
  int idx = 0;
  if (ch == 0) {
    children[idx++] = ch;
    ch = c->array[base ^ ch].sibling;
  }
  if (append_label) children[idx++] = label;
  while (ch != 0) {
    children[idx++] = ch;
    ch = c->array[base ^ ch].sibling;
  }
  return idx;
}

/* resolve cedar.cr:582-655 */
static int32_t resolve(orc_cedar *c, int32_t from_n, int32_t base_n, uint8_t label_n) {
  int32_t to_pn = base_n ^ label_n;
  int32_t from_p = c->array[to_pn].check;
  int32_t base_p = nbase(&c->array[from_p]);
  /* consult cedar.cr:432-434 */
  int flag = child_num(&c->array[from_n]) < child_num(&c->array[from_p]);
  uint8_t children_[257];
  memset(children_, 0, sizeof(children_));
  int children_size;
  if (flag)
    children_size = set_child(c, base_n, c->array[from_n].child, label_n, 1, children_);
  else
    children_size = set_child(c, base_p, c->array[from_p].child, 255, 0, children_);
  int32_t base = children_size == 1 ? find_place(c) : find_places(c, children_, children_size);
  base ^= children_[0];
  int32_t from, base_;
  if (flag) {
    from = from_n;
    base_ = base_n;
  } else {
    from = from_p;
    base_ = base_p;
  }
  if (flag && children_[0] == label_n) c->array[from].child = label_n;
  c->array[from].value = -base - 1;
  for (int i = 0; i < children_size; i++) {
    uint8_t chl = children_[i];
    int32_t to = pop_enode(c, base, chl, from);
    int32_t to_ = base_ ^ chl;
    node_t *n = &c->array[to];
    n->sibling = (i == children_size - 1) ? 0 : children_[i + 1];
    if (flag && to_ == to_pn) continue;
    node_t *n_ = &c->array[to_];
    n->value = n_->value;
    if (n_->value >= 0 && n_->value != VALUE_LIMIT) c->leafs[n_->value] = to;
    n->flags = n_->flags;
    if (n->value < 0 && chl != 0) {
      uint8_t ch = c->array[to_].child;
      c->array[to].child = ch;
      node_t *ptr = &c->array[nbase(n) ^ ch];
      ptr->check = to;
      ch = ptr->sibling;
      while (ch != 0) {
        ptr = &c->array[nbase(n) ^ ch];
        ptr->check = to;
        ch = ptr->sibling;
      }
    }
    if (!flag && to_ == from_n) from_n = to;
    if (!flag && to_ == to_pn) {
      push_sibling(c, from_n, to_pn ^ label_n, label_n, 1);
      c->array[to_].child = 0;
      n_->value = VALUE_LIMIT;
      n_->check = from_n;
      /* NB: flags of the reused slot are NOT cleared (cedar.cr:642-648) */
    } else {
      push_enode(c, to_);
    }
  }
  if (flag) return base ^ label_n;
  return to_pn;
}

/* follow cedar.cr:244-259 */
static int32_t follow(orc_cedar *c, int32_t from, uint8_t label) {
  int32_t base = nbase(&c->array[from]);
  int32_t to = base ^ label;
  if (base < 0 || c->array[to].check < 0) {
    int has_child = base >= 0 && (c->array[base ^ c->array[from].child].check == from);
    to = pop_enode(c, base, label, from);
    push_sibling(c, from, to ^ label, label, has_child);
  } else if (c->array[to].check != from) {
    to = resolve(c, from, base, label);
  }
  return to;
}

/* get cedar.cr:224-241; returns node id or ORC_E_ZERO_BYTE */
static int32_t cedar_get_create(orc_cedar *c, const uint8_t *key, int32_t len, int32_t from,
                                int32_t start) {
  for (int32_t pos = start; pos < len; pos++) {
    int32_t value = c->array[from].value;
    if (value >= 0 && value != VALUE_LIMIT) {
      int32_t to = follow(c, from, 0);
      c->array[to].value = value;
      c->leafs[value] = to;
    }
    if (key[pos] == 0) return ORC_E_ZERO_BYTE;
    from = follow(c, from, key[pos]);
  }
  return c->array[from].value < 0 ? follow(c, from, 0) : from;
}

/* jump(byte) cedar.cr:678-686 */
static inline int32_t jump1(const orc_cedar *c, uint8_t byte, int32_t from) {
  const node_t *fp = &c->array[from];
  if (fp->value >= 0) return -1;
  int32_t to = nbase(fp) ^ byte;
  if (c->array[to].check != from) return -1;
  return to;
}

/* value cedar.cr:726-734 */
static inline int32_t cedar_value(const orc_cedar *c, int32_t id) {
  const node_t *p = &c->array[id];
  int32_t val = p->value;
  if (val >= 0) return val;
  int32_t to = nbase(p);
  const node_t *tp = &c->array[to];
  if (tp->check == id && tp->value >= 0 && tp->value != VALUE_LIMIT) return tp->value;
  return -1;
}

/* []?(key) cedar.cr:822-828 via jump(path) :688-694 */
int32_t orc_cedar_get(const orc_cedar *c, const uint8_t *key, int32_t len) {
  int32_t from = 0;
  for (int32_t i = 0; i < len; i++) {
    from = jump1(c, key[i], from);
    if (from < 0) return -1;
  }
  int32_t vk = cedar_value(c, from);
  return vk < 0 ? -1 : vk;
}

/* insert cedar.cr:755-778 */
int32_t orc_cedar_insert(orc_cedar *c, const uint8_t *key, int32_t len) {
  if (len == 0) return ORC_E_EMPTY_KEY;
  int32_t id = orc_cedar_get(c, key, len);
  if (id >= 0) return id;
  int32_t p = cedar_get_create(c, key, len, 0, 0);
  if (p < 0) return p;
  if (c->free_leaf_slot != VALUE_MIN) {
    int32_t cur = -c->free_leaf_slot - 1;
    c->free_leaf_slot = c->leafs[cur];
    id = cur;
  } else {
    if (c->leaf_size == c->key_capacity) {
      c->key_capacity *= 2;
      c->leafs = (int32_t *)realloc(c->leafs, (size_t)c->key_capacity * sizeof(int32_t));
    }
    c->leaf_size += 1;
    id = c->leaf_size - 1;
  }
  c->array[p].value = id;
  c->array[p].flags |= END_MASK; /* end! cedar.cr:77-79 */
  c->leafs[id] = p;
  c->key_num += 1;
  return id;
}

/* delete cedar.cr:785-815 */
int32_t orc_cedar_delete(orc_cedar *c, const uint8_t *key, int32_t len) {
  int32_t to = 0;
  for (int32_t i = 0; i < len; i++) {
    to = jump1(c, key[i], to);
    if (to < 0) return -1;
  }
  int32_t vk = cedar_value(c, to);
  if (vk < 0) return -1;
  if (c->array[to].value < 0) {
    int32_t base = nbase(&c->array[to]);
    if (c->array[base].check == to) to = base;
  }
  for (;;) {
    node_t *tp = &c->array[to];
    int32_t from = tp->check;
    node_t *fp = &c->array[from];
    int32_t base = nbase(fp);
    uint8_t label = (uint8_t)(to ^ base);
    if (tp->sibling != 0 || fp->child != label) {
      pop_sibling(c, from, label);
      push_enode(c, to);
      break;
    }
    push_enode(c, to);
    to = from;
  }
  c->key_num -= 1;
  c->leafs[vk] = c->free_leaf_slot;
  c->free_leaf_slot = -(vk + 1);
  return vk;
}

/* key(id) cedar.cr:707-722 + [](sid) :747-749 (String.new stops at the NUL) */
int32_t orc_cedar_key(const orc_cedar *c, int32_t sid, uint8_t *buf, int32_t cap) {
  if (sid < 0 || sid >= c->leaf_size) return ORC_E_NOT_FOUND;
  int32_t id = c->leafs[sid];
  int32_t n = 0;
  /* first pass: length */
  for (int32_t x = id; x > 0;) {
    int32_t from = c->array[x].check;
    if (from < 0) return ORC_E_NOT_FOUND; /* "no path" */
    int32_t chr = nbase(&c->array[from]) ^ x;
    if (chr != 0) n++;
    x = from;
  }
  if (n == 0) return ORC_E_NOT_FOUND; /* "invalid key" */
  if (n > cap) return n;
  int32_t w = n;
  for (int32_t x = id; x > 0;) {
    int32_t from = c->array[x].check;
    int32_t chr = nbase(&c->array[from]) ^ x;
    if (chr != 0) buf[--w] = (uint8_t)chr;
    x = from;
  }
  return n;
}

/* child cedar.cr:441-447 */
static inline int32_t cedar_child(const orc_cedar *c, int32_t id, uint8_t label) {
  int32_t base = nbase(&c->array[id]);
  int32_t cid = base ^ label;
  if (cid < 0 || cid >= c->array_size || c->array[cid].check != id) return -1;
  return cid;
}

/* is_end? cedar.cr:657-660 */
static inline int cedar_is_end(const orc_cedar *c, int32_t id) {
  if (c->array[id].flags & END_MASK) return 1;
  return c->array[id].child == 0;
}

/* ------------------------------------------------------------------ AC */

typedef struct {
  int32_t id;
  int32_t len;
} qent_t;

/* AC.compile(da) ac.cr:71-112 */
orc_ac *orc_ac_compile_cedar(orc_cedar *da) {
  orc_ac *a = (orc_ac *)calloc(1, sizeof(*a));
  a->da = da;
  int32_t nlen = da->array_size;
  a->fails = (int32_t *)malloc((size_t)nlen * sizeof(int32_t));
  a->output = (outnode_t *)malloc((size_t)nlen * sizeof(outnode_t));
  for (int32_t i = 0; i < nlen; i++) {
    a->fails[i] = -1;
    a->output[i].next = -1;
    a->output[i].value = -1;
  }
  size_t nk = da->leaf_size > 0 ? (size_t)da->leaf_size : 1;
  a->key_lens = (uint32_t *)calloc(nk, sizeof(uint32_t));
  qent_t *q = (qent_t *)malloc((size_t)nlen * sizeof(qent_t));
  size_t qh = 0, qt = 0;
  const int32_t ro = 0;
  a->fails[ro] = ro;
  /* children(ro) cedar.cr:450-463 */
  {
    const node_t *pp = &da->array[ro];
    int32_t base = nbase(pp);
    uint8_t s = pp->child;
    if (s == 0 && base > 0) s = da->array[base].sibling;
    while (s != 0) {
      int32_t to = base ^ s;
      if (to < 0) break;
      a->fails[to] = ro;
      q[qt].id = to;
      q[qt].len = 1;
      qt++;
      s = da->array[to].sibling;
    }
  }
  while (qh < qt) {
    qent_t e = q[qh++];
    int32_t nid = e.id;
    int32_t l = e.len;
    if (cedar_is_end(da, nid)) {
      int32_t vk = cedar_value(da, nid);
      /* reference: key_lens[vk] = l (ac.cr:91) -- with a stale END flag
       * (cedar.cr:642-648) vk is -1 and the store is out of bounds; the
       * oracle skips that store (unobservable through match). */
      if (vk >= 0) {
        a->key_lens[vk] = (uint32_t)l;
        if ((uint32_t)l > a->max_len) a->max_len = (uint32_t)l;
      }
      a->output[nid].value = vk;
    }
    const node_t *pp = &da->array[nid];
    int32_t base = nbase(pp);
    uint8_t s = pp->child;
    if (s == 0 && base > 0) s = da->array[base].sibling;
    while (s != 0) {
      int32_t cid = base ^ s;
      if (cid < 0) break;
      q[qt].id = cid;
      q[qt].len = l + 1;
      qt++;
      int32_t fid = nid;
      while (fid != ro) {
        int32_t fs = a->fails[fid];
        int32_t t = cedar_child(da, fs, s);
        if (t >= 0) {
          fid = t;
          break;
        }
        fid = a->fails[fid];
      }
      a->fails[cid] = fid;
      if (cedar_is_end(da, fid)) a->output[cid].next = fid;
      s = da->array[cid].sibling;
    }
  }
  free(q);
  return a;
}

/* AC.compile(keys) ac.cr:62-69 */
orc_ac *orc_ac_compile_keys(const uint8_t *blob, const uint64_t *offs, uint32_t K, int *err,
                            uint32_t *err_key) {
  orc_cedar *da = orc_cedar_new();
  for (uint32_t i = 0; i < K; i++) {
    int32_t kid = orc_cedar_insert(da, blob + offs[i], (int32_t)(offs[i + 1] - offs[i]));
    if (kid < 0 || (uint32_t)kid != i) {
      if (err) *err = kid < 0 ? kid : ORC_E_DUP_KEY;
      if (err_key) *err_key = i;
      orc_cedar_free(da);
      return NULL;
    }
  }
  if (err) *err = ORC_OK;
  return orc_ac_compile_cedar(da);
}

void orc_ac_free(orc_ac *a) {
  if (!a) return;
  orc_cedar_free(a->da);
  free(a->output);
  free(a->fails);
  free(a->key_lens);
  free(a);
}

int32_t orc_ac_slots(const orc_ac *a) { return a->da->array_size; }
int32_t orc_ac_keys(const orc_ac *a) { return a->da->leaf_size; }
uint32_t orc_ac_max_key_len(const orc_ac *a) { return a->max_len; }
int32_t orc_ac_key(const orc_ac *a, int32_t id, uint8_t *buf, int32_t cap) {
  return orc_cedar_key(a->da, id, buf, cap);
}
int32_t orc_ac_id(const orc_ac *a, const uint8_t *key, int32_t len) {
  return orc_cedar_get(a->da, key, len);
}

#define KEY_LEN_MASK 0x7FFFFFFFu /* ac.cr:237 */

/* BitArray lookup helper for match(seq, sep) ac.cr:324-336:
 * returns 1 when `chr < sep.size && !sep[chr]` (i.e. the hit is blocked). */
static inline int sep_blocks(const uint8_t *sep_bits, int32_t sep_size, uint8_t chr) {
  if ((int32_t)chr >= sep_size) return 0;
  return !((sep_bits[chr >> 3] >> (chr & 7)) & 1);
}

/* char_map matcher.cr:14-22 is materialised lazily as a running lead-byte
 * count: char_of_byte[p] = (#bytes b in seq[0..p] with (b&0xC0)!=0x80) - 1
 * for valid UTF-8.  The String overload (matcher.cr:34-39) maps
 * Hit(s,e,v) -> Hit(char_of_byte[s], char_of_byte[e-1]+1, v). */
static int32_t *build_char_map(const uint8_t *text, int64_t n) {
  int32_t *m = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
  int32_t ci = -1;
  for (int64_t p = 0; p < n; p++) {
    if ((text[p] & 0xC0) != 0x80) ci++;
    m[p] = ci;
  }
  return m;
}

int64_t orc_ac_match(const orc_ac *a, const uint8_t *text, int64_t n, int char_offsets,
                     const uint8_t *sep_bits, int32_t sep_size, orc_hit *out, int64_t cap) {
  if (sep_bits && sep_size > 256) return ORC_E_SEP_SIZE;
  const orc_cedar *da = a->da;
  const node_t *array = da->array;
  const int32_t array_size = da->array_size;
  int32_t *cmap = char_offsets ? build_char_map(text, n) : NULL;
  int64_t cnt = 0;
  int32_t nid = 0;
  /* match_ ac.cr:176-192 */
  for (int64_t i = 0; i < n; i++) {
    uint8_t b = text[i];
    for (;;) {
      /* child cedar.cr:441-447 (inlined) */
      int32_t base = nbase(&array[nid]);
      int32_t cid = base ^ b;
      int32_t nid_ = (cid < 0 || cid >= array_size || array[cid].check != nid) ? -1 : cid;
      if (nid_ >= 0) {
        nid = nid_;
        if ((array[nid].flags & END_MASK) || array[nid].child == 0) { /* is_end? :657-660 */
          /* match(seq, sep) right-neighbour test ac.cr:324-329 */
          if (sep_bits && i + 1 < n && sep_blocks(sep_bits, sep_size, text[i + 1])) break;
          /* fetch ac.cr:265-278 */
          const outnode_t *e = &a->output[nid];
          while (e->value >= 0) {
            int32_t val = e->value;
            if (a->key_lens[val] < KEY_LEN_MASK) {
              int32_t len = (int32_t)(a->key_lens[val] & KEY_LEN_MASK);
              int32_t s = (int32_t)i - len + 1;
              int32_t en = (int32_t)i + 1;
              /* left-neighbour test ac.cr:331-336 */
              int blocked = sep_bits && s > 0 && sep_blocks(sep_bits, sep_size, text[s - 1]);
              if (!blocked) {
                if (cnt < cap) {
                  if (cmap) {
                    out[cnt].start = cmap[s];
                    out[cnt].end = cmap[en - 1] + 1;
                  } else {
                    out[cnt].start = s;
                    out[cnt].end = en;
                  }
                  out[cnt].value = val;
                }
                cnt++;
              }
            }
            if (!(e->next >= 0)) break;
            e = &a->output[e->next];
          }
        }
        break;
      }
      if (nid == 0) break;
      nid = a->fails[nid];
      /* NUL-input contract: the reference dereferences
       * array[-1] on the byte after a NUL that reached a terminator node;
       * wherever it is defined, a NUL byte leaves the state at root without
       * emission.  The oracle pins that defined behaviour. */
      if (nid < 0) {
        nid = 0;
        break;
      }
    }
    if (b == 0) nid = 0;
  }
  free(cmap);
  return cnt;
}

/* fetch_one ac.cr:249-263 */
static inline void fetch_one(const orc_ac *a, int64_t idx, int32_t nid, const int32_t *cmap,
                             orc_hit *out, int64_t cap, int64_t *cnt) {
  const outnode_t *e = &a->output[nid];
  while (e->value >= 0) {
    int32_t val = e->value;
    if (a->key_lens[val] < KEY_LEN_MASK) {
      int32_t len = (int32_t)(a->key_lens[val] & KEY_LEN_MASK);
      int32_t s = (int32_t)idx - len + 1;
      int32_t en = (int32_t)idx + 1;
      if (*cnt < cap) {
        out[*cnt].start = cmap ? cmap[s] : s;
        out[*cnt].end = cmap ? cmap[en - 1] + 1 : en;
        out[*cnt].value = val;
      }
      (*cnt)++;
      break;
    }
    if (!(e->next >= 0)) break;
    e = &a->output[e->next];
  }
}

/* Nodes the compile BFS reached whose is_end? is true although they hold no key: stale END flags of slots reused
 * inside resolve (cedar.cr:642-648).  Unobservable through match; observable through match_longest (a stale node
 * replaces the pending end by one that yields nothing).  Test infrastructure: lets the parity tests tell documents
 * inside the GPU path's match_longest contract from those outside it. */
int32_t orc_ac_stale_ends(const orc_ac *a) {
  int32_t n = 0;
  for (int32_t id = 0; id < a->da->array_size; id++) {
    if (id != 0 && a->fails[id] < 0) continue; /* never visited */
    if (a->da->array[id].check < 0 && id != 0) continue;
    if (cedar_is_end(a->da, id) && a->output[id].value < 0) n++;
  }
  return n;
}

/* The same nodes as byte strings: for each one a 4-byte little-endian length, then the labels on its path from the
 * root (parent = check, label = id ^ base(parent): cedar.cr:441-447 read backwards).  Returns the bytes needed; writes
 * only when they fit cap.  Test infrastructure: the product derives the same set by its own replay of Cedar's inserts
 * (aha_amd/csrc/cedar_replay.cpp) and tests/test_host_logic.py compares set against set. */
int64_t orc_ac_stale_paths(const orc_ac *a, uint8_t *buf, int64_t cap) {
  int64_t need = 0;
  for (int pass = 0; pass < 2; pass++) {
    int64_t w = 0;
    for (int32_t id = 1; id < a->da->array_size; id++) {
      if (a->fails[id] < 0 || a->da->array[id].check < 0) continue;
      if (!(cedar_is_end(a->da, id) && a->output[id].value < 0)) continue;
      uint32_t len = 0;
      for (int32_t x = id; x != 0; x = a->da->array[x].check) len++;
      if (pass == 1) {
        memcpy(buf + w, &len, 4);
        int32_t x = id;
        for (uint32_t k = len; k > 0; k--) {
          const int32_t par = a->da->array[x].check;
          buf[w + 4 + k - 1] = (uint8_t)(x ^ nbase(&a->da->array[par]));
          x = par;
        }
      }
      w += 4 + len;
    }
    need = w;
    if (pass == 0 && (!buf || need > cap)) return need;
  }
  return need;
}

/* match_longest_ ac.cr:118-143 + match_longest :297-310 */
int64_t orc_ac_match_longest(const orc_ac *a, const uint8_t *text, int64_t n, int intersectable,
                             int char_offsets, orc_hit *out, int64_t cap) {
  const orc_cedar *da = a->da;
  int32_t *cmap = char_offsets ? build_char_map(text, n) : NULL;
  int64_t cnt = 0;
  int32_t nid = 0;
  int64_t prev_i = -1;
  int32_t prev_nid = -1;
  for (int64_t i = 0; i < n; i++) {
    uint8_t b = text[i];
    for (;;) {
      int32_t nid_ = cedar_child(da, nid, b);
      if (nid_ >= 0) {
        nid = nid_;
        if (cedar_is_end(da, nid)) {
          prev_i = i;
          prev_nid = nid;
        }
        break;
      }
      if (prev_i != -1) {
        fetch_one(a, prev_i, prev_nid, cmap, out, cap, &cnt);
        prev_i = -1;
        if (!intersectable) nid = 0;
      }
      if (nid == 0) break;
      nid = a->fails[nid];
      if (nid < 0) { /* NUL contract, as in orc_ac_match */
        nid = 0;
        break;
      }
    }
  }
  if (prev_i != -1) fetch_one(a, prev_i, prev_nid, cmap, out, cap, &cnt);
  free(cmap);
  return cnt;
}

int64_t orc_ac_match_batch(const orc_ac *a, const uint8_t *corpus, const uint64_t *doc_offsets,
                           uint64_t D, int char_offsets, orc_hit *out, int64_t cap,
                           uint64_t *doc_hit_offsets) {
  int64_t total = 0;
  for (uint64_t d = 0; d < D; d++) {
    if (doc_hit_offsets) doc_hit_offsets[d] = (uint64_t)total;
    int64_t room = cap > total ? cap - total : 0;
    int64_t c = orc_ac_match(a, corpus + doc_offsets[d], (int64_t)(doc_offsets[d + 1] - doc_offsets[d]),
                             char_offsets, NULL, 0, room ? out + total : out, room);
    total += c;
  }
  if (doc_hit_offsets) doc_hit_offsets[D] = (uint64_t)total;
  return total;
}
