// cedar_replay.cpp -- which states carry one of Cedar's STALE END flags?  (product code, host side)
//
// Why: Aha::AC#match_longest (src/aha/ac.cr:118-143) asks `@da.is_end? nid` (src/aha/cedar.cr:657-660) after every
// goto.  In the reference's Cedar a slot that `resolve` hands over directly from an evicted child to the node being
// inserted keeps the evicted node's `flags` word (cedar.cr:642-648 resets child, value and check only; push_enode
// :394-396 is the path that clears flags).  If the evicted node ended a key, the new path node answers is_end? = true
// although it holds no key: ac.cr:126-128 then replaces the pending longest end by one for which fetch_one
// (ac.cr:249-263) yields nothing.  Which nodes are hit is a property of Cedar's slot history -- block lists, free
// rings, the `reject` heuristics and the order of the inserts -- not of the key set, so the only way to reproduce the
// reference's match_longest bit for bit is to replay that history.  This file does exactly that and nothing else: it
// inserts the keys in compile order into a private model of the double array (AC.compile(keys), ac.cr:62-69, with
// CedarX.new's ordered = false), then walks the product's own automaton beside it and reports the states whose
// Cedar node has is_end? true and no value.  `match` is unaffected (a stale node ends an output chain exactly like
// a non-end node, ac.cr:267), so only the match_longest kernels read the result (DevAut::stale_bits).
//
// The model follows src/aha/cedar.cr: Node :29-96, Block :98-129, initialize :198-221, get :224-241, follow :244-259,
// pop_block / push_block / add_block / transfer_block :266-327, pop_enode :329-359, push_enode :361-397,
// push_sibling :403-417, consult :432-434, set_child :498-524, find_place(s) :526-580, resolve :582-655,
// is_end? :657-660, value :726-734, insert :751-778.  Crystal 0.23 integer arithmetic wraps; `base` of a fresh leaf
// (value = Int32::MAX, :355) relies on it (:81-83).
#include "cedar_replay.hpp"

#include <array>
#include <cstring>

namespace aha {

namespace {

constexpr int32_t kFresh = INT32_MAX;  // CedarX.value_limit (cedar.cr:10-12): an allocated node without value or base
constexpr uint16_t kEndFlag = 1u << 9;  // Node::END_MASK (:63)
constexpr uint16_t kCountMask = (1u << 9) - 1;  // Node::CHILD_NUM_MASK (:62)

struct Cell {  // one slot of @array
  int32_t value;
  int32_t check;
  uint8_t sibling = 0;
  uint8_t child = 0;
  uint16_t flags = 0;
  int32_t base() const { return (int32_t)(0u - ((uint32_t)value + 1u)); }  // -(value + 1), wrapping (:81-83)
};

struct Ring {  // one 256-slot block (:98-129)
  int32_t prev = 0, next = 0;
  int32_t num = 256, reject = 257, trial = 0;
  int32_t ehead = 0;
};

class Replay {
 public:
  Replay() {
    cells_.resize(256);
    cells_[0] = Cell{-1, -1};
    for (int32_t i = 1; i < 256; i++) cells_[i] = Cell{-(i - 1), -(i + 1)};
    cells_[1].value = -255;
    cells_[255].check = -1;
    rings_.resize(1);
    rings_[0].ehead = 1;
    for (int i = 0; i <= 256; i++) reject_[i] = i + 1;
  }

  // AC.compile(keys): da.insert(key) for every key, ids in order (the caller has already rejected empty keys, NUL
  // bytes and duplicates at the reference's index, so the lookup in front of insert (:755-757) never finds the key)
  void insert(const uint8_t *key, uint32_t len) {
    int32_t from = 0;
    for (uint32_t pos = 0; pos < len; pos++) {  // get (:224-241)
      const int32_t v = cells_[from].value;
      if (v >= 0 && v != kFresh) {  // a leaf gains a child: its value moves to a label-0 child
        const int32_t to = follow(from, 0);
        cells_[to].value = v;
      }
      from = follow(from, key[pos]);
    }
    const int32_t p = cells_[from].value < 0 ? follow(from, 0) : from;
    cells_[p].value = n_keys_++;
    cells_[p].flags |= kEndFlag;
  }

  int32_t child(int32_t id, uint8_t label) const {  // :441-447
    const int32_t cid = cells_[id].base() ^ (int32_t)label;
    if (cid < 0 || cid >= (int32_t)cells_.size() || cells_[cid].check != id) return -1;
    return cid;
  }
  bool is_end(int32_t id) const { return (cells_[id].flags & kEndFlag) != 0 || cells_[id].child == 0; }  // :657-660
  int32_t value(int32_t id) const {  // :726-734
    const int32_t v = cells_[id].value;
    if (v >= 0) return v;
    const int32_t to = cells_[id].base();
    if (to >= 0 && to < (int32_t)cells_.size() && cells_[to].check == id && cells_[to].value >= 0 &&
        cells_[to].value != kFresh)
      return cells_[to].value;
    return -1;
  }

 private:
  std::vector<Cell> cells_;
  std::vector<Ring> rings_;
  int32_t reject_[257];
  int32_t head_full_ = 0, head_closed_ = 0, head_open_ = 0;
  int32_t n_keys_ = 0;
  static constexpr int32_t kMaxTrial = 1;

  int32_t child_count(int32_t id) const { return cells_[id].flags & kCountMask; }
  // child_num= (:65-67) does not mask the new value: a count of 512 (a stale 256 plus 256 own children) would run into
  // the END bit, exactly as in the reference
  void set_child_count(int32_t id, int32_t n) {
    cells_[id].flags = (uint16_t)((cells_[id].flags & ~kCountMask) | (uint16_t)n);
  }

  // ---- block lists (:266-327) ----
  void unlink_ring(int32_t bi, int32_t &head, bool last) {
    if (last) {
      head = 0;
      return;
    }
    Ring &b = rings_[bi];
    rings_[b.prev].next = b.next;
    rings_[b.next].prev = b.prev;
    if (bi == head) head = b.next;
  }
  void link_ring(int32_t bi, int32_t &head, bool empty) {
    Ring &b = rings_[bi];
    if (empty) {
      head = b.prev = b.next = bi;
      return;
    }
    Ring &tail_of = rings_[head];
    b.prev = tail_of.prev;
    b.next = head;
    rings_[tail_of.prev].next = bi;
    tail_of.prev = bi;
    head = bi;
  }
  void move_ring(int32_t bi, int32_t &from, int32_t &to) {
    unlink_ring(bi, from, bi == rings_[bi].next);
    link_ring(bi, to, to == 0 && rings_[bi].num != 0);
  }
  int32_t grow() {
    const int32_t at = (int32_t)cells_.size();
    cells_.resize((size_t)at + 256);
    rings_.emplace_back();
    rings_.back().ehead = at;
    for (int32_t i = 0; i < 256; i++) cells_[at + i] = Cell{-(((i + 255) & 255) + at), -(((i + 1) & 255) + at)};
    link_ring(at >> 8, head_open_, head_open_ == 0);
    return at >> 8;
  }

  // ---- free slots (:329-397) ----
  int32_t any_place() {  // find_place (:526-530)
    if (head_closed_ != 0) return rings_[head_closed_].ehead;
    if (head_open_ != 0) return rings_[head_open_].ehead;
    return grow() << 8;
  }
  int32_t take(int32_t base, uint8_t label, int32_t from) {  // pop_enode
    const int32_t e = base < 0 ? any_place() : (base ^ (int32_t)label);
    const int32_t bi = e >> 8;
    Ring &b = rings_[bi];
    b.num--;
    if (b.num == 0) {
      if (bi != 0) move_ring(bi, head_closed_, head_full_);
    } else {
      Cell &n = cells_[e];
      cells_[-n.value].check = n.check;
      cells_[-n.check].value = n.value;
      if (e == b.ehead) b.ehead = -n.check;
      if (bi != 0 && b.num == 1 && b.trial != kMaxTrial) move_ring(bi, head_open_, head_closed_);
    }
    cells_[e].value = kFresh;
    cells_[e].check = from;
    if (base < 0) cells_[from].value = -(e ^ (int32_t)label) - 1;
    return e;
  }
  void give_back(int32_t e) {  // push_enode
    const int32_t bi = e >> 8;
    Ring &b = rings_[bi];
    b.num++;
    if (b.num == 1) {
      b.ehead = e;
      cells_[e].value = -e;
      cells_[e].check = -e;
      if (bi != 0) move_ring(bi, head_full_, head_closed_);
    } else {
      const int32_t prev = b.ehead;
      const int32_t next = -cells_[prev].check;
      cells_[e].value = -prev;
      cells_[e].check = -next;
      cells_[prev].check = -e;
      cells_[next].value = -e;
      if (b.num == 2 || b.trial == kMaxTrial) {
        if (bi != 0) move_ring(bi, head_closed_, head_open_);
      }
      b.trial = 0;
    }
    if (b.reject < reject_[b.num]) b.reject = reject_[b.num];
    cells_[e].child = 0;
    cells_[e].sibling = 0;
    cells_[e].flags = 0;
  }

  // ---- sibling lists (:403-417, ordered = false) ----
  void add_sibling(int32_t from, int32_t base, uint8_t label, bool has_child) {
    uint8_t *link = &cells_[from].child;
    if (has_child && *link == 0) link = &cells_[base ^ (int32_t)*link].sibling;  // keep the label-0 child first
    cells_[base ^ (int32_t)label].sibling = *link;
    *link = label;
    set_child_count(from, child_count(from) + 1);
  }

  int32_t follow(int32_t from, uint8_t label) {  // :244-259
    const int32_t base = cells_[from].base();
    int32_t to = base ^ (int32_t)label;
    if (base < 0 || cells_[to].check < 0) {
      const bool has_child = base >= 0 && cells_[base ^ (int32_t)cells_[from].child].check == from;
      to = take(base, label, from);
      add_sibling(from, to ^ (int32_t)label, label, has_child);
    } else if (cells_[to].check != from) {
      to = relocate(from, base, label);
    }
    return to;
  }

  // labels of a family in list order, the label-0 child first; optionally with the new label (:498-524, unordered)
  int collect(int32_t base, uint8_t c, uint8_t label, bool with_label, uint8_t *out) const {
    int n = 0;
    if (c == 0) {
      out[n++] = c;
      c = cells_[base ^ (int32_t)c].sibling;
    }
    if (with_label) out[n++] = label;
    while (c != 0) {
      out[n++] = c;
      c = cells_[base ^ (int32_t)c].sibling;
    }
    return n;
  }

  int32_t places_for(const uint8_t *labels, int n) {  // find_places (:533-580)
    int32_t bi = head_open_;
    if (bi != 0) {
      const int32_t last = rings_[head_open_].prev;
      for (;;) {
        Ring &b = rings_[bi];
        if (b.num >= n && n < b.reject) {
          int32_t e = b.ehead;
          for (;;) {
            const int32_t base = e ^ (int32_t)labels[0];
            bool fits = true;
            for (int i = 0; i < n && fits; i++) fits = cells_[base ^ (int32_t)labels[i]].check < 0;
            if (fits) {
              b.ehead = e;
              return e;
            }
            e = -cells_[e].check;
            if (e == b.ehead) break;
          }
        }
        b.reject = n;
        if (b.reject < reject_[b.num]) reject_[b.num] = b.reject;
        const int32_t next = b.next;
        b.trial++;
        if (b.trial == kMaxTrial) move_ring(bi, head_open_, head_closed_);
        if (bi == last) break;
        bi = next;
      }
    }
    return grow() << 8;
  }

  // resolve (:582-655): the slot base_n ^ label_n belongs to another parent; the family with fewer children moves
  int32_t relocate(int32_t from_n, int32_t base_n, uint8_t label_n) {
    const int32_t wanted = base_n ^ (int32_t)label_n;
    const int32_t from_p = cells_[wanted].check;
    const int32_t base_p = cells_[from_p].base();
    const bool move_new = child_count(from_n) < child_count(from_p);  // consult (:432-434)
    std::array<uint8_t, 257> labels{};
    const int n = move_new ? collect(base_n, cells_[from_n].child, label_n, true, labels.data())
                           : collect(base_p, cells_[from_p].child, 255, false, labels.data());
    int32_t base = (n == 1 ? any_place() : places_for(labels.data(), n)) ^ (int32_t)labels[0];
    const int32_t from = move_new ? from_n : from_p;
    const int32_t old_base = move_new ? base_n : base_p;
    if (move_new && labels[0] == label_n) cells_[from].child = label_n;
    cells_[from].value = -base - 1;
    for (int i = 0; i < n; i++) {
      const uint8_t l = labels[i];
      const int32_t to = take(base, l, from);
      const int32_t was = old_base ^ (int32_t)l;
      cells_[to].sibling = i == n - 1 ? 0 : labels[i + 1];
      if (move_new && was == wanted) continue;  // the new node itself: nothing to carry over
      cells_[to].value = cells_[was].value;
      cells_[to].flags = cells_[was].flags;
      if (cells_[to].value < 0 && l != 0) {  // re-parent the grandchildren
        uint8_t c = cells_[was].child;
        cells_[to].child = c;
        const int32_t gb = cells_[to].base();
        do {
          Cell &g = cells_[gb ^ (int32_t)c];
          g.check = to;
          c = g.sibling;
        } while (c != 0);
      }
      if (!move_new && was == from_n) from_n = to;
      if (!move_new && was == wanted) {
        // the evicted child's slot goes straight to the new node: child, value and check are reset -- flags are NOT
        // (:642-648).  This is where a stale END flag (and a stale child count) is born.
        add_sibling(from_n, wanted ^ (int32_t)label_n, label_n, true);
        cells_[was].child = 0;
        cells_[was].value = kFresh;
        cells_[was].check = from_n;
      } else {
        give_back(was);
      }
    }
    return move_new ? (base ^ (int32_t)label_n) : wanted;
  }
};

}  // namespace

void cedar_stale_ends(const Automaton &a, std::vector<uint32_t> &stale_states) {
  stale_states.clear();
  Replay da;
  for (uint32_t k = 0; k < a.n_keys; k++)
    da.insert(a.blob.data() + a.offs[k], (uint32_t)(a.offs[k + 1] - a.offs[k]));
  // the product's states are numbered breadth-first with consecutive children: one pass pairs every state with its
  // Cedar node (both tries hold exactly the prefixes of the keys)
  std::vector<int32_t> node(a.n_states, -1);
  node[0] = 0;
  for (uint32_t s = 0; s < a.n_states; s++) {
    const int32_t id = node[s];
    if (id < 0) continue;  // cannot happen: every prefix of a key is a Cedar path
    if (s != 0 && da.is_end(id) && da.value(id) < 0) stale_states.push_back(s);
    for (uint32_t j = 0; j < a.n_child[s]; j++) {
      const uint32_t c = a.first_child[s] + j;
      node[c] = da.child(id, a.in_label[c]);
    }
  }
}

}  // namespace aha
