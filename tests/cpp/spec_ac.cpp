// C++ twin of the reference's spec/ac_spec.cr, run against libaha_hip.so
// through include/aha/ac.hpp (built and executed by tests/test_cpp_wrapper.py
// on the GPU box).
#include <cstdio>
#include <utility>
#include <vector>

#include "aha/ac.hpp"

using Pair = std::pair<int, int>;
static int fails = 0;
static void expect(const char *name, const std::vector<Pair> &got, const std::vector<Pair> &want) {
  if (got != want) {
    fails++;
    std::printf("FAIL %s: got", name);
    for (auto &p : got) std::printf(" {%d,%d}", p.first, p.second);
    std::printf("\n");
  } else {
    std::printf("ok   %s\n", name);
  }
}

int main() {
  {  // it "ac"  spec/ac_spec.cr:5-12
    auto matcher = aha::AC::compile({"我", "我是", "是中"});
    std::vector<Pair> matched;
    matcher.match_string("我是中国人", [&](const aha::Hit &hit) { matched.push_back({hit.end, hit.value}); });
    expect("ac", matched, {{1, 0}, {2, 1}, {3, 2}});
    // byte-level triples (Bytes overload)
    std::vector<Pair> b;
    matcher.match("我是中国人", [&](const aha::Hit &hit) { b.push_back({hit.start, hit.end}); });
    expect("ac bytes", b, {{0, 3}, {0, 6}, {3, 9}});
    if (matcher[1] != "我是" || matcher["是中"] != 2) {
      fails++;
      std::printf("FAIL []\n");
    }
  }
  {  // it "ac save load"  spec/ac_spec.cr:14-23 (here the LOADED automaton is the one matched)
    auto matcher = aha::AC::compile({"我", "我是", "是中"});
    auto loaded = aha::AC::from_bytes(matcher.to_bytes());
    std::vector<Pair> matched;
    loaded.match_string("我是中国人", [&](const aha::Hit &hit) { matched.push_back({hit.end, hit.value}); });
    expect("ac save load", matched, {{1, 0}, {2, 1}, {3, 2}});
  }
  {  // it "ac with sep"  spec/ac_spec.cr:25-34
    auto matcher = aha::AC::compile({"a", "aa"});
    aha::BitArray sep(256);
    sep.set(' ');
    std::vector<Pair> matched;
    matcher.match("a aaa", sep, [&](const aha::Hit &hit) { matched.push_back({hit.end, hit.value}); }, true);
    expect("ac with sep", matched, {{1, 0}});
  }
  {  // spec/ac_longest_match_spec.cr:5-18 and :20-33
    auto m1 = aha::AC::compile({"Ruby", "ruby", "rub"});
    std::vector<Pair> got;
    m1.match_longest("Ruby on rub", false, [&](const aha::Hit &hit) { got.push_back({hit.start, hit.end}); });
    expect("match_longest", got, {{0, 4}, {8, 11}});
    auto m2 = aha::AC::compile({"Ruby", "ruby", "uby "});
    got.clear();
    m2.match_longest("ruby ", true, [&](const aha::Hit &hit) { got.push_back({hit.start, hit.end}); });
    expect("match_longest intersectable", got, {{0, 4}, {1, 5}});
  }
  {  // a batch of documents: the state is per sequence (ac.cr:177)
    auto matcher = aha::AC::compile({"ab", "b"});
    std::vector<uint64_t> dho;
    auto hits = matcher.match_batch("abab", {0, 1, 1, 4}, &dho);  // "a", "", "bab"
    std::vector<Pair> got;
    for (auto &h : hits) got.push_back({h.end, h.value});
    expect("match_batch", got, {{1, 1}, {3, 0}, {3, 1}});
    if (dho != std::vector<uint64_t>{0, 0, 0, 3}) {
      fails++;
      std::printf("FAIL match_batch offsets\n");
    }
  }
  {  // the same batch resident in HBM: uploaded once through the C ABI (aha_corpus_upload), matched by the device
     // entry point on raw device pointers, hits downloaded with aha_buffer_download -- no GPU framework involved
    auto matcher = aha::AC::compile({"ab", "b"});
    aha::Corpus resident("abab", {0, 1, 1, 4});
    for (int pass = 0; pass < 2; pass++) {  // matched twice: the upload is not repeated
      std::vector<uint64_t> dho;
      auto hits = matcher.match_corpus(resident, &dho);
      std::vector<Pair> got;
      for (auto &h : hits) got.push_back({h.end, h.value});
      expect("match_corpus (device resident)", got, {{1, 1}, {3, 0}, {3, 1}});
      if (dho != std::vector<uint64_t>{0, 0, 0, 3}) {
        fails++;
        std::printf("FAIL match_corpus offsets\n");
      }
    }
    auto other = aha::AC::compile({"a"});  // another automaton over the same resident batch
    auto hits = other.match_corpus(resident);
    std::vector<Pair> got;
    for (auto &h : hits) got.push_back({h.end, h.value});
    expect("match_corpus, second automaton", got, {{1, 0}, {2, 0}});
    try {
      aha::Corpus bad("abab", {0, 3, 2, 4});
      fails++;
      std::printf("FAIL corpus offsets: no exception\n");
    } catch (const aha::Error &e) {
      std::printf("ok   corpus offsets rejected\n");
    }
  }
  {  // error behaviour: raise "key:... appear twice."  ac.cr:66
    try {
      aha::AC::compile({"ab", "cd", "ab"});
      fails++;
      std::printf("FAIL dup: no exception\n");
    } catch (const aha::Error &e) {
      if (std::string(e.what()) != "key:ab appear twice." || e.key_index != 2) {
        fails++;
        std::printf("FAIL dup message: %s\n", e.what());
      } else {
        std::printf("ok   dup key\n");
      }
    }
  }
  return fails ? 1 : 0;
}
