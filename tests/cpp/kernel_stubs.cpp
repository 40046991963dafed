// kernel_stubs.cpp -- link-time stand-ins for the HIP kernel launchers, used ONLY by the sanitizer build of the
// host side (aha_amd/csrc/Makefile, target asan): automaton.cpp, capi.cpp and group.cpp are compiled with
// g++ -fsanitize=address,undefined and exercised under AHA_OPT_HOST_ONLY, where no launcher is ever reached.
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime_api.h>

#include "../../aha_amd/csrc/image.hpp"

namespace aha {
[[noreturn]] static void no_gpu(const char *what) {
  fprintf(stderr, "sanitizer build: %s reached (host-only library)\n", what);
  abort();
}
size_t v2_lds_bytes(uint32_t, bool) { return 0; }
size_t unit_lds_bytes(uint32_t) { return 0; }
int unit_prepare(uint32_t) { return 0; }
void unit_launch_traverse(const UnitDev &, const V2Args &, uint32_t, void *) { no_gpu("unit_launch_traverse"); }
int skip_prepare(uint32_t, uint32_t) { return 0; }
size_t skip_bitmap_bytes(uint64_t) { return 0; }
void skip_launch_mark(const SkipDev &, const V2Args &, void *, uint32_t, void *) { no_gpu("skip_launch_mark"); }
void skip_launch_traverse(const UnitDev &, const V2Args &, const void *, uint32_t, void *) { no_gpu("skip_launch_traverse"); }
int pair_prepare(uint32_t, uint32_t, uint32_t) { return 0; }
uint32_t pair_tile_bytes() { return 2048; }
size_t pair_cand_bytes(uint64_t) { return 0; }
size_t pair_walk_bytes(uint64_t) { return 0; }
void pair_launch(const PairDev &, const UnitDev &, const DevAut &, const V2Args &, void *, void *, void *, uint64_t, uint32_t, void *, void *) { no_gpu("pair_launch"); }
int v2_prepare(bool, size_t) { return 0; }
void v2_launch_traverse(const DevAut &, const V2Args &, uint32_t, void *) { no_gpu("v2_launch_traverse"); }
void v2_launch_chunk_scan(const V2Args &, void *) { no_gpu("v2_launch_chunk_scan"); }
void v2_launch_sort(const DevAut &, const V2Args &, uint64_t, void *) { no_gpu("v2_launch_sort"); }
void v2_launch_expand(const DevAut &, const V2Args &, uint64_t, void *) { no_gpu("v2_launch_expand"); }
void v2_launch_direct_post(const DevAut &, const V2Args &, void *, void *, bool) { no_gpu("v2_launch_direct_post"); }
void launch_has_nul(const uint8_t *, uint64_t, uint64_t *, void *) { no_gpu("launch_has_nul"); }
bool filter_image_in_lds(uint32_t, uint32_t, bool) { return false; }
int filter_prepare() { return 0; }
void filter_launch_filter(const FilterDev &, const V2Args &, void *, void *, unsigned long long *, uint32_t, void *) { no_gpu("filter_launch_filter"); }
size_t filter_chunk_rec_bytes() { return 16; }
void filter_launch_walk(const DevAut &, const V2Args &, const void *, const void *, const unsigned long long *, uint32_t, void *) { no_gpu("filter_launch_walk"); }
void unit_launch_regroup(const DevAut &, const V2Args &, void *) { no_gpu("unit_launch_regroup"); }
void unit_launch_expand(const uint2 *, const DevAut &, const V2Args &, uint32_t, void *) { no_gpu("unit_launch_expand"); }
void v2_launch_hit_scan(const V2Args &, void *) { no_gpu("v2_launch_hit_scan"); }
void v2_launch_lead_scan(const V2Args &, void *) { no_gpu("v2_launch_lead_scan"); }
void launch_hits_pack(const int32_t *, uint64_t, int32_t *, void *) { no_gpu("launch_hits_pack"); }
void launch_hits_unpack(const DevAut &, const int32_t *, uint64_t, int, int32_t *, void *) { no_gpu("launch_hits_unpack"); }
void launch_hits_pack4(const int32_t *, uint64_t, uint32_t *, unsigned long long *, StreamFmt, void *) { no_gpu("launch_hits_pack4"); }
void launch_hits_unpack4(const DevAut &, const uint32_t *, uint64_t, int, int32_t *, StreamFmt, void *) { no_gpu("launch_hits_unpack4"); }
void launch_hits_unpack4_segs(const DevAut &, const uint32_t *, const uint64_t *, const uint64_t *, const uint64_t *, uint32_t,
                              int, int32_t *, StreamFmt, void *) {
  no_gpu("launch_hits_unpack4_segs");
}
void launch_check_docs(const uint64_t *, uint64_t, uint64_t, uint32_t *, unsigned long long *, void *) { no_gpu("launch_check_docs"); }
void launch_publish_words(const unsigned long long *, unsigned long long *, int, void *) { no_gpu("launch_publish_words"); }
void launch_count(const DevAut &, const MatchArgs &, void *) { no_gpu("launch_count"); }
void launch_scan_blocks(const MatchArgs &, uint64_t, void *) { no_gpu("launch_scan_blocks"); }
void launch_docg(const MatchArgs &, void *) { no_gpu("launch_docg"); }
void launch_write(const DevAut &, const MatchArgs &, void *) { no_gpu("launch_write"); }
void launch_longest(const DevAut &, const MatchArgs &, int, bool, void *) { no_gpu("launch_longest"); }
}  // namespace aha
