// proto_pp.hip -- PROTOTYPE (not part of the product path): measures how fast a
// position-parallel "shallow classify" pass can run on MI355X.
//
// Observation: for boundary depth d0 = 2 the shallow AC state
// (longest suffix of depth <= 2 that is a trie path) is a pure function of the
// last two bytes, so it can be computed for every position independently:
// coalesced 16-byte loads, no chunks, no warm-up, no per-lane positions.  The
// pass then (1) counts shallow END events, (2) compacts the positions whose
// state is a boundary (depth-2) state and (3) asks the lookahead Bloom filter
// about the next 3 bytes on the compacted list only (all lanes busy).
// This file only COUNTS events and suspects -- enough to time the idea and to
// check the counts against a numpy model (tools/proto_pp.py).
//
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/libproto_pp.so tools/proto_pp.hip
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../aha_amd/csrc/automaton.hpp"

using namespace aha;

namespace {
constexpr int kThreads = 1024;  // 16 waves per CU; rows + a 64 KiB filter + per-wave staging fit 160 KiB of LDS
constexpr int kTile = 1024;  // bytes per wave tile (64 lanes x 16 B)

__device__ __forceinline__ bool bloom_test(const uint32_t *flt, uint32_t words, uint32_t B, uint32_t w) {
  const uint32_t h = filter_hash(B, w);
  const uint32_t m = filter_mask(h);
  return (flt[filter_word(h, words)] & m) == m;
}

__global__ __launch_bounds__(kThreads) void k_pp(const uint32_t *rows_g, uint32_t t_rows, const uint32_t *bloom_g,
                                                  uint32_t words, const uint8_t *text, uint64_t n,
                                                  unsigned long long *counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t *rows = reinterpret_cast<uint32_t *>(smem);
  uint32_t *flt = rows + t_rows;
  uint8_t *wbase = reinterpret_cast<uint8_t *>(flt + words);
  for (uint32_t i = threadIdx.x; i < t_rows; i += kThreads) rows[i] = rows_g[i];
  for (uint32_t i = threadIdx.x; i < words; i += kThreads) flt[i] = bloom_g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per wave: tile text copy (kTile + 16 halo) and the compacted boundary list (kTile x 8 B)
  uint8_t *ttext = wbase + (size_t)wave * (kTile + 16 + kTile * 4);
  uint32_t *blist = reinterpret_cast<uint32_t *>(ttext + kTile + 16);  // pos << 22 | boundary base
  const uint64_t n_tiles = (n + kTile - 1) / kTile;
  const uint64_t wave_id = (uint64_t)blockIdx.x * (kThreads / 64) + wave;
  const uint64_t n_waves = (uint64_t)gridDim.x * (kThreads / 64);
  unsigned long long n_ev = 0, n_bnd = 0, n_sus = 0;
  for (uint64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
    const uint64_t t0 = tile * kTile;
    const uint64_t g = t0 + (uint64_t)lane * 16;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (g + 16 <= n) w = *reinterpret_cast<const uint4 *>(text + g);
    // previous byte of this lane's first position (lane-1's last byte; tile edge from memory)
    uint32_t prevb = __shfl_up(w.w >> 24, 1, 64);
    if (lane == 0) prevb = t0 ? text[t0 - 1] : 0u;
    // stage the tile (+ 3 lookahead bytes) in LDS for the filter phase
    *reinterpret_cast<uint4 *>(ttext + lane * 16) = w;
    if (lane < 4) ttext[kTile + lane] = (t0 + kTile + lane < n) ? text[t0 + kTile + lane] : 0;
    // ---- phase 1: shallow state of every position = f(previous byte, byte)
    const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
    uint32_t e1prev = prevb ? rows[prevb] : 0u;  // root base is 0: depth-1 entry of the previous byte
    bool e1prev_ok = prevb != 0 && (e1prev & 0xFFu) == prevb;
    uint32_t bmask = 0, ubase[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t b = (wd[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
      const uint32_t e1 = rows[b];
      const bool ok1 = b != 0 && (e1 & 0xFFu) == b;
      const uint32_t pbase = (e1prev >> C_BASE_SHIFT) & C_BASE_MASK;
      const uint32_t idx2 = pbase ^ b;
      const uint32_t e2 = (e1prev_ok && idx2 < t_rows) ? rows[idx2] : 0u;
      const bool ok2 = e1prev_ok && b != 0 && (e2 & 0xFFu) == b;  // depth-2 (boundary) state
      const uint32_t ex = ok2 ? e2 : (ok1 ? e1 : 0u);
      n_ev += (ex & C_END) ? 1u : 0u;
      bmask |= ok2 ? (1u << j) : 0u;
      ubase[j] = (e2 >> C_BASE_SHIFT) & C_BASE_MASK;
      e1prev = e1;
      e1prev_ok = ok1;
    }
    // ---- compaction of boundary positions (wave prefix over popcounts)
    const uint32_t cnt = __popc(bmask);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t wpos = incl - cnt;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (bmask & (1u << j)) blist[wpos++] = (((uint32_t)lane * 16 + j) << 22) | ubase[j];
    }
    n_bnd += cnt;
    // ---- phase 2: lookahead filter on the dense list
    for (uint32_t i = lane; i < total; i += 64) {
      const uint32_t it = blist[i];
      const uint32_t p = it >> 22, ub = it & C_BASE_MASK;  // boundary state after byte p: next bytes p+1..p+3
      const uint32_t x1 = ttext[p + 1], x2 = ttext[p + 2], x3 = ttext[p + 3];
      const bool s = bloom_test(flt, words, ub, filter_key(1, x1, 0, 0)) ||
                     bloom_test(flt, words, ub, filter_key(2, x1, x2, 0)) ||
                     bloom_test(flt, words, ub, filter_key(3, x1, x2, x3));
      n_sus += s ? 1u : 0u;
    }
  }
  // wave reduce and publish
  for (int d = 32; d >= 1; d >>= 1) {
    n_ev += __shfl_down(n_ev, d, 64);
    n_bnd += __shfl_down(n_bnd, d, 64);
    n_sus += __shfl_down(n_sus, d, 64);
  }
  if (lane == 0) {
    atomicAdd(counts + 0, n_ev);
    atomicAdd(counts + 1, n_bnd);
    atomicAdd(counts + 2, n_sus);
  }
}
}  // namespace

extern "C" int proto_pp_run(const uint32_t *rows, uint32_t t_rows, const uint32_t *bloom, uint32_t words,
                            const uint8_t *text, uint64_t n, unsigned long long *counts, int grid, void *stream) {
  const size_t lds = (size_t)t_rows * 4 + (size_t)words * 4 + (size_t)(kThreads / 64) * (kTile + 16 + kTile * 4);
  hipError_t e = hipFuncSetAttribute((const void *)k_pp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_pp, dim3(grid), dim3(kThreads), lds, (hipStream_t)stream, rows, t_rows, bloom, words, text, n,
                     counts);
  return (int)hipGetLastError();
}
