// Lab: the expansion's store pattern without the expansion.  One wave per block (7.4 KiB of LDS: 21 blocks a CU, as k2d_expand),
// a block per stream of `run` bytes; the wave alternates `delay` busy cycles of ALU work with one burst of 6 KiB (six 16-byte
// stores a lane) and, optionally, waits for the burst's acknowledgement before it goes on (s_waitcnt vmcnt(0)) -- what the
// in-order vmcnt forces on a wave that also has loads to wait for.
// hipcc -O3 --offload-arch=gfx950 -o burst_rate burst_rate.hip && ./burst_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool WAIT>
__global__ __launch_bounds__(64) void burst(uint4 *out, uint32_t n_streams, uint32_t windows, uint32_t delay, uint32_t *sink) {
  __shared__ uint32_t pad[1860];
  const int lane = threadIdx.x;
  pad[lane] = lane;
  uint32_t acc = lane;
  for (uint32_t sidx = blockIdx.x; sidx < n_streams; sidx += gridDim.x) {
    uint4 *dst = out + (uint64_t)sidx * windows * 384;
    for (uint32_t w = 0; w < windows; w++) {
      for (uint32_t d = 0; d < delay; d++) acc = acc * 1664525u + 1013904223u;  // dependent ALU chain: ~8 cycles an iteration
#pragma unroll
      for (int k = 0; k < 6; k++) dst[(uint64_t)w * 384 + k * 64 + lane] = make_uint4(acc, w, k, lane);
      if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (acc == 0x12345u) sink[0] = acc + pad[(lane + 1) & 63];
}


// the same with the expansion's own steps around the burst: MODE bit 0 = the window staged in LDS (24 ds_write_b32 a lane at a
// 12-byte stride, six ds_read_b128 back), bit 1 = eight 8-byte gathers a lane from a table of `tab_n` entries, runs of ~9
// consecutive entries per "event" (the flattened chains), consumed before the burst
template <int MODE>
__global__ __launch_bounds__(64) void burst2(uint4 *out, uint32_t n_streams, uint32_t windows, const uint2 *tab, uint32_t tab_n, uint32_t *sink) {
  __shared__ __attribute__((aligned(16))) uint32_t hbuf[1540 + 320];
  const int lane = threadIdx.x;
  uint32_t acc = lane;
  for (uint32_t sidx = blockIdx.x; sidx < n_streams; sidx += gridDim.x) {
    uint4 *dst = out + (uint64_t)sidx * windows * 384;
    for (uint32_t w = 0; w < windows; w++) {
      uint2 ce[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t j = (uint32_t)lane + 64u * k;           // hit index in the window
        const uint32_t ev = (sidx * 977u + w * 131u + j / 9u) * 2654435761u;  // its event: nine hits each
        if (MODE & 2) ce[k] = tab[(ev % (tab_n - 16)) + j % 9u];
        else ce[k] = make_uint2(ev, j);
      }
      if (MODE & 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const uint32_t j = (uint32_t)lane + 64u * k;
          hbuf[j * 3] = ce[k].x;
          hbuf[j * 3 + 1] = acc;
          hbuf[j * 3 + 2] = ce[k].y;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const uint4 v = reinterpret_cast<const uint4 *>(hbuf)[k * 64 + lane];
          if (MODE & 8) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u dv = {v.x, v.y, v.z, v.w};
            uint4 *ap = dst + (uint64_t)w * 384 + k * 64 + lane;
            asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ap), "v"(dv) : "memory");
          } else if (MODE & 16) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u dv = {v.x, v.y, v.z, v.w};
            uint4 *ap = dst + (uint64_t)w * 384 + k * 64 + lane;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ap), "v"(dv) : "memory");
          } else if (!(MODE & 4)) dst[(uint64_t)w * 384 + k * 64 + lane] = v;
          else acc ^= v.x ^ v.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
#pragma unroll
        for (int k = 0; k < 6; k++) dst[(uint64_t)w * 384 + k * 64 + lane] = make_uint4(ce[k].x, ce[k].y, ce[k + 2].x, ce[k + 2].y);
      }
      acc += ce[7].x;
    }
  }
  if (acc == 0x12345u) sink[0] = acc;
}

int main() {
  const uint32_t windows = 14;                       // 84 KiB a stream (cfg 5: ~87 KB of hits per chunk)
  const uint32_t n_streams = 131072;                 // 11.0 GB
  const uint64_t bytes = (uint64_t)n_streams * windows * 6144;
  void *a; uint32_t *sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(a, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wait = 0; wait < 2; wait++)
    for (uint32_t delay : {0u, 150u, 300u, 600u, 1200u}) {
      float best = 1e9f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        if (wait) hipLaunchKernelGGL(burst<true>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, delay, sink);
        else hipLaunchKernelGGL(burst<false>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, delay, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      // the ALU work alone: delay iterations * windows * streams / (20 waves on 4 SIMDs ...) -- measured by the delay-only run below
      printf("wait %d  delay %4u iterations/window: %7.3f ms  %7.1f GB/s\n", wait, delay, best, (double)bytes / best * 1e3 / 1e9);
    }
  uint2 *tab; if (hipMalloc(&tab, (size_t)(32u << 20) * 8) != hipSuccess) return 1;
  hipMemset(tab, 1, (size_t)(32u << 20) * 8);
  for (uint32_t tab_n : {1u << 17, 1u << 20, 5u << 20, 32u << 20})   // 1 MB, 8 MB, 40 MB, 256 MB of 8-byte entries
    for (int mode : {3, 7, 11, 19}) {
      float best = 1e9f;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        if (mode == 3) hipLaunchKernelGGL(burst2<3>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, tab, tab_n, sink);
        else if (mode == 7) hipLaunchKernelGGL(burst2<7>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, tab, tab_n, sink);
        else if (mode == 11) hipLaunchKernelGGL(burst2<11>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, tab, tab_n, sink);
        else hipLaunchKernelGGL(burst2<19>, dim3(n_streams), dim3(64), 0, 0, (uint4 *)a, n_streams, windows, tab, tab_n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
      }
      printf("table %4u MB, gathers + LDS staging, stores %s: %7.3f ms\n", tab_n >> 17, mode == 3 ? "plain" : mode == 7 ? "none" : mode == 11 ? "nt" : "sc0 sc1 nt", best);
    }
  hipDeviceSynchronize();
  return 0;
}
