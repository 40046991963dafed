// filter_lab.hip -- timing variants of the prefix filter's inner loop (scan_filter.hip kf_filter) on synthetic text.
// Not part of the product; results in profiles/r05_filter_lab.txt.   hipcc -O3 --offload-arch=gfx950 -o filter_lab filter_lab.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef LOG2
#define LOG2 12
#endif

__global__ void fill(uint8_t *t, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    t[i] = (x % 7 == 0) ? ' ' : (uint8_t)('a' + x % 26);
  }
}

// V: 0 product form, 1 shift-and bit test, 2 (1) + conflict-free fake index (timing only), 3 (1) + 24-bit multiplies,
//    4 (1) + no LDS read at all (timing only: the VALU floor)
template <int V>
__global__ __launch_bounds__(1024) void kf(const uint32_t *bloom, const uint8_t *__restrict__ text, uint64_t n_bytes,
                                           uint16_t *__restrict__ bitmap, unsigned long long *n_cand) {
  __shared__ uint32_t bl[1 << LOG2];
  for (uint32_t i = threadIdx.x; i < (1u << LOG2); i += 1024) bl[i] = bloom[i];
  __syncthreads();
  const uint64_t n_pieces = n_bytes / 16;
  uint32_t mine = 0;
  for (uint64_t p = (uint64_t)blockIdx.x * 1024 + threadIdx.x; p < n_pieces; p += (uint64_t)gridDim.x * 1024) {
    const uint64_t g = p * 16;
    uint32_t d[5] = {0, 0, 0, 0, 0};
    const uint4 v = *reinterpret_cast<const uint4 *>(text + g);
    d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    if (g + 20 <= n_bytes) d[4] = *reinterpret_cast<const uint32_t *>(text + g + 16);
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t w = (k & 3) ? __builtin_amdgcn_alignbyte(d[(k >> 2) + 1], d[k >> 2], (uint32_t)(k & 3)) : d[k >> 2];
      uint32_t h;
      if (V == 7) {  // no fold: word from the product's top bits, the two bit positions from the ten bits below
        h = w * 0x9E3779B1u;
        const uint32_t word = bl[h >> (32 - LOG2)];
        bits |= ((word >> (h >> (32 - LOG2 - 5))) & (word >> (h >> (32 - LOG2 - 10))) & 1u) << k;
        continue;
      }
      if (V == 8) {  // the product's HIGH word: its low bits are the middle of the 64-bit product -- word address by one AND
        h = __umulhi(w, 0x9E3779B1u);
        const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(bl) + (h & (((1u << LOG2) - 1u) << 2)));
        bits |= ((word >> (h >> 16)) & (word >> (h >> 21)) & 1u) << k;
        continue;
      }
      if (V == 3) {
        h = __umul24(w, 0x9E3779u) + __umul24(w >> 8, 0x85EBCBu);
        h ^= h >> 11;
        h <<= 8;
      } else {
        h = w * 0x9E3779B1u;
        h ^= h >> 15;
      }
      if (V == 0) {
        const uint32_t m = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
        bits |= ((bl[h >> (32 - LOG2)] & m) == m) ? (1u << k) : 0u;
      } else {
        uint32_t word;
        if (V == 2) word = bl[((h >> (32 - LOG2)) & ~63u) | (threadIdx.x & 63)];
        else if (V == 4) word = h * 3u;
        else word = bl[h >> (32 - LOG2)];
        const uint32_t t = (word >> (h & 31u)) & (word >> ((h >> 5) & 31u)) & 1u;
        bits |= t << k;
      }
    }
    bitmap[p] = (uint16_t)bits;
    mine += (uint32_t)__builtin_popcount(bits);
  }
  for (int s = 32; s >= 1; s >>= 1) mine += __shfl_xor(mine, s, 64);
  if ((V == 1 || V == 7 && blockIdx.x == 0xFFFFFF) && (threadIdx.x & 63) == 0 && mine) atomicAdd(n_cand, (unsigned long long)mine);
  if ((V == 7 || V == 8) && (threadIdx.x & 63) == 0 && mine && (blockIdx.x & 7) == 0) atomicAdd(n_cand, (unsigned long long)mine * 8);
}

// V5 with the next iteration's text loaded before the current one is hashed
__global__ __launch_bounds__(1024) void kfp(const uint32_t *bloom, const uint8_t *__restrict__ text, uint64_t n_bytes,
                                            uint16_t *__restrict__ bitmap, unsigned long long *n_cand) {
  __shared__ uint32_t bl[1 << LOG2];
  for (uint32_t i = threadIdx.x; i < (1u << LOG2); i += 1024) bl[i] = bloom[i];
  __syncthreads();
  const uint64_t n_pieces = n_bytes / 16;
  const uint64_t stride = (uint64_t)gridDim.x * 1024;
  uint64_t p = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
  uint4 nv = make_uint4(0, 0, 0, 0);
  uint32_t n4 = 0;
  if (p < n_pieces) {
    nv = *reinterpret_cast<const uint4 *>(text + p * 16);
    if (p * 16 + 20 <= n_bytes) n4 = *reinterpret_cast<const uint32_t *>(text + p * 16 + 16);
  }
  for (; p < n_pieces; p += stride) {
    uint32_t d[5] = {nv.x, nv.y, nv.z, nv.w, n4};
    const uint64_t q = p + stride;
    if (q < n_pieces) {
      nv = *reinterpret_cast<const uint4 *>(text + q * 16);
      n4 = (q * 16 + 20 <= n_bytes) ? *reinterpret_cast<const uint32_t *>(text + q * 16 + 16) : 0u;
    }
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t w = (k & 3) ? __builtin_amdgcn_alignbyte(d[(k >> 2) + 1], d[k >> 2], (uint32_t)(k & 3)) : d[k >> 2];
      uint32_t h = w * 0x9E3779B1u;
      h ^= h >> 15;
      const uint32_t word = bl[h >> (32 - LOG2)];
      bits |= ((word >> (h & 31u)) & (word >> ((h >> 5) & 31u)) & 1u) << k;
    }
    bitmap[p] = (uint16_t)bits;
  }
}

template <int V>
void run(const uint32_t *bloom, const uint8_t *text, uint64_t n, uint16_t *bm, unsigned long long *cnt, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto K = V == 6 ? kfp : kf<V == 6 ? 5 : V>;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(K, dim3(grid), dim3(1024), 0, 0, bloom, text, n, bm, cnt);
  hipMemset(cnt, 0, 8);
  hipEventRecord(a, 0);
  const int R = 10;
  for (int i = 0; i < R; i++) hipLaunchKernelGGL(K, dim3(grid), dim3(1024), 0, 0, bloom, text, n, bm, cnt);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned long long c = 0;
  hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost);
  printf("variant %d grid %d: %.4f ms per launch, %.1f GB/s, candidates/launch %llu\n", V, grid, ms / R, n / (ms / R) / 1e6,
         c / R);
}

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 1ull << 30;
  uint8_t *text;
  uint16_t *bm;
  uint32_t *bloom;
  unsigned long long *cnt;
  hipMalloc(&text, n + 64);
  hipMalloc(&bm, n / 8 + 64);
  hipMalloc(&bloom, 4 << LOG2);
  hipMalloc(&cnt, 8);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, text, n);
  std::vector<uint32_t> hb(1 << LOG2, 0), hb7(1 << LOG2, 0), hb8(1 << LOG2, 0);
  uint64_t x = 12345;
  for (int i = 0; i < 1000; i++) {  // a thousand random lower-case 4-grams
    uint32_t w = 0;
    for (int j = 0; j < 4; j++) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      w |= (uint32_t)('a' + (x >> 33) % 26) << (8 * j);
    }
    uint32_t h = w * 0x9E3779B1u;
    h ^= h >> 15;
    hb[h >> (32 - LOG2)] |= (1u << (h & 31)) | (1u << ((h >> 5) & 31));
    const uint32_t g = w * 0x9E3779B1u;
    const uint32_t mh = (uint32_t)(((uint64_t)w * 0x9E3779B1ull) >> 32);
    hb8[(mh >> 2) & ((1u << LOG2) - 1u)] |= (1u << ((mh >> 16) & 31)) | (1u << ((mh >> 21) & 31));
    hb7[g >> (32 - LOG2)] |= (1u << ((g >> (32 - LOG2 - 5)) & 31)) | (1u << ((g >> (32 - LOG2 - 10)) & 31));
  }
  hipMemcpy(bloom, hb.data(), 4 << LOG2, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  for (int grid : {512}) {
    run<1>(bloom, text, n, bm, cnt, grid);
    run<5>(bloom, text, n, bm, cnt, grid);
    hipMemcpy(bloom, hb7.data(), 4 << LOG2, hipMemcpyHostToDevice);
    run<7>(bloom, text, n, bm, cnt, grid);
    hipMemcpy(bloom, hb8.data(), 4 << LOG2, hipMemcpyHostToDevice);
    run<8>(bloom, text, n, bm, cnt, grid);
    hipMemcpy(bloom, hb.data(), 4 << LOG2, hipMemcpyHostToDevice);
  }
  return 0;
}
