// Lab: what a write-only stream reaches on this chip (the expansion of cfg 5 writes 11.2 GB per step).  Stores of 16 bytes a
// lane, grid-stride over `bytes`; 12-byte records as three dword stores a lane (the uncoalesced form) for comparison; a copy.
// hipcc -O3 --offload-arch=gfx950 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill16(uint4 *p, uint64_t n16, uint32_t v) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
__global__ __launch_bounds__(256) void fill16_tile(uint4 *p, uint64_t n16, uint32_t v) {  // a block owns 64 KiB runs
  const uint64_t per = 4096;  // uint4 per block-run
  for (uint64_t r = blockIdx.x; r * per < n16; r += gridDim.x)
    for (uint64_t i = r * per + threadIdx.x; i < min(n16, (r + 1) * per); i += 256) p[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
__global__ __launch_bounds__(256) void fill12(uint32_t *p, uint64_t n12, uint32_t v) {  // record i = 12 bytes, three dword stores
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n12; i += (uint64_t)gridDim.x * 256) {
    p[3 * i] = v; p[3 * i + 1] = v + 1; p[3 * i + 2] = (uint32_t)i;
  }
}
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ a, uint4 *__restrict__ b, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read16(const uint4 *__restrict__ a, uint64_t n16, uint32_t *out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *out = acc;
}

int main() {
  const uint64_t bytes = 8ull << 30;
  void *a, *b; uint32_t *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch, double gb) {
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, gb / best * 1e3 / 1e9);
    return 0;
  };
  const double gb = (double)bytes;
  for (uint32_t g : {1024u, 4096u, 16384u, 65536u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "fill 16 B/lane, grid %u", g);
    timeit(nm, [&] { hipLaunchKernelGGL(fill16, dim3(g), dim3(256), 0, 0, (uint4 *)a, bytes / 16, 7u); }, gb);
  }
  timeit("fill 16 B/lane, 64 KiB runs per block, 2048", [&] { hipLaunchKernelGGL(fill16_tile, dim3(2048), dim3(256), 0, 0, (uint4 *)a, bytes / 16, 7u); }, gb);
  timeit("fill 16 B/lane, 64 KiB runs per block, 16384", [&] { hipLaunchKernelGGL(fill16_tile, dim3(16384), dim3(256), 0, 0, (uint4 *)a, bytes / 16, 7u); }, gb);
  timeit("fill 12-byte records, 3 dword stores, 16384", [&] { hipLaunchKernelGGL(fill12, dim3(16384), dim3(256), 0, 0, (uint32_t *)a, bytes / 12, 7u); }, gb);
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(a, 3, bytes, 0); }, gb);
  timeit("read 16 B/lane, grid 16384", [&] { hipLaunchKernelGGL(read16, dim3(16384), dim3(256), 0, 0, (const uint4 *)a, bytes / 16, o); }, gb);
  timeit("copy 16 B/lane, grid 16384 (read + write bytes)", [&] { hipLaunchKernelGGL(copy16, dim3(16384), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); }, 2 * gb);
  CK(hipDeviceSynchronize());
  return 0;
}
