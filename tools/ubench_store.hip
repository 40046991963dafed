// ubench_store.hip -- does a pending global store delay the next dependent load?  (gfx9 has ONE vmcnt for loads
// and stores: s_waitcnt vmcnt(0) behind a load also waits for every older store.)  Dependent random loads from an
// L2-resident table, 16 waves per CU like k2_traverse, with an 8-byte store to a per-lane region every step / every
// 4th step / never.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_store tools/ubench_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int EVERY>  // store every EVERY-th step; 0 = never
__global__ __launch_bounds__(1024) void chase(const uint32_t* __restrict__ T, uint32_t mask, int steps, uint2* out,
                                               uint32_t stride) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = (tid * 2654435761u) & mask;
  uint2* reg = out + (size_t)tid * stride;
  uint32_t seq = 0;
  for (int s = 0; s < steps; s++) {
    x = T[x];
    if (EVERY && (s % EVERY) == 0) {
      reg[seq & (stride - 1)] = make_uint2(x, (uint32_t)s);
      seq++;
    }
  }
  if (x == 0xFFFFFFFFu) out[0] = make_uint2(x, seq);
}

static std::vector<uint32_t> make_cycle(uint32_t n, uint32_t seed) {
  std::vector<uint32_t> perm(n), T(n);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 g(seed);
  std::shuffle(perm.begin(), perm.end(), g);
  for (uint32_t i = 0; i < n; i++) T[perm[i]] = perm[(i + 1) % n];
  return T;
}

template <int EVERY>
static void run(const uint32_t* dT, uint32_t n, uint2* dout, uint32_t stride) {
  const int steps = 2000, blocks = 256, threads = 1024;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(chase<EVERY>, dim3(blocks), dim3(threads), 0, 0, dT, n - 1, 100, dout, stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chase<EVERY>, dim3(blocks), dim3(threads), 0, 0, dT, n - 1, steps, dout, stride);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("table %7.1f KB, store every %d steps: %7.1f ns/step (%.0f cycles at 2.1 GHz)\n", n * 4 / 1024.0, EVERY,
         ms * 1e6 / steps, ms * 1e6 / steps * 2.1);
}

int main() {
  const uint32_t stride = 512;  // 4 KB region per lane, like evd
  uint2* dout; CK(hipMalloc(&dout, (size_t)256 * 1024 * stride * sizeof(uint2)));
  for (uint32_t n : {1u << 18, 1u << 21}) {
    auto T = make_cycle(n, 1234);
    uint32_t* dT; CK(hipMalloc(&dT, (size_t)n * 4));
    CK(hipMemcpy(dT, T.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    run<0>(dT, n, dout, stride);
    run<1>(dT, n, dout, stride);
    run<4>(dT, n, dout, stride);
    run<32>(dT, n, dout, stride);
    CK(hipFree(dT));
  }
  return 0;
}
