// ubench.hip -- microbenchmarks that size the match-kernel design on MI355X:
// dependent random-lookup rates from global memory (by table size, chains per
// lane, waves per CU) and from LDS.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ILP>
__global__ void chase_global(const uint32_t* __restrict__ T, uint32_t mask, int steps, uint32_t* out) {
  uint32_t x[ILP];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = (tid * 2654435761u + i * 40503u) & mask;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = T[x[i]];
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc ^= x[i];
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

template <int ILP>
__global__ void chase_lds(const uint32_t* __restrict__ T, uint32_t n, int steps, uint32_t* out) {
  extern __shared__ uint32_t L[];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) L[i] = T[i];
  __syncthreads();
  uint32_t x[ILP];
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = (tid * 2654435761u + i * 40503u) % n;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = L[x[i]];
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc ^= x[i];
  if (acc == 0xFFFFFFFFu) out[0] = acc;
}

static std::vector<uint32_t> make_cycle(uint32_t n, uint32_t seed) {
  std::vector<uint32_t> perm(n), T(n);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 g(seed);
  std::shuffle(perm.begin(), perm.end(), g);
  for (uint32_t i = 0; i < n; i++) T[perm[i]] = perm[(i + 1) % n];
  return T;
}

template <int ILP>
static void run_global(const uint32_t* dT, uint32_t n, int blocks, int threads, uint32_t* dout) {
  int steps = 2000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(chase_global<ILP>, dim3(blocks), dim3(threads), 0, 0, dT, n - 1, 100, dout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chase_global<ILP>, dim3(blocks), dim3(threads), 0, 0, dT, n - 1, steps, dout);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double loads = (double)blocks * threads * ILP * steps;
  printf("global n=%9u (%7.1f KB) ilp=%d blocks=%5d thr=%4d : %8.2f Gload/s  %7.1f ns/step\n", n, n * 4 / 1024.0, ILP,
         blocks, threads, loads / ms / 1e6, ms * 1e6 / steps);
}

template <int ILP>
static void run_lds(const uint32_t* dT, uint32_t n, int blocks, int threads, uint32_t* dout) {
  int steps = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipFuncSetAttribute((const void*)chase_lds<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4));
  hipLaunchKernelGGL(chase_lds<ILP>, dim3(blocks), dim3(threads), n * 4, 0, dT, n, 100, dout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chase_lds<ILP>, dim3(blocks), dim3(threads), n * 4, 0, dT, n, steps, dout);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double loads = (double)blocks * threads * ILP * steps;
  printf("lds    n=%9u (%7.1f KB) ilp=%d blocks=%5d thr=%4d : %8.2f Gload/s  %7.1f ns/step\n", n, n * 4 / 1024.0, ILP,
         blocks, threads, loads / ms / 1e6, ms * 1e6 / steps);
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 64));
  for (uint32_t n : {1u << 13, 1u << 16, 1u << 18, 1u << 20, 1u << 22, 1u << 24, 1u << 27}) {
    auto T = make_cycle(n, 1234);
    uint32_t* dT; CK(hipMalloc(&dT, (size_t)n * 4));
    CK(hipMemcpy(dT, T.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    // 256 CUs; vary waves per CU: blocks x threads
    run_global<1>(dT, n, 256 * 2, 256, dout);   // 8 waves/CU
    run_global<1>(dT, n, 256 * 8, 256, dout);   // 32 waves/CU
    run_global<2>(dT, n, 256 * 8, 256, dout);
    run_global<4>(dT, n, 256 * 8, 256, dout);
    run_global<8>(dT, n, 256 * 4, 256, dout);
    if (n <= (1u << 15)) {
      run_lds<1>(dT, n, 256 * 4, 256, dout);
      run_lds<1>(dT, n, 256 * 4, 512, dout);
      run_lds<2>(dT, n, 256 * 4, 512, dout);
      run_lds<4>(dT, n, 256 * 4, 512, dout);
    }
    CK(hipFree(dT));
  }
  {
    uint32_t n = 32768;  // 128 KB LDS table, one block per CU
    auto T = make_cycle(n, 99);
    uint32_t* dT; CK(hipMalloc(&dT, (size_t)n * 4));
    CK(hipMemcpy(dT, T.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    run_lds<1>(dT, n, 256, 1024, dout);
    run_lds<2>(dT, n, 256, 1024, dout);
    run_lds<4>(dT, n, 256, 1024, dout);
    run_lds<8>(dT, n, 256, 512, dout);
  }
  return 0;
}
