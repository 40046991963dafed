// ubench_flat.hip -- what does one dependent table lookup per step cost at 16 waves per CU, as a function of how
// many lanes go to the far (L2-resident) table and how the near lanes are served?
//   mode 0: every lane loads from the global table (global_load)
//   mode 1: one flat_load per step; FAR_PCT % of the lanes address the global table, the rest an LDS table
//   mode 2: ds_read for the near lanes + global_load under an exec mask for the far lanes (two instructions)
//   mode 3: every lane reads the LDS table (ds_read)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_flat tools/ubench_flat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <numeric>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr uint32_t LN = 16384;  // LDS table entries (64 KB)

template <int MODE>
__global__ __launch_bounds__(1024) void chase(const uint32_t* __restrict__ G, uint32_t gmask, int steps, uint32_t far_pct,
                                               uint32_t* out) {
  __shared__ uint32_t L[LN];
  for (uint32_t i = threadIdx.x; i < LN; i += 1024) L[i] = G[i] & (LN - 1);
  __syncthreads();
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = (tid * 2654435761u) & gmask;
  // a lane is "far" for the whole run: far_pct % of the lanes, spread over every wave
  const bool far = ((tid * 40503u) >> 8) % 100u < far_pct;
  for (int s = 0; s < steps; s++) {
    if (MODE == 0) {
      x = G[x & gmask];
    } else if (MODE == 3) {
      x = L[x & (LN - 1)];
    } else if (MODE == 1) {
      const uint32_t* p = far ? G + (x & gmask) : static_cast<const uint32_t*>(L + (x & (LN - 1)));
      x = *p;
    } else {
      uint32_t v = L[x & (LN - 1)];
      if (far) v = G[x & gmask];
      x = v;
    }
  }
  if (x == 0xFFFFFFFFu) out[0] = x;
}

template <int MODE>
static void run(const uint32_t* dG, uint32_t n, uint32_t far_pct, uint32_t* dout) {
  const int steps = 2000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(chase<MODE>, dim3(256), dim3(1024), 0, 0, dG, n - 1, 100, far_pct, dout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(chase<MODE>, dim3(256), dim3(1024), 0, 0, dG, n - 1, steps, far_pct, dout);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("mode %d far %3u%%: %7.1f ns/step (%5.0f cycles at 2.1 GHz)\n", MODE, far_pct, ms * 1e6 / steps, ms * 1e6 / steps * 2.1);
}

int main() {
  uint32_t* dout; CK(hipMalloc(&dout, 64));
  const uint32_t n = 1u << 20;  // 4 MB global table (L2 resident)
  std::vector<uint32_t> T(n);
  std::mt19937 g(7);
  for (auto& v : T) v = g() & (n - 1);
  uint32_t* dG; CK(hipMalloc(&dG, (size_t)n * 4));
  CK(hipMemcpy(dG, T.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  run<3>(dG, n, 0, dout);
  run<0>(dG, n, 100, dout);
  for (uint32_t pct : {100u, 50u, 25u, 10u, 0u}) run<1>(dG, n, pct, dout);
  for (uint32_t pct : {100u, 50u, 25u, 10u, 0u}) run<2>(dG, n, pct, dout);
  return 0;
}
