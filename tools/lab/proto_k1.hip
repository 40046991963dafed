// proto_k1.hip -- lab prototype (not product code): the fast path of a hash-keyed character walk.
// Every lane walks a 64-byte piece character by character (raw UTF-8 bytes, no symbol decode): the pair (previous character,
// this character) is hashed and looked up in a blocked Bloom filter in LDS over the two-character trie paths; VARIANT 1 adds a
// global probe of a 16-byte table entry for every Bloom positive (the "medium path").  Timing + no false negatives only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define K1_THREADS 1024
constexpr int kPiece = 64;          // bytes per lane
constexpr int kRow = 80;            // LDS row: the piece + 16 bytes of look-ahead (20 dwords: 2-way conflicts at worst)
constexpr int kTile = 64 * kPiece;  // bytes per wave and iteration

template <int VARIANT, int ILP>
__global__ __launch_bounds__(K1_THREADS) void k1_filter(const uint8_t *__restrict__ text, uint64_t n_bytes,
                                                        const uint32_t *__restrict__ bloom, uint32_t bloom_log2_words,
                                                        const uint8_t *__restrict__ lentab_g, uint64_t *__restrict__ out,
                                                        const uint4 *__restrict__ table, uint32_t table_mask,
                                                        uint32_t *__restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t bloom_words = 1u << bloom_log2_words;
  uint32_t *bl = reinterpret_cast<uint32_t *>(smem);
  uint8_t *lentab = smem + (size_t)bloom_words * 4;
  uint8_t *rows = lentab + 256;
  for (uint32_t i = threadIdx.x; i < bloom_words; i += K1_THREADS) bl[i] = bloom[i];
  for (uint32_t i = threadIdx.x; i < 256; i += K1_THREADS) lentab[i] = lentab_g[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *row = rows + ((size_t)wave * 64 + lane) * kRow;
  const uint32_t lb = (uint32_t)(row - smem);
  const uint32_t ltab = (uint32_t)(lentab - smem);
  const uint64_t n_tiles = (n_bytes + kTile - 1) / kTile;
  const uint64_t wave_id = (uint64_t)blockIdx.x * (K1_THREADS / 64) + wave, n_waves = (uint64_t)gridDim.x * (K1_THREADS / 64);
  const uint32_t wshift = 32u - bloom_log2_words;
  uint32_t acc_sink = 0;
  constexpr int kT = kTile * ILP;
  const uint64_t n_tiles2 = (n_bytes + kT - 1) / kT;
  for (uint64_t tile = wave_id; tile < n_tiles2; tile += n_waves) {
    uint32_t o[ILP], po[ILP], gp[ILP], clo[ILP], chi[ILP];
#pragma unroll
    for (int w = 0; w < ILP; w++) {
      const uint64_t g0 = tile * kT + (uint64_t)(w * 64 + lane) * kPiece;
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0, va = v0;
      if (g0 + 80 <= n_bytes) {
        const uint4 *p = reinterpret_cast<const uint4 *>(text + g0);
        v0 = p[0];
        v1 = p[1];
        v2 = p[2];
        v3 = p[3];
        va = p[4];
      }
      uint4 *d = reinterpret_cast<uint4 *>(row + (size_t)w * K1_THREADS * kRow);
      d[0] = v0;
      d[1] = v1;
      d[2] = v2;
      d[3] = v3;
      d[4] = va;
      o[w] = 0;
      po[w] = 64;
      gp[w] = 0;
      clo[w] = chi[w] = 0;
    }
    for (;;) {
      bool more = false;
#pragma unroll
      for (int w = 0; w < ILP; w++) more |= po[w] < 64u || o[w] == 0u;
      if (__builtin_amdgcn_ballot_w64(more) == 0ull) break;
      uint32_t x[ILP], s[ILP], c[ILP], h[ILP], bw[ILP];
#pragma unroll
      for (int w = 0; w < ILP; w++) x[w] = *reinterpret_cast<const uint32_t *>(smem + lb + w * K1_THREADS * kRow + (min(o[w], 72u) & ~3u));
#pragma unroll
      for (int w = 0; w < ILP; w++) {
        if (VARIANT & 8) {
        } else {
          const uint32_t *q = reinterpret_cast<const uint32_t *>(smem + lb + w * K1_THREADS * kRow + (min(o[w], 72u) & ~3u));
          x[w] = __builtin_amdgcn_alignbyte(q[1], x[w], o[w] & 3u);
        }
      }
#pragma unroll
      for (int w = 0; w < ILP; w++) {
        if (VARIANT & 2) {
          const uint32_t b0 = x[w] & 0xFFu;
          s[w] = 8u + (b0 >= 0xC0u ? 8u : 0u) + (b0 >= 0xE0u ? 8u : 0u);
        } else {
          s[w] = lentab[x[w] & 0xFFu];
        }
      }
#pragma unroll
      for (int w = 0; w < ILP; w++) {
        c[w] = __builtin_amdgcn_ubfe(x[w], 0u, s[w]);
        h[w] = __umul24(c[w], 0x9E3779u) + gp[w];
        const uint32_t g = __umul24(c[w], 0x85EBCBu);
        gp[w] = __builtin_amdgcn_alignbit(g, g, 11);
        h[w] ^= h[w] >> 16;
        bw[w] = (VARIANT & 4) ? h[w] : bl[h[w] >> wshift];
      }
#pragma unroll
      for (int w = 0; w < ILP; w++) {
        const uint32_t m = (1u << (h[w] & 31u)) | (1u << ((h[w] >> 5) & 31u));
        bool pass = po[w] < 64u & (bw[w] & m) == m;
        if (VARIANT & 1) {
          if (pass) {
            const uint4 e = table[(h[w] * 0x2545F491u >> 8) & table_mask];
            pass = e.x != c[w];  // (never equal: keeps the load alive)
            acc_sink += e.y;
          }
        }
        const unsigned long long bit = (unsigned long long)(pass ? 1u : 0u) << (po[w] & 63u);
        clo[w] |= (uint32_t)bit;
        chi[w] |= (uint32_t)(bit >> 32);
        po[w] = o[w];
        o[w] += s[w] >> 3;
      }
    }
#pragma unroll
    for (int w = 0; w < ILP; w++) {
      const uint64_t g0 = tile * kT + (uint64_t)(w * 64 + lane) * kPiece;
      if (g0 < n_bytes) out[g0 / 64] = (unsigned long long)chi[w] << 32 | clo[w];
    }
  }
  if ((VARIANT & 1) && acc_sink == 0x12345u) sink[0] = acc_sink;
}

extern "C" int proto_k1_run(const uint8_t *text, uint64_t n_bytes, const uint32_t *bloom, uint32_t bloom_log2_words,
                            const uint8_t *lentab, uint64_t *out, const void *table, uint32_t table_mask, uint32_t *sink,
                            int variant, int reps, float *ms_out) {
  const int ilp = variant >> 8 ? variant >> 8 : 1, var = variant & 255;
  const size_t lds = ((size_t)4 << bloom_log2_words) + 256 + (size_t)K1_THREADS * kRow * ilp;
  void (*kern)(const uint8_t *, uint64_t, const uint32_t *, uint32_t, const uint8_t *, uint64_t *, const uint4 *, uint32_t, uint32_t *) = nullptr;
#define PICK(V, I) if (var == V && ilp == I) kern = k1_filter<V, I>;
  PICK(0, 1) PICK(1, 1) PICK(2, 1) PICK(4, 1) PICK(6, 1) PICK(0, 2) PICK(1, 2) PICK(2, 2) PICK(6, 2) PICK(8,1)
  if (!kern) return -3;
  if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(K1_THREADS), lds, 0, text, n_bytes, bloom, bloom_log2_words, lentab, out,
                       (const uint4 *)table, table_mask, sink);
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  *ms_out = best;
  return (int)hipGetLastError();
}
