// ubench_valu.hip -- what does one integer VALU wave-instruction cost on gfx950, per SIMD, as a function of the
// waves resident on the SIMD and of the dependency between consecutive instructions?
//
// One workgroup per CU with 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD (the traversal kernel runs 4).
// Each wave executes REPS x 64 instructions of one kind, either as one dependent chain (every instruction reads
// the previous result) or as 4 independent chains interleaved.  Reported: shader cycles (s_memtime) per
// wave-instruction per SIMD = elapsed cycles / (instructions per wave x waves per SIMD); 2.0 is the SIMD-32 peak
// (64 lanes over 2 cycles), 4.0 is what MI355X_MICROARCH.md quotes for one wave alone.
//
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);         \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int REPS = 2000;

// 16 instructions per macro use; DEP: one chain through %0; IND: four chains %0..%3
#define R4(x) x x x x
#define R16(x) R4(R4(x))

#define OPS_DEP(ins) R16(ins " %0, %0, %4\n")
#define OPS_IND(ins) R4(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4\n")

enum Kind {
  K_XOR, K_AND, K_ADD, K_LSHR, K_CNDMASK_VCC, K_CNDMASK_SGPR, K_CMP_CND, K_BFE, K_LSHL_ADD, K_BITOP3, K_CMP_SDWA,
  K_MUL_LO, K_PERM, K_AND_OR, K_CMP_ONLY, K_MAD_U24,
  K_CMP32_CND32, K_CND32_OTHER_DST, K_CND64_VCC, K_XOR_E64, K_MIX_23, K_MOV, K_SALU_MIX, K_XOR3, K_ADD3, K_CMP32_ONLY,
  K_COUNT
};
static const char *kNames[K_COUNT] = {
    "v_xor_b32", "v_and_b32", "v_add_u32", "v_lshrrev_b32", "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)",
    "v_cmp_eq_u32 + v_cndmask (pair, counted as 2)", "v_bfe_u32", "v_lshl_add_u32", "v_bitop3_b32",
    "v_cmp_eq_u32_sdwa + v_cndmask (2)", "v_mul_lo_u32", "v_perm_b32", "v_and_or_b32", "v_cmp_eq_u32 e64 (sgpr dst)",
    "v_mad_u32_u24",
    "v_cmp_eq_u32_e32 vcc + v_cndmask_b32_e32 vcc (2)", "v_cndmask_b32_e32 vcc, dst != src", "v_cndmask_b32_e64 ..., vcc",
    "v_xor_b32_e64 (VOP3 encoding of a VOP2 op)", "v_xor_b32 / v_bfe_u32 alternating (VOP2, VOP3)", "v_mov_b32",
    "v_xor_b32 + s_and_b64 alternating (VALU counted)", "v_or3_b32", "v_add3_u32", "v_cmp_eq_u32_e32 (vcc dst)"};
// instructions per macro expansion (16 uses of the pattern)
static const int kPerBlock[K_COUNT] = {16, 16, 16, 16, 16, 16, 32, 16, 16, 16, 32, 16, 16, 16, 16, 16,
                                       32, 16, 16, 16, 32, 16, 16, 16, 16, 16};

template <int KIND, bool DEP>
__global__ __launch_bounds__(1024) void valu(unsigned long long *cyc, uint32_t *sink, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = a * 3u;
  const uint32_t k = seed | 1u;
  unsigned long long t0, t1;
  __builtin_amdgcn_s_barrier();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int r = 0; r < REPS; r++) {
#define BODY4(dep, ind)                                                                        \
  if (DEP)                                                                                     \
    asm volatile(R4(dep) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k) : "vcc", "scc", "s20", "s21"); \
  else                                                                                         \
    asm volatile(R4(ind) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k) : "vcc", "scc", "s20", "s21");
    if (KIND == K_XOR) { BODY4(OPS_DEP("v_xor_b32"), OPS_IND("v_xor_b32")) }
    if (KIND == K_AND) { BODY4(OPS_DEP("v_and_b32"), OPS_IND("v_and_b32")) }
    if (KIND == K_ADD) { BODY4(OPS_DEP("v_add_u32"), OPS_IND("v_add_u32")) }
    if (KIND == K_LSHR) { BODY4(R16("v_lshrrev_b32 %0, 1, %0\n"), R4("v_lshrrev_b32 %0, 1, %0\nv_lshrrev_b32 %1, 1, %1\nv_lshrrev_b32 %2, 1, %2\nv_lshrrev_b32 %3, 1, %3\n")) }
    if (KIND == K_CNDMASK_VCC) {
      BODY4(R16("v_cndmask_b32_e32 %0, %0, %4, vcc\n"),
            R4("v_cndmask_b32_e32 %0, %0, %4, vcc\nv_cndmask_b32_e32 %1, %1, %4, vcc\nv_cndmask_b32_e32 %2, %2, %4, vcc\nv_cndmask_b32_e32 %3, %3, %4, vcc\n"))
    }
    if (KIND == K_CNDMASK_SGPR) {
      BODY4(R16("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n"),
            R4("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\nv_cndmask_b32_e64 %1, %1, %4, s[20:21]\nv_cndmask_b32_e64 %2, %2, %4, s[20:21]\nv_cndmask_b32_e64 %3, %3, %4, s[20:21]\n"))
    }
    if (KIND == K_CMP_CND) {
      BODY4(R16("v_cmp_eq_u32_e64 s[20:21], %0, %4\nv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n"),
            R4("v_cmp_eq_u32_e64 s[20:21], %0, %4\nv_cndmask_b32_e64 %1, %1, %4, s[20:21]\nv_cmp_eq_u32_e64 s[20:21], %1, %4\nv_cndmask_b32_e64 %2, %2, %4, s[20:21]\n"
               "v_cmp_eq_u32_e64 s[20:21], %2, %4\nv_cndmask_b32_e64 %3, %3, %4, s[20:21]\nv_cmp_eq_u32_e64 s[20:21], %3, %4\nv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n"))
    }
    if (KIND == K_BFE) { BODY4(R16("v_bfe_u32 %0, %0, 1, 31\n"), R4("v_bfe_u32 %0, %0, 1, 31\nv_bfe_u32 %1, %1, 1, 31\nv_bfe_u32 %2, %2, 1, 31\nv_bfe_u32 %3, %3, 1, 31\n")) }
    if (KIND == K_LSHL_ADD) { BODY4(R16("v_lshl_add_u32 %0, %0, 2, %4\n"), R4("v_lshl_add_u32 %0, %0, 2, %4\nv_lshl_add_u32 %1, %1, 2, %4\nv_lshl_add_u32 %2, %2, 2, %4\nv_lshl_add_u32 %3, %3, 2, %4\n")) }
    if (KIND == K_BITOP3) { BODY4(R16("v_bitop3_b32 %0, %0, %4, %4 bitop3:0x78\n"), R4("v_bitop3_b32 %0, %0, %4, %4 bitop3:0x78\nv_bitop3_b32 %1, %1, %4, %4 bitop3:0x78\nv_bitop3_b32 %2, %2, %4, %4 bitop3:0x78\nv_bitop3_b32 %3, %3, %4, %4 bitop3:0x78\n")) }
    if (KIND == K_CMP_SDWA) {
      BODY4(R16("v_cmp_eq_u32_sdwa s[20:21], %0, %4 src0_sel:BYTE_0 src1_sel:DWORD\nv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n"),
            R4("v_cmp_eq_u32_sdwa s[20:21], %0, %4 src0_sel:BYTE_0 src1_sel:DWORD\nv_cndmask_b32_e64 %1, %1, %4, s[20:21]\nv_cmp_eq_u32_sdwa s[20:21], %1, %4 src0_sel:BYTE_0 src1_sel:DWORD\nv_cndmask_b32_e64 %2, %2, %4, s[20:21]\n"
               "v_cmp_eq_u32_sdwa s[20:21], %2, %4 src0_sel:BYTE_0 src1_sel:DWORD\nv_cndmask_b32_e64 %3, %3, %4, s[20:21]\nv_cmp_eq_u32_sdwa s[20:21], %3, %4 src0_sel:BYTE_0 src1_sel:DWORD\nv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n"))
    }
    if (KIND == K_MUL_LO) { BODY4(R16("v_mul_lo_u32 %0, %0, %4\n"), R4("v_mul_lo_u32 %0, %0, %4\nv_mul_lo_u32 %1, %1, %4\nv_mul_lo_u32 %2, %2, %4\nv_mul_lo_u32 %3, %3, %4\n")) }
    if (KIND == K_PERM) { BODY4(R16("v_perm_b32 %0, %0, %4, %4\n"), R4("v_perm_b32 %0, %0, %4, %4\nv_perm_b32 %1, %1, %4, %4\nv_perm_b32 %2, %2, %4, %4\nv_perm_b32 %3, %3, %4, %4\n")) }
    if (KIND == K_AND_OR) { BODY4(R16("v_and_or_b32 %0, %0, %4, %4\n"), R4("v_and_or_b32 %0, %0, %4, %4\nv_and_or_b32 %1, %1, %4, %4\nv_and_or_b32 %2, %2, %4, %4\nv_and_or_b32 %3, %3, %4, %4\n")) }
    if (KIND == K_CMP_ONLY) { BODY4(R16("v_cmp_eq_u32_e64 s[20:21], %0, %4\n"), R4("v_cmp_eq_u32_e64 s[20:21], %0, %4\nv_cmp_eq_u32_e64 s[20:21], %1, %4\nv_cmp_eq_u32_e64 s[20:21], %2, %4\nv_cmp_eq_u32_e64 s[20:21], %3, %4\n")) }
    if (KIND == K_CMP32_CND32) {
      BODY4(R16("v_cmp_eq_u32_e32 vcc, %0, %4\nv_cndmask_b32_e32 %0, %0, %4, vcc\n"),
            R4("v_cmp_eq_u32_e32 vcc, %0, %4\nv_cndmask_b32_e32 %1, %1, %4, vcc\nv_cmp_eq_u32_e32 vcc, %1, %4\nv_cndmask_b32_e32 %2, %2, %4, vcc\n"
               "v_cmp_eq_u32_e32 vcc, %2, %4\nv_cndmask_b32_e32 %3, %3, %4, vcc\nv_cmp_eq_u32_e32 vcc, %3, %4\nv_cndmask_b32_e32 %0, %0, %4, vcc\n"))
    }
    if (KIND == K_CND32_OTHER_DST) {
      BODY4(R16("v_cndmask_b32_e32 %1, %0, %4, vcc\n"),
            R4("v_cndmask_b32_e32 %1, %0, %4, vcc\nv_cndmask_b32_e32 %2, %0, %4, vcc\nv_cndmask_b32_e32 %3, %0, %4, vcc\nv_cndmask_b32_e32 %1, %0, %4, vcc\n"))
    }
    if (KIND == K_CND64_VCC) {
      BODY4(R16("v_cndmask_b32_e64 %0, %0, %4, vcc\n"),
            R4("v_cndmask_b32_e64 %0, %0, %4, vcc\nv_cndmask_b32_e64 %1, %1, %4, vcc\nv_cndmask_b32_e64 %2, %2, %4, vcc\nv_cndmask_b32_e64 %3, %3, %4, vcc\n"))
    }
    if (KIND == K_XOR_E64) {
      BODY4(R16("v_xor_b32_e64 %0, %0, %4\n"),
            R4("v_xor_b32_e64 %0, %0, %4\nv_xor_b32_e64 %1, %1, %4\nv_xor_b32_e64 %2, %2, %4\nv_xor_b32_e64 %3, %3, %4\n"))
    }
    if (KIND == K_MIX_23) {
      BODY4(R4(R4("v_xor_b32 %0, %0, %4\nv_bfe_u32 %0, %0, 1, 31\n") ) ,
            R4(R4("v_xor_b32 %0, %0, %4\nv_bfe_u32 %1, %1, 1, 31\n")))
    }
    if (KIND == K_MOV) {
      BODY4(R16("v_mov_b32 %0, %4\n"), R4("v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4\n"))
    }
    if (KIND == K_SALU_MIX) {
      BODY4(R16("v_xor_b32 %0, %0, %4\ns_and_b64 s[20:21], s[20:21], s[20:21]\n"),
            R4("v_xor_b32 %0, %0, %4\ns_and_b64 s[20:21], s[20:21], s[20:21]\nv_xor_b32 %1, %1, %4\ns_and_b64 s[20:21], s[20:21], s[20:21]\n"
               "v_xor_b32 %2, %2, %4\ns_and_b64 s[20:21], s[20:21], s[20:21]\nv_xor_b32 %3, %3, %4\ns_and_b64 s[20:21], s[20:21], s[20:21]\n"))
    }
    if (KIND == K_XOR3) {
      BODY4(R16("v_or3_b32 %0, %0, %4, %4\n"), R4("v_or3_b32 %0, %0, %4, %4\nv_or3_b32 %1, %1, %4, %4\nv_or3_b32 %2, %2, %4, %4\nv_or3_b32 %3, %3, %4, %4\n"))
    }
    if (KIND == K_ADD3) {
      BODY4(R16("v_add3_u32 %0, %0, %4, %4\n"), R4("v_add3_u32 %0, %0, %4, %4\nv_add3_u32 %1, %1, %4, %4\nv_add3_u32 %2, %2, %4, %4\nv_add3_u32 %3, %3, %4, %4\n"))
    }
    if (KIND == K_CMP32_ONLY) {
      BODY4(R16("v_cmp_eq_u32_e32 vcc, %0, %4\n"), R4("v_cmp_eq_u32_e32 vcc, %0, %4\nv_cmp_eq_u32_e32 vcc, %1, %4\nv_cmp_eq_u32_e32 vcc, %2, %4\nv_cmp_eq_u32_e32 vcc, %3, %4\n"))
    }
    if (KIND == K_MAD_U24) { BODY4(R16("v_mad_u32_u24 %0, %0, %4, %4\n"), R4("v_mad_u32_u24 %0, %0, %4, %4\nv_mad_u32_u24 %1, %1, %4, %4\nv_mad_u32_u24 %2, %2, %4, %4\nv_mad_u32_u24 %3, %3, %4, %4\n")) }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((a ^ b ^ c ^ d) == 0x12345u) sink[0] = a;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND, bool DEP>
static void run_one(int threads, unsigned long long *dcyc, uint32_t *dsink) {
  const int blocks = 256, waves = blocks * threads / 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((valu<KIND, DEP>), dim3(blocks), dim3(threads), 0, 0, dcyc, dsink, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((valu<KIND, DEP>), dim3(blocks), dim3(threads), 0, 0, dcyc, dsink, 3u);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(waves);
  CK(hipMemcpy(h.data(), dcyc, waves * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double per_wave = sum / waves;
  const double insts = (double)REPS * 4 * kPerBlock[KIND];
  const int wps = threads / 256;
  printf("%-46s %s  %d waves/SIMD: %6.2f cyc per wave-instr per SIMD (%6.2f in the wave's own time), %6.1f T lane-ops/s chip-wide (wall %.3f ms)\n",
         kNames[KIND], DEP ? "dependent  " : "independent", wps, per_wave / insts / wps, per_wave / insts,
         insts * waves * 64 / (ms * 1e-3) / 1e12, ms);
}

template <int KIND>
static void run_kind(unsigned long long *dcyc, uint32_t *dsink) {
  for (int threads : {256, 512, 1024}) {
    run_one<KIND, true>(threads, dcyc, dsink);
    run_one<KIND, false>(threads, dcyc, dsink);
  }
}

int main() {
  unsigned long long *dcyc;
  uint32_t *dsink;
  CK(hipMalloc(&dcyc, 256 * 16 * 8));
  CK(hipMalloc(&dsink, 64));
  run_kind<K_CMP32_CND32>(dcyc, dsink);
  run_kind<K_CND32_OTHER_DST>(dcyc, dsink);
  run_kind<K_CND64_VCC>(dcyc, dsink);
  run_kind<K_CMP32_ONLY>(dcyc, dsink);
  run_kind<K_XOR_E64>(dcyc, dsink);
  run_kind<K_MIX_23>(dcyc, dsink);
  run_kind<K_MOV>(dcyc, dsink);
  run_kind<K_SALU_MIX>(dcyc, dsink);
  run_kind<K_XOR3>(dcyc, dsink);
  run_kind<K_ADD3>(dcyc, dsink);
  run_kind<K_XOR>(dcyc, dsink);
  run_kind<K_AND>(dcyc, dsink);
  run_kind<K_ADD>(dcyc, dsink);
  run_kind<K_LSHR>(dcyc, dsink);
  run_kind<K_CNDMASK_VCC>(dcyc, dsink);
  run_kind<K_CNDMASK_SGPR>(dcyc, dsink);
  run_kind<K_CMP_CND>(dcyc, dsink);
  run_kind<K_CMP_ONLY>(dcyc, dsink);
  run_kind<K_CMP_SDWA>(dcyc, dsink);
  run_kind<K_BFE>(dcyc, dsink);
  run_kind<K_LSHL_ADD>(dcyc, dsink);
  run_kind<K_BITOP3>(dcyc, dsink);
  run_kind<K_AND_OR>(dcyc, dsink);
  run_kind<K_PERM>(dcyc, dsink);
  run_kind<K_MAD_U24>(dcyc, dsink);
  run_kind<K_MUL_LO>(dcyc, dsink);
  return 0;
}
