// devcommon.hpp -- device helpers shared by kernels.hip (two-pass engine) and
// scan_v2.hip (single-traversal engine).
#pragma once
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "image.hpp"

namespace aha {

// ---------------------------------------------------------------- automaton
// One probe of the XOR double array.  B = base of the current state (its
// identity), b != 0.  Returns true when the byte was consumed.
template <bool COMPACT>
struct Probe;

template <>
struct Probe<false> {
  // returns: 0 = miss, 1 = hit, 2 = hit on an end state (key set)
  static __device__ __forceinline__ int go(const DevAut &A, uint32_t &B, uint32_t b, uint32_t &key) {
    const uint2 *slots = reinterpret_cast<const uint2 *>(A.slots);
    uint2 e = slots[B ^ b];
    if ((e.y & 0xFFu) == b) {
      B = e.x & W_BASE_MASK;
      if (e.x & W_END) {
        key = e.y >> 8;
        return 2;
      }
      return 1;
    }
    return 0;
  }
  static __device__ __forceinline__ uint32_t fail(const DevAut &A, uint32_t B) {
    return reinterpret_cast<const uint2 *>(A.slots)[B].x & W_BASE_MASK;
  }
  // child(B, b) or A.root when there is none
  static __device__ __forceinline__ uint32_t child_or_root(const DevAut &A, uint32_t B, uint32_t b) {
    const uint2 e = reinterpret_cast<const uint2 *>(A.slots)[B ^ b];
    return (e.y & 0xFFu) == b ? (e.x & W_BASE_MASK) : A.root;
  }
};

template <>
struct Probe<true> {
  static __device__ __forceinline__ int go(const DevAut &A, uint32_t &B, uint32_t b, uint32_t &key) {
    const uint32_t *slots = reinterpret_cast<const uint32_t *>(A.slots);
    uint32_t e = slots[B ^ b];
    if ((e & 0xFFu) == b) {
      B = (e >> C_BASE_SHIFT) & C_BASE_MASK;
      if (e & C_END) {
        key = (uint32_t)A.end_key[B];
        return 2;
      }
      return 1;
    }
    return 0;
  }
  static __device__ __forceinline__ uint32_t fail(const DevAut &A, uint32_t B) {
    return (reinterpret_cast<const uint32_t *>(A.slots)[B] >> C_BASE_SHIFT) & C_BASE_MASK;
  }
  static __device__ __forceinline__ uint32_t child_or_root(const DevAut &A, uint32_t B, uint32_t b) {
    const uint32_t e = reinterpret_cast<const uint32_t *>(A.slots)[B ^ b];
    return (e & 0xFFu) == b ? ((e >> C_BASE_SHIFT) & C_BASE_MASK) : A.root;
  }
};

// fails[nid] (ac.cr:189) of a non-root state B.  tp points at the current (not yet consumed) input byte; the
// bytes before it spell the state.  Only the root and the states with base >= s2_hi own a fail header; the other
// fail targets are functions of the last one or two bytes (automaton.hpp, Placement::headerless).
template <bool COMPACT>
__device__ __forceinline__ uint32_t fail_of(const DevAut &A, uint32_t B, const uint8_t *tp) {
  if (B >= A.s2_hi) return Probe<COMPACT>::fail(A, B);  // header (every state when the ranges are all 0)
  if (B < A.s1_lo) return A.root;                       // depth 1
  const uint32_t y = tp[-1];
  if (B >= A.s2_lo) {  // depth >= 3: the deepest state of depth <= 2 spelled by the last two bytes
    const uint32_t x = tp[-2];
    const uint32_t s1 = Probe<COMPACT>::child_or_root(A, A.root, x);
    if (s1 != A.root) {
      const uint32_t s2 = Probe<COMPACT>::child_or_root(A, s1, y);
      if (s2 != A.root) return s2;
    }
  }
  return Probe<COMPACT>::child_or_root(A, A.root, y);  // depth-1 state of the last byte, or root
}

// delta(B, b): goto/fail loop of match_ (ac.cr:179-190).  Returns true when
// the new state ends a key (is_end?, cedar.cr:657-660 <=> output.value >= 0).
template <bool COMPACT>
__device__ __forceinline__ bool aut_step(const DevAut &A, uint32_t &B, const uint8_t *tp, uint32_t &key) {
  const uint32_t b = *tp;
  if (b == 0) {  // NUL contract: state := root, nothing reported
    B = A.root;
    return false;
  }
  for (;;) {
    int r = Probe<COMPACT>::go(A, B, b, key);
    if (r) return r == 2;
    if (B == A.root) return false;
    B = fail_of<COMPACT>(A, B, tp);
  }
}

__device__ __forceinline__ bool sep_blocked(const MatchArgs &M, uint32_t c) {
  return (M.sep_block[c >> 5] >> (c & 31)) & 1u;
}

__device__ __forceinline__ bool sep_blocked_bits(const uint32_t *bits, uint32_t c) {
  return (bits[c >> 5] >> (c & 31)) & 1u;
}

// first d in [0, D] with doc_off[d] >= a
__device__ __forceinline__ uint64_t first_boundary(const uint64_t *doc_off, uint64_t D, uint64_t a) {
  uint64_t lo = 0, hi = D;  // doc_off[D] = N >= a
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if (doc_off[mid] < a)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------ block helpers
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// 32-bit scans over the wave with DPP moves instead of six trips through the LDS crossbar (__shfl_up is ds_bpermute):
// row_shr 1, 2, 4, 8 scan each row of 16 lanes; row_bcast15 adds the last lane of rows 0 / 2 to rows 1 / 3, row_bcast31 the
// last lane of row 1 to rows 2 and 3.  A lane without a source (or in a row the mask leaves out) receives `old`, the
// operation's identity.
#define AHA_DPP_SCAN(V, ID, OP)                                                                   \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x111, 0xf, 0xf, false)); \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x112, 0xf, 0xf, false)); \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x114, 0xf, 0xf, false)); \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x118, 0xf, 0xf, false)); \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x142, 0xa, 0xf, false)); \
  V = OP(V, (uint32_t)__builtin_amdgcn_update_dpp((int)(ID), (int)(V), 0x143, 0xc, 0xf, false));
__device__ __forceinline__ uint32_t dpp_op_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_op_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  AHA_DPP_SCAN(v, 0u, dpp_op_add)
  return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {
  AHA_DPP_SCAN(v, 0u, dpp_op_max)
  return v;
}
// lane i <- lane i - 1 (lane 0 <- first), lane 63's value for everyone
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// exclusive scan over the block's kBlock threads; total returned via *total
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *smem /*[kBlock/64]*/, T *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  T inc = wave_incl_scan(v);
  if (lane == 63) smem[w] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / 64; i++) {
    T s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  if (total) *total = tot;
  return base + inc - v;
}


// exclusive scan of in[0..n) into out, the sum into *total: one block of 1024 threads (sm: 16 words of its LDS).  k2_scan_small's
// body, and the tail of kf_walk's last block (scan_filter.hip)
__device__ __forceinline__ void scan_small_block(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out,
                                                 uint64_t *total, uint64_t *sm) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t carry = 0;
  for (uint64_t r0 = 0; r0 < n; r0 += 16384) {
    const uint64_t i0 = r0 + (uint64_t)threadIdx.x * 16;
    uint32_t v[16];
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint4 x = reinterpret_cast<const uint4 *>(in + i0)[q];
        v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = i0 + j < n ? in[i0 + j] : 0u;
    }
    uint64_t mine = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) mine += v[j];
    const uint64_t inc = wave_incl_scan(mine);
    if (lane == 63) sm[w] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t x = sm[j];
      if (j < w) base += x;
      tot += x;
    }
    uint64_t run = carry + base + inc - mine;
    if (i0 + 16 <= n) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint64_t a = run, b = run + v[2 * q];
        run = b + v[2 * q + 1];
        reinterpret_cast<ulonglong2 *>(out + i0)[q] = make_ulonglong2(a, b);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (i0 + j < n) out[i0 + j] = run;
        run += v[j];
      }
    }
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

}  // namespace aha
