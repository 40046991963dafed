// scan_filter.hip -- the PREFIX-FILTER engine for gfx950: byte-level Aho-Corasick for key sets and text where few positions
// can start a key at all (keyword lists over logs: BASELINE config 2).  Replaces src/aha/ac.cr:176-192 (match_) for calls
// with byte offsets and no separator filter on handles whose keys are at least kfMinD bytes long.
//
// The reference's state after byte j is the LONGEST suffix of the text that is a trie path (ac.cr:176-192), i.e. the goto
// walk of the EARLIEST start that is still alive at j.  So instead of carrying a state through every byte:
//   kf_filter   every byte position asks a blocked Bloom filter in LDS (64 KiB) whether its next D bytes (D = min(4, shortest
//               key)) are the first D bytes of some key: one bit per position.  Stateless, position-parallel, coalesced.
//   kf_walk     a wave takes a chunk of 4 KiB: the candidate bits of the chunk (and of the kfWarm bytes before it: walks that
//               reach into the chunk) become batches of 64 candidates in position order, a lane walks ONE candidate's goto
//               path through the byte-level image (no fail links, no state between candidates); an END step is the
//               reference's event exactly when no earlier start's walk is still alive at its end -- the exclusive prefix
//               maximum of the walks' reaches -- and then, in (start, step) order, the events are already in position order.
//               A start the filter rejects dies within D - 1 bytes: it ends nothing (keys are at least D bytes long) and
//               outlives no later start's END (which lies at least D bytes behind that start).
// The events leave in the byte-level engine's region format ({END state, end offset in the document} at
// evd[chunk * ev_stride + seq], ev_cnt, doc_ev_rank), so count / scan / expansion / document offsets are scan_v2.hip's
// (v2_launch_direct_post).  Candidates are what the text makes of the key set: capi.cpp looks at their number after kf_filter
// and hands dense batches (or key sets whose filter would be full) to the other engines.
#include <hip/hip_runtime.h>

#include "automaton.hpp"
#include "devcommon.hpp"
#include "image.hpp"

namespace aha {

namespace {

constexpr int kfTile = 4096, kfWarm = 64, kfAhead = 64;
constexpr int kfRow = kfWarm + kfTile + kfAhead;
constexpr int kfMaxEnds = 4;  // END steps a walk keeps; a walk with more hands the call to the other engines

__device__ __forceinline__ uint32_t wave_incl_scan_max(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v = max(v, o);
  }
  return v;
}

__device__ __forceinline__ uint32_t kf_hash(uint32_t w) {
  uint32_t h = w * 0x9E3779B1u;
  return h ^ (h >> 15);
}

// ---- the filter: bit p of the bitmap <=> text[p .. p + D) may start a key
__global__ __launch_bounds__(1024) void kf_filter(FilterDev F, const uint8_t *__restrict__ text, uint64_t n_bytes,
                                                   uint16_t *__restrict__ bitmap, unsigned long long *n_cand) {
  __shared__ uint32_t bl[1 << kFilterLog2];
  for (uint32_t i = threadIdx.x; i < (1u << kFilterLog2); i += 1024) bl[i] = F.bloom[i];
  __syncthreads();
  const uint32_t dmask = F.d >= 4 ? 0xFFFFFFFFu : ((1u << (8 * F.d)) - 1u);
  const uint64_t n_pieces = ((n_bytes + 63) / 64) * 4;  // whole 64-bit words of the bitmap (pieces beyond the text: no bit)
  uint32_t mine = 0;
  for (uint64_t p = (uint64_t)blockIdx.x * 1024 + threadIdx.x; p < n_pieces; p += (uint64_t)gridDim.x * 1024) {
    const uint64_t g = p * 16;
    uint32_t d[5] = {0, 0, 0, 0, 0};  // 16 bytes + 3 of look-ahead (bytes beyond the text read as 0: no key holds a NUL)
    if (g >= n_bytes) {
      bitmap[p] = 0;
      continue;
    }
    if (g + 20 <= n_bytes) {
      const uint4 v = *reinterpret_cast<const uint4 *>(text + g);
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
      d[4] = *reinterpret_cast<const uint32_t *>(text + g + 16);
    } else {
      for (int j = 0; j < 20 && g + j < n_bytes; j++) d[j >> 2] |= (uint32_t)text[g + j] << ((j & 3) * 8);
    }
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t w = ((k & 3) ? __builtin_amdgcn_alignbyte(d[(k >> 2) + 1], d[k >> 2], (uint32_t)(k & 3)) : d[k >> 2]) & dmask;
      const uint32_t h = kf_hash(w);
      const uint32_t m = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
      bits |= ((bl[h >> (32 - kFilterLog2)] & m) == m) ? (1u << k) : 0u;
    }
    bitmap[p] = (uint16_t)bits;
    mine += (uint32_t)__builtin_popcount(bits);
  }
  // the batch's candidates (the host decides with it whether the walks are worth it)
  for (int s = 32; s >= 1; s >>= 1) mine += __shfl_xor(mine, s, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_cand, (unsigned long long)mine);
}

// first d with doc_off[d] >= chunk start, for every chunk (kf_walk reads one word per chunk instead of searching)
__global__ __launch_bounds__(256) void kf_chunk_doc(V2Args M, uint32_t *chunk_dn) {
  if (M.cursor[1] >= 16ull) return;
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < M.n_chunks) chunk_dn[c] = (uint32_t)first_boundary(M.doc_off, M.n_docs, c * (uint64_t)kfTile);
}

// ---- the walks
__global__ __launch_bounds__(256) void kf_walk(DevAut A, V2Args M, const unsigned long long *__restrict__ bitmap,
                                                const uint32_t *__restrict__ chunk_dn) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (M.cursor[1] >= 16ull) return;  // the doc offsets are not what the call says (k_check_docs ran in front)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *row = smem + (size_t)wave * (kfRow + 64 * 2 + 16);
  uint16_t *list = reinterpret_cast<uint16_t *>(row + kfRow);
  const uint32_t *slots = reinterpret_cast<const uint32_t *>(A.slots);
  const int64_t N = (int64_t)M.n_bytes;
  const uint64_t D = M.n_docs;
  const int warm = A.max_len > 1 ? (int)min(A.max_len - 1u, (uint32_t)(kfWarm - 1)) : 0;
  const uint64_t wid = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
  for (uint64_t chunk = wid; chunk < M.n_chunks; chunk += nw) {
    const int64_t a = (int64_t)chunk * kfTile;
    const int64_t e = min(a + kfTile, N);
    // ---- the chunk's text: LDS offset o <-> text position a - kfWarm + o (bytes outside the text read as 0)
    {
      const int64_t g0 = a - kfWarm;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int idx = k * 64 + lane;  // 16-byte piece
        if (idx * 16 < kfRow) {
          const int64_t g = g0 + (int64_t)idx * 16;
          uint4 v = make_uint4(0, 0, 0, 0);
          if (g >= 0 && g + 16 <= N) {
            v = *reinterpret_cast<const uint4 *>(M.text + g);
          } else if (g + 16 > 0 && g < N) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; j++)
              if (g + j >= 0 && g + j < N) w[j >> 2] |= (uint32_t)M.text[g + j] << ((j & 3) * 8);
            v = make_uint4(w[0], w[1], w[2], w[3]);
          }
          *reinterpret_cast<uint4 *>(row + idx * 16) = v;
        }
      }
    }
    // ---- documents: dn = first boundary at or behind the chunk start.  Usually none lies near the chunk: one document.
    const uint64_t dn = chunk_dn[chunk];
    const int64_t b_next = (int64_t)M.doc_off[dn];                         // >= a
    const int64_t b_prev = dn > 0 ? (int64_t)M.doc_off[dn - 1] : 0;         // < a (the document that holds a - 1), or 0
    const bool plain = b_next >= e + kfAhead && b_prev <= a - warm;         // no boundary where a walk of this chunk could meet it
    if (!plain) {  // the documents that start inside the chunk: their events-before counts start at 0
      for (uint64_t d = dn + (uint64_t)lane; d <= D && (int64_t)M.doc_off[d] < e; d += 64) M.doc_ev_rank[d] = 0;
    }
    // ---- candidate bits: lane l its own 64 bytes, lane 0 also the 64 bytes before (starts that can reach into the chunk)
    const uint64_t p0 = (uint64_t)a / 64;
    unsigned long long m = (int64_t)((p0 + (uint64_t)lane) * 64) < N ? bitmap[p0 + lane] : 0ull;
    unsigned long long mw = 0;
    if (lane == 0 && p0 > 0 && warm > 0) mw = bitmap[p0 - 1] & (~0ull << (64 - warm));
    uint32_t carry = 0;  // furthest LDS offset (exclusive) an earlier start's walk is alive at
    uint32_t seq = 0;    // events of the chunk so far (wave-uniform)
    uint2 *reg = M.evd + chunk * (uint64_t)M.ev_stride;
    for (;;) {
      // ---- the next 64 candidates, in position order
      const uint32_t cnt = (uint32_t)__popcll(m) + (uint32_t)__popcll(mw);
      uint32_t incl = wave_incl_scan(cnt);
      const uint32_t total = __shfl(incl, 63, 64);
      if (total == 0) break;
      if (total > 384u) {  // about a candidate per ten bytes: not this engine's text -- the other engines take the call
        if (lane == 0) M.cursor[1] = 3ull;
        break;
      }
      uint32_t rank = incl - cnt;
      for (int it = 0; it < 64; it++) {  // (a lane gives at most 64 candidates to a batch)
        const bool go = rank < 64u && (mw | m) != 0ull;
        if (!__builtin_amdgcn_ballot_w64(go)) break;
        if (go) {
          const bool fromw = mw != 0ull;
          const unsigned long long cur = fromw ? mw : m;
          const uint32_t b = (uint32_t)__builtin_ctzll(cur);
          list[rank] = (uint16_t)((fromw ? 0u : (uint32_t)(kfWarm + lane * 64)) + b);
          if (fromw) mw &= mw - 1; else m &= m - 1;
          rank++;
        }
      }
      const bool have = (uint32_t)lane < min(total, 64u);
      const uint32_t q = have ? list[lane] : 0u;
      // ---- the candidate's document: [ds, de) in LDS offsets (a start in a document that ends before the chunk is dead)
      int32_t ds = 0;  // the document's first byte as an LDS offset (negative: it starts before the window)
      uint32_t de = 0;
      bool ok = have;
      if (plain) {
        ds = (int32_t)(b_prev - (a - kfWarm));
        de = (uint32_t)min<int64_t>(min(b_next, N) - (a - kfWarm), kfRow);
      } else if (have) {
        const int64_t qa = a - kfWarm + (int64_t)q;
        uint64_t d = dn > 0 ? dn - 1 : 0;
        if (qa < (int64_t)M.doc_off[d]) {
          ok = false;  // (in the warm-up, in a document before the one that reaches the chunk)
        } else {
          while (d + 1 <= D && (int64_t)M.doc_off[d + 1] <= qa) d++;
          ds = (int32_t)((int64_t)M.doc_off[d] - (a - kfWarm));
          de = (uint32_t)min<int64_t>((d + 1 <= D ? (int64_t)M.doc_off[d + 1] : N) - (a - kfWarm), kfRow);
        }
      }
      // ---- the goto walk from q (byte-level image: slot[base ^ byte] belongs to the state iff its label is the byte)
      uint32_t reach = 0;  // LDS offset (exclusive) of the last byte the walk is alive at; 0: not even one byte
      uint32_t ej[kfMaxEnds], eb[kfMaxEnds];
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) ej[k] = eb[k] = 0;
      uint32_t ne = 0;
      bool over = false;
      uint32_t B = A.root, p = q;
      bool alive = ok;
      for (uint32_t step = 0; step < A.max_len; step++) {
        const bool go = alive & p < de;
        if (!__builtin_amdgcn_ballot_w64(go)) break;
        const uint32_t byte = row[min(p, (uint32_t)(kfRow - 1))];
        const uint32_t en = slots[go ? (B ^ byte) : 0u];
        const bool hit = go & byte != 0u & (en & 0xFFu) == byte;  // (label 0 is a fail header's: NUL follows no goto)
        alive = hit;
        if (hit) {
          B = (en >> C_BASE_SHIFT) & C_BASE_MASK;
          p++;
          reach = p;
          if (en & C_END) {
            if (ne < (uint32_t)kfMaxEnds) {
#pragma unroll
              for (int k = 0; k < kfMaxEnds; k++)
                if ((uint32_t)k == ne) {
                  ej[k] = p;
                  eb[k] = B;
                }
              ne++;
            } else {
              over = true;
            }
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(over)) {  // more END steps on one walk than a lane keeps: the other engines take the call
        if (lane == 0) M.cursor[1] = 3ull;
        break;
      }
      // ---- an END step at offset j is the reference's event when no earlier start is alive there
      uint32_t pm = wave_incl_scan_max(reach);
      const uint32_t last = __shfl(pm, 63, 64);
      uint32_t before = __shfl_up(pm, 1, 64);
      before = lane == 0 ? carry : max(before, carry);
      carry = max(carry, last);
      uint32_t nv = 0;
      bool val[kfMaxEnds];
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++) {
        // this chunk reports the ends in [a, e): offsets (kfWarm, kfWarm + (e - a)]
        val[k] = (uint32_t)k < ne && ej[k] > before && ej[k] > (uint32_t)kfWarm && ej[k] <= (uint32_t)(kfWarm + (e - a));
        nv += val[k] ? 1u : 0u;
      }
      const uint32_t vincl = wave_incl_scan(nv);
      const uint32_t vtot = __shfl(vincl, 63, 64);
      uint32_t at = seq + vincl - nv;
#pragma unroll
      for (int k = 0; k < kfMaxEnds; k++)
        if (val[k]) {
          if (at < M.ev_stride) reg[at] = make_uint2(eb[k], (uint32_t)((int32_t)ej[k] - ds));  // {END state, end offset in the document}
          at++;
        }
      if (!plain && vtot) {  // documents that start inside the chunk: the events that end at or before their first byte
        for (uint64_t d = dn; d <= D && (int64_t)M.doc_off[d] < e; d++) {
          const uint32_t bo = (uint32_t)((int64_t)M.doc_off[d] - (a - kfWarm));
          uint32_t c = 0;
#pragma unroll
          for (int k = 0; k < kfMaxEnds; k++) c += (val[k] && ej[k] <= bo) ? 1u : 0u;
          for (int s = 32; s >= 1; s >>= 1) c += __shfl_xor(c, s, 64);
          if ((uint64_t)lane == ((d - dn) & 63) && c) M.doc_ev_rank[d] += c;  // (the lane that zeroed it)
        }
      }
      seq += vtot;
    }
    if (lane == 0) {
      M.ev_cnt[chunk] = seq;
      if (seq > M.ev_stride) M.cursor[1] = 2ull;  // region full: the host repeats the call with larger regions
    }
  }
}

}  // namespace

size_t filter_walk_lds() { return (size_t)4 * (kfRow + 64 * 2 + 16); }

void filter_launch_filter(const FilterDev &F, const uint8_t *text, uint64_t n_bytes, void *bitmap, unsigned long long *n_cand,
                          uint32_t grid, void *stream) {
  hipLaunchKernelGGL(kf_filter, dim3(grid), dim3(1024), 0, (hipStream_t)stream, F, text, n_bytes, (uint16_t *)bitmap, n_cand);
}

void filter_launch_walk(const DevAut &A, const V2Args &M, const void *bitmap, uint32_t *chunk_dn, uint32_t grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(kf_chunk_doc, dim3((uint32_t)((M.n_chunks + 255) / 256)), dim3(256), 0, s, M, chunk_dn);
  hipLaunchKernelGGL(kf_walk, dim3(grid), dim3(256), filter_walk_lds(), s, A, M, (const unsigned long long *)bitmap, chunk_dn);
}

}  // namespace aha
