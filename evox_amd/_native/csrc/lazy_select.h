// Pieces shared by the round-wise removal loops (IBEA removal, SPEA2 truncation): a score per member that only grows
// while members are removed, the K members of smallest score picked by a search on the order-preserving bits of the
// score, wave-wide minima on DPP moves.
#pragma once
#include "hip_common.h"

namespace evox {
namespace lazy {

constexpr int kSel = 1024;          // threads of the select workgroup
constexpr int kK = 256;             // members a round plays through (the set S)
constexpr int kPerLane = kK / 64;
constexpr unsigned kDeadKey = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// the float behind a key (kDeadKey reads as a NaN)
__device__ __forceinline__ float key_value(unsigned key) {
  return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}

// +inf (removed) and NaN never take part
__device__ __forceinline__ unsigned member_key(float f) { return f < INFINITY ? ordered_bits(f) : kDeadKey; }

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// minimum over the wave, the same value in every lane: four DPP steps inside each row of 16 lanes, then the four rows
__device__ __forceinline__ unsigned wave_min_dpp(unsigned v) {
  v = min(v, (unsigned)dpp_i<0xB1>((int)v));
  v = min(v, (unsigned)dpp_i<0x4E>((int)v));
  v = min(v, (unsigned)dpp_i<0x141>((int)v));
  v = min(v, (unsigned)dpp_i<0x140>((int)v));
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
  v = min(v, dpp_i<0xB1>(v));
  v = min(v, dpp_i<0x4E>(v));
  v = min(v, dpp_i<0x141>(v));
  v = min(v, dpp_i<0x140>(v));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

// number of keys below `pivot` in the workgroup; `buf` alternates between calls so that one barrier per call is enough
template <int E>
__device__ __forceinline__ int block_count_below(const unsigned (&key)[E], unsigned pivot, int* buf) {
  int c = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) c += __popcll(__ballot(key[e] < pivot));      // wave-uniform
  if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = c;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int w = 0; w < kSel / 64; ++w) total += buf[w];
  return total;
}

// τ with count(key < τ) ≤ K, as large as a short search finds: count(lo) ≤ K < count(hi) throughout.  Keys only grow from
// round to round, so the previous round's τ is still a valid lower end, and the distance it moved last time is a good
// first step from there.  `tie`: the search ran out with fewer than K/2 members below τ — more than K/2 share the key τ.
template <int E, int K = kK>
__device__ __forceinline__ unsigned find_tau(const unsigned (&key)[E], unsigned tau_prev, unsigned step, int (*s_cnt)[kSel / 64],
                                             int& flip, int& bisections, int& below, bool& tie) {
  unsigned lo = tau_prev, hi = kDeadKey;
  below = 0;
  if (lo != 0) { below = block_count_below<E>(key, lo, s_cnt[flip]); flip ^= 1; ++bisections; }
  while (step != 0 && below < K / 2) {
    const unsigned cand = lo + step;
    if (cand < lo || cand >= kDeadKey) break;
    const int c = block_count_below<E>(key, cand, s_cnt[flip]); flip ^= 1;
    ++bisections;
    if (c <= K) { lo = cand; below = c; step = step < 0x40000000u ? step * 2 : step; } else { hi = cand; break; }
  }
  while (hi - lo > 1 && below < K / 2) {
    const unsigned mid = lo + (hi - lo) / 2;
    const int c = block_count_below<E>(key, mid, s_cnt[flip]); flip ^= 1;
    if (c <= K) { lo = mid; below = c; } else hi = mid;
    ++bisections;
  }
  tie = below < K / 2;
  return lo;
}

}  // namespace lazy
}  // namespace evox
