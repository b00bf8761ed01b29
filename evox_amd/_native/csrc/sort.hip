// Stable argsort of up to a few ten-thousand fp32 keys by counting ranks.
//
// The populations sorted every generation hold 10³–10⁴ fitness values.  A radix or merge sort
// of that size is six launch-bound passes (≈35 µs on the device, most of it idle); counting,
// for every key, the keys that precede it is n² trivially parallel compare-adds — 10⁸ at
// n = 10⁴, a few µs across 256 CUs — in ONE kernel, and it yields the order and the rank
// (inverse permutation) together:
//     rank[i] = #{ j : k_j < k_i  or  (k_j = k_i and j < i) },   order[rank[i]] = i.
// Keys are mapped to order-preserving unsigned integers (−0 = +0, NaN last, like the tensor
// library's sort).
#include "hip_common.h"

namespace {

constexpr int kChunk = 8192;    // keys staged in LDS per pass (32 KB)
constexpr int kRows = 64;       // keys ranked per workgroup (one per lane)
constexpr int kWaves = 16;      // waves per workgroup; they split the j range
constexpr int kThreads = kWaves * 64;

__device__ __forceinline__ uint32_t sortable(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFEu;      // NaN: after +inf
  if (u == 0x80000000u) u = 0u;                                  // −0 → +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// keys of sk[a, b) below thr (a, b multiples of 4, wave-uniform)
__device__ __forceinline__ int count_below(const uint32_t* sk, int a, int b, uint32_t thr) {
  int cnt = 0;
  const uint4* p = reinterpret_cast<const uint4*>(sk);
#pragma unroll 4
  for (int g = a >> 2; g < (b >> 2); ++g) {
    const uint4 k = p[g];
    cnt += (int)(k.x < thr) + (int)(k.y < thr) + (int)(k.z < thr) + (int)(k.w < thr);
  }
  return cnt;
}

__global__ __launch_bounds__(kThreads) void rank_sort_kernel(const float* __restrict__ keys, int n,
                                                             int64_t* __restrict__ order,
                                                             int64_t* __restrict__ rank_out) {
  __shared__ __attribute__((aligned(16))) uint32_t sk[kChunk];
  __shared__ int partial[kWaves][kRows];
  const int il = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // wave index, known uniform
  const int i0 = blockIdx.x * kRows;
  const int i = i0 + il;
  const uint32_t ki = i < n ? sortable(keys[i]) : 0xFFFFFFFEu;
  int cnt = 0;
  for (int base = 0; base < n; base += kChunk) {
    const int len = min(kChunk, n - base);
    const int padded = (len + 4 * kWaves - 1) / (4 * kWaves) * (4 * kWaves);   // kWaves slices of whole uint4s
    for (int t = threadIdx.x; t < padded; t += kThreads) sk[t] = t < len ? sortable(keys[base + t]) : 0xFFFFFFFFu;
    __syncthreads();
    const int q = padded / kWaves;
    const int lo = s * q, hi = lo + q;
    // Both i0 and every group start are multiples of 4, so a group of four j lies wholly below
    // the block's rows (ties count: j < i), wholly above them (ties do not), or inside
    // [i0, i0+64), where the tie rule depends on the lane.  ki ≤ 0xFFFFFFFE, so ki+1 cannot wrap,
    // and the 0xFFFFFFFF padding never counts.
    const int m0 = min(max(i0 - base, lo), hi), m1 = min(max(i0 + kRows - base, lo), hi);
    cnt += count_below(sk, lo, m0, ki + 1u);
    {
      // at most 16 groups: unrolled so that the LDS reads go out together
      const uint4* p4 = reinterpret_cast<const uint4*>(sk);
#pragma unroll 16
      for (int g = m0 >> 2; g < (m1 >> 2); ++g) {
        const uint4 k = p4[g];
        const int gj = base + 4 * g;
        cnt += (int)(k.x < ki + (uint32_t)(gj + 0 < i)) + (int)(k.y < ki + (uint32_t)(gj + 1 < i)) +
               (int)(k.z < ki + (uint32_t)(gj + 2 < i)) + (int)(k.w < ki + (uint32_t)(gj + 3 < i));
      }
    }
    cnt += count_below(sk, m1, hi, ki);
    __syncthreads();
  }
  partial[s][il] = cnt;
  __syncthreads();
  if (s == 0 && i < n) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) r += partial[w][il];
    order[r] = i;
    if (rank_out) rank_out[i] = r;
  }
}

}  // namespace

// order[r] = index of the r-th smallest key (stable); rank_out (optional) = inverse permutation
EVOX_API int evox_hip_rank_sort(const float* keys, int n, int64_t* order, int64_t* rank_out, void* stream) {
  if (n <= 0) return 0;
  rank_sort_kernel<<<(n + kRows - 1) / kRows, kThreads, 0, (hipStream_t)stream>>>(keys, n, order, rank_out);
  return evox::last_launch_status();
}
