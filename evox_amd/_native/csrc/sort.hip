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

// The same count spread over more workgroups for key vectors that would otherwise leave most of the chip idle
// (10^4 keys are 157 blocks of 64 rows): the j range is cut into kSlices slices, block (rows, slice) writes the
// partial count of its rows, and the second kernel adds the slices of a row (integers: any order gives the same sum)
// and scatters the order.
constexpr int kSlices = 4;      // upper bound; a launch uses as many as keep rows x slices within one round of blocks

// Rows [row_lo, row_lo + rows_n) are ranked against all n keys (row_lo a multiple of 4; the whole vector: 0, n).
__global__ __launch_bounds__(kThreads) void rank_count_kernel(const float* __restrict__ keys, int n, int* __restrict__ partial,
                                                              int slices, int row_lo, int rows_n) {
  __shared__ __attribute__((aligned(16))) uint32_t sk[kChunk];
  __shared__ int part[kWaves][kRows];
  const int il = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = row_lo + blockIdx.x * kRows;
  const int i = i0 + il;
  const bool valid = i < row_lo + rows_n;
  const uint32_t ki = valid ? sortable(keys[i]) : 0xFFFFFFFEu;
  // this block's slice of the keys, in whole groups of four
  const int per = ((n + slices - 1) / slices + 3) / 4 * 4;
  const int j0 = blockIdx.y * per, j1 = min(n, j0 + per);
  int cnt = 0;
  for (int base = j0; base < j1; base += kChunk) {
    const int len = min(kChunk, j1 - base);
    const int padded = (len + 4 * kWaves - 1) / (4 * kWaves) * (4 * kWaves);
    for (int t = threadIdx.x; t < padded; t += kThreads) sk[t] = t < len ? sortable(keys[base + t]) : 0xFFFFFFFFu;
    __syncthreads();
    const int q = padded / kWaves;
    const int lo = s * q, hi = lo + q;
    const int m0 = min(max(i0 - base, lo), hi), m1 = min(max(i0 + kRows - base, lo), hi);
    cnt += count_below(sk, lo, m0, ki + 1u);
    {
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
  part[s][il] = cnt;
  __syncthreads();
  if (s == 0 && valid) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) r += part[w][il];
    partial[(size_t)blockIdx.y * rows_n + (i - row_lo)] = r;
  }
}

__global__ __launch_bounds__(256) void rank_scatter_kernel(const int* __restrict__ partial, int n, int64_t* __restrict__ order,
                                                           int64_t* __restrict__ rank_out, int slices,
                                                           const float* __restrict__ keys, float* __restrict__ sorted_out,
                                                           float* __restrict__ keys_copy) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = 0;
  for (int sl = 0; sl < slices; ++sl) r += partial[(size_t)sl * n + i];
  if (order) order[r] = i;
  if (rank_out) rank_out[i] = r;
  if (sorted_out) sorted_out[r] = keys[i];
  if (keys_copy) keys_copy[i] = keys[i];
}

}  // namespace

// order[r] = index of the r-th smallest key (stable); rank_out (optional) = inverse permutation
EVOX_API int evox_hip_rank_sort(const float* keys, int n, int64_t* order, int64_t* rank_out, void* stream) {
  if (n <= 0) return 0;
  rank_sort_kernel<<<(n + kRows - 1) / kRows, kThreads, 0, (hipStream_t)stream>>>(keys, n, order, rank_out);
  return evox::last_launch_status();
}

// The sliced form: `scratch` holds evox_hip_rank_sort_scratch(n) ints.
EVOX_API int64_t evox_hip_rank_sort_scratch(int n) { return (int64_t)kSlices * (n > 0 ? n : 0); }

EVOX_API int evox_hip_rank_sort_sliced(const float* keys, int n, int64_t* order, int64_t* rank_out, int* scratch, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // two 1024-thread blocks fit a CU: stay within one round of 512 blocks
  const int rows = (n + kRows - 1) / kRows;
  int slices = 2 * evox::kNumCU / rows;
  slices = slices < 1 ? 1 : (slices > kSlices ? kSlices : slices);
  rank_count_kernel<<<dim3(rows, slices), kThreads, 0, st>>>(keys, n, scratch, slices, 0, n);
  rank_scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(scratch, n, order, rank_out, slices, nullptr, nullptr, nullptr);
  return evox::last_launch_status();
}

// The sorted keys themselves (sorted_out[rank[i]] = keys[i]) next to the rank of every key; `order` and `keys_copy`
// (an unsorted copy of the keys, e.g. the other half of a message) may be null.
EVOX_API int evox_hip_rank_sort_values(const float* keys, int n, int64_t* order, int64_t* rank_out, float* sorted_out,
                                       float* keys_copy, int* scratch, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int rows = (n + kRows - 1) / kRows;
  int slices = 2 * evox::kNumCU / rows;
  slices = slices < 1 ? 1 : (slices > kSlices ? kSlices : slices);
  rank_count_kernel<<<dim3(rows, slices), kThreads, 0, st>>>(keys, n, scratch, slices, 0, n);
  rank_scatter_kernel<<<(n + 255) / 256, 256, 0, st>>>(scratch, n, order, rank_out, slices, keys, sorted_out, keys_copy);
  return evox::last_launch_status();
}

// One-launch counting rank with the compared keys held in SCALAR registers.
//
// The kernels above stage the keys in LDS and read them back as broadcasts; every wave-instruction of such a read
// occupies the LDS pipeline as long as a full-width one, which makes LDS a co-bottleneck of the count.  Here a
// wavefront keeps its 64 row keys in a vector register (one per lane) and walks over its share of the n keys with a
// wave-uniform index: the loads are scalar loads (eight or sixteen consecutive keys per instruction, served by the
// scalar data cache), and a compare takes its second operand straight from a scalar register — two vector
// instructions per 64 compares and nothing else.  The compares are done on the floats themselves: -0 = +0 as in the
// integer image used above; a NaN row key (all of its compares are false) gets a second pass that counts every
// ordered key plus the NaN keys in front of it, taken only by wavefronts that hold such a row.
// Sixteen wavefronts split the key range, the 64 rows of a workgroup are finished in LDS and scattered directly.
namespace {

constexpr int kSB = 16;          // keys per scalar load

template <bool TIE_LE>
__device__ __forceinline__ int count_block(const float (&kj)[kSB], float ki) {
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < kSB; ++u) cnt += TIE_LE ? (int)(kj[u] <= ki) : (int)(kj[u] < ki);
  return cnt;
}

__device__ __forceinline__ void load_block(float (&kj)[kSB], const float* __restrict__ p) {
#pragma unroll
  for (int u = 0; u < kSB; ++u) kj[u] = p[u];
}

// keys[a, b) that precede ki (a, b wave-uniform).  Blocks of 16 keys (one scalar load each) are double-buffered: the
// load of the next block is in flight during the compares of the current one, so the scalar-cache latency is covered by
// a wavefront's own work and not only by its neighbours on the SIMD.  (The compiler hoists one further load; with 16-key
// blocks three live blocks still fit the scalar register file.)
template <bool TIE_LE>
__device__ __forceinline__ int count_uniform(const float* __restrict__ keys, int a, int b, float ki) {
  int cnt = 0;
  int j = a;
  if (j + kSB <= b) {
    float bufA[kSB], bufB[kSB];
    load_block(bufA, keys + j);
    j += kSB;
    while (j + 2 * kSB <= b) {
      load_block(bufB, keys + j);
      cnt += count_block<TIE_LE>(bufA, ki);
      load_block(bufA, keys + j + kSB);
      cnt += count_block<TIE_LE>(bufB, ki);
      j += 2 * kSB;
    }
    if (j + kSB <= b) {
      load_block(bufB, keys + j);
      cnt += count_block<TIE_LE>(bufA, ki);
      cnt += count_block<TIE_LE>(bufB, ki);
      j += kSB;
    } else {
      cnt += count_block<TIE_LE>(bufA, ki);
    }
  }
  for (; j < b; ++j) {
    const float kj = keys[j];
    cnt += TIE_LE ? (int)(kj <= ki) : (int)(kj < ki);
  }
  return cnt;
}

__global__ __launch_bounds__(kThreads) void rank_sort_scalar_kernel(const float* __restrict__ keys, int n,
                                                                    int64_t* __restrict__ order,
                                                                    int64_t* __restrict__ rank_out) {
  __shared__ int partial[kWaves][kRows];
  const int il = threadIdx.x & 63;
  const int s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i0 = blockIdx.x * kRows;
  const int i = i0 + il;
  const float ki = keys[min(i, n - 1)];
  const int per = ((n + kWaves - 1) / kWaves + 15) & ~15;
  const int ja = min(n, s * per), jb = min(n, ja + per);
  // keys in front of the workgroup's rows win ties, keys behind them do not, keys among them by their index
  const int m0 = min(max(i0, ja), jb), m1 = min(max(i0 + kRows, ja), jb);
  int cnt = count_uniform<true>(keys, ja, m0, ki);
  if (m0 < m1) {
    // the workgroup's own rows (at most 64 keys, in one wavefront's share): one vector load, then lane by lane through
    // a scalar register — a scalar load per key would expose its latency 64 times on the workgroup's critical path
    const int kv = __float_as_int(keys[min(m0 + il, n - 1)]);
#pragma unroll 8
    for (int u = 0; u < m1 - m0; ++u) {
      const float kj = __int_as_float(__builtin_amdgcn_readlane(kv, u));
      cnt += (int)(kj < ki) + (int)(kj == ki && m0 + u < i);
    }
  }
  cnt += count_uniform<false>(keys, m1, jb, ki);
  const bool nan_i = ki != ki;
  if (__ballot(nan_i) != 0ull) {
    int c2 = 0;
    for (int j = ja; j < jb; ++j) {
      const float kj = keys[j];
      c2 += (kj == kj) ? 1 : (int)(j < i);
    }
    if (nan_i) cnt = c2;
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

// order[r] = index of the r-th smallest key (stable, NaN last, -0 = +0); rank_out (optional) = inverse permutation.
EVOX_API int evox_hip_rank_sort_scalar(const float* keys, int n, int64_t* order, int64_t* rank_out, void* stream) {
  if (n <= 0) return 0;
  rank_sort_scalar_kernel<<<(n + kRows - 1) / kRows, kThreads, 0, (hipStream_t)stream>>>(keys, n, order, rank_out);
  return evox::last_launch_status();
}

// One-launch rank by splitters: n log n instead of n^2 compares.
//
// Counting every key against every other is 10^8 compare-adds at n = 10^4: 14-17 us however the keys are fed to the
// compares (LDS broadcasts or scalar registers above).  Here every workgroup first partitions ALL keys — redundantly, in
// LDS, it costs a few thousand cycles — into 257 buckets bounded by 256 splitters (a sorted regular sample), with the
// keys of every bucket listed together; the rank of one of its own 64 rows is then the number of keys in the buckets in
// front plus a count over the members of the row's own bucket only (about n / 257 of them).  Equal keys always share a
// bucket, so ties go by index exactly as in the stable sort; the result does not depend on the order in which the
// members of a bucket were listed.  A key vector with few distinct values degenerates to the plain count.
// Measured at n = 10^4: 21 us — the partition's dependent LDS round trips (eight per binary search) cost more than the
// compares they save, so the scalar-register count above (17 us) stays the default; this form is kept selectable.
namespace {

constexpr int kBkSplit = 256;

__device__ __forceinline__ int bucket_of(const uint32_t* spl, uint32_t k) {
  int c = 0;
#pragma unroll
  for (int step = kBkSplit / 2; step >= 1; step >>= 1)
    if (spl[c + step - 1] < k) c += step;
  if (c == kBkSplit - 1 && spl[kBkSplit - 1] < k) c = kBkSplit;
  return c;                                      // number of splitters below k
}

__global__ __launch_bounds__(kThreads) void rank_sort_bucket_kernel(const float* __restrict__ keys, int n, int npad,
                                                                    int64_t* __restrict__ order,
                                                                    int64_t* __restrict__ rank_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bk_smem[];
  uint32_t* sk = reinterpret_cast<uint32_t*>(bk_smem);            // npad sortable keys
  uint16_t* members = reinterpret_cast<uint16_t*>(sk + npad);     // npad key indices, bucket by bucket
  uint32_t* spl = reinterpret_cast<uint32_t*>(members + npad);    // kBkSplit sorted splitters
  uint32_t* samp = spl + kBkSplit;                                // the sample before it is sorted
  int* hist = reinterpret_cast<int*>(samp + kBkSplit);            // kBkSplit + 1 bucket sizes
  int* start = hist + kBkSplit + 4;                               // kBkSplit + 2 bucket starts
  int* cursor = start + kBkSplit + 4;
  int* partial = cursor + kBkSplit + 4;                           // kWaves x kRows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int t = tid; t < n; t += kThreads) sk[t] = sortable(keys[t]);
  if (tid < kBkSplit + 1) { hist[tid] = 0; cursor[tid] = 0; }
  if (tid < kBkSplit) spl[tid] = 0xFFFFFFFFu;
  __syncthreads();
  const int S = min(kBkSplit, n);
  if (tid < S) samp[tid] = sk[(int)((long long)tid * n / S)];
  __syncthreads();
  {
    // rank of every sample among the samples: four threads per sample, 64 compares each
    const int t = tid >> 2, q = tid & 3;
    const uint32_t kt = t < S ? samp[t] : 0u;
    int c = 0;
    for (int u = q * (kBkSplit / 4); u < min(S, (q + 1) * (kBkSplit / 4)); ++u) {
      const uint32_t ku = samp[u];
      c += (int)(ku < kt) + (int)(ku == kt && u < t);
    }
    c += __shfl_xor(c, 1, 64);
    c += __shfl_xor(c, 2, 64);
    if (q == 0 && t < S) spl[c] = kt;
  }
  __syncthreads();
  uint16_t bkt[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int t = tid + m * kThreads;
    if (t < n) {
      const int b = bucket_of(spl, sk[t]);
      bkt[m] = (uint16_t)b;
      atomicAdd(&hist[b], 1);
    }
  }
  __syncthreads();
  if (wave == 0) {
    // exclusive scan of the 257 bucket sizes: five consecutive entries per lane
    int v[5], sum = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int b = lane * 5 + k;
      v[k] = b <= kBkSplit ? hist[b] : 0;
      sum += v[k];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    int run = incl - sum;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int b = lane * 5 + k;
      if (b <= kBkSplit + 1) start[b] = run;
      run += v[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int t = tid + m * kThreads;
    if (t < n) {
      const int b = bkt[m];
      members[start[b] + atomicAdd(&cursor[b], 1)] = (uint16_t)t;
    }
  }
  __syncthreads();
  // the workgroup's own rows: lane = row, the sixteen wavefronts share the members of a row's bucket
  const int i = blockIdx.x * kRows + lane;
  const uint32_t ki = i < n ? sk[i] : 0xFFFFFFFEu;
  const int bi = i < n ? bucket_of(spl, ki) : 0;
  const int lo = start[bi], hi = i < n ? start[bi + 1] : lo;
  int cnt = 0;
  for (int m = lo + wave; m < hi; m += kWaves) {
    const int j = members[m];
    const uint32_t kj = sk[j];
    cnt += (int)(kj < ki) + (int)(kj == ki && j < i);
  }
  partial[wave * kRows + lane] = cnt;
  __syncthreads();
  if (wave == 0 && i < n) {
    int r = lo;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) r += partial[w * kRows + lane];
    order[r] = i;
    if (rank_out) rank_out[i] = r;
  }
}

inline int bucket_lds_bytes(int npad) { return npad * 6 + (2 * kBkSplit + 3 * (kBkSplit + 4) + kWaves * kRows) * 4; }

}  // namespace

// order[r] = index of the r-th smallest key (stable, NaN last, -0 = +0); rank_out (optional) = inverse permutation.
// n <= 16384.
EVOX_API int evox_hip_rank_sort_bucket(const float* keys, int n, int64_t* order, int64_t* rank_out, void* stream) {
  if (n <= 0) return 0;
  if (n > 16 * kThreads) return (int)hipErrorInvalidValue;
  const int npad = (n + 7) & ~7;
  const int lds = bucket_lds_bytes(npad);
  static bool configured[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_sort_bucket_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bucket_lds_bytes(16 * kThreads));
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  rank_sort_bucket_kernel<<<(n + kRows - 1) / kRows, kThreads, lds, (hipStream_t)stream>>>(keys, n, npad, order, rank_out);
  return evox::last_launch_status();
}

// Partial rank counts of the rows [row_lo, row_lo + rows_n) against all n keys: scratch[s * rows_n + i] for s below the
// returned number of slices (at most evox_hip_rank_window_slices()); the rank of row i is their sum.  What a process
// that owns only those rows of a larger population needs: rows_n x n compares instead of a sort of all n keys.
// Returns the number of slices, or a negative error.
constexpr int kWindowSlices = 8;
EVOX_API int evox_hip_rank_window_slices() { return kWindowSlices; }

EVOX_API int evox_hip_rank_window_count(const float* keys, int n, int row_lo, int rows_n, int* scratch, void* stream) {
  if (n <= 0 || rows_n <= 0 || row_lo < 0 || row_lo + rows_n > n || (row_lo & 3)) return -(int)hipErrorInvalidValue;
  const int rows = (rows_n + kRows - 1) / kRows;
  int slices = 4 * evox::kNumCU / rows;                       // about two rounds of blocks at most ...
  slices = slices > n / 2048 ? n / 2048 : slices;            // ... of at least 2048 keys each
  slices = slices < 1 ? 1 : (slices > kWindowSlices ? kWindowSlices : slices);
  rank_count_kernel<<<dim3(rows, slices), kThreads, 0, (hipStream_t)stream>>>(keys, n, scratch, slices, row_lo, rows_n);
  const int st = evox::last_launch_status();
  return st ? -st : slices;
}

// Selection of the k smallest (or largest) of n fp32 keys without sorting them: one persistent workgroup
// runs a three-pass radix select (11 + 11 + 10 bits of the order-preserving integer image of the keys,
// histograms in LDS) to find the key of the k-th element, then a fourth pass writes the indices of the
// selected elements in ascending index order (all keys beyond the threshold, plus the first ties at it).
// The keys are read four times from L2; for populations up to a few hundred thousand that is far cheaper
// than the merge sort behind a library top-k (p-best selection of adaptive DE: 50 000 keys, 55 → ≈20 µs).
namespace {

constexpr int kSelThreads = 1024;
constexpr int kSelBins = 2048;

__device__ __forceinline__ uint32_t select_key(float v, bool largest) {
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // ascending floats -> ascending integers
  if (largest) u = ~u;
  return (v != v) ? 0xffffffffu : u;                    // NaN is never preferred
}

constexpr int kSelBatch = 16;          // independent loads in flight per thread: a lone workgroup must hide the
                                       // memory latency itself, or every pass costs one latency per element

// Histogram update that survives clustered keys (fitness values of a converged population share their leading
// bits, so in the first pass a plain atomic per lane would serialise 64-fold on one LDS address): the lanes that
// share the first active lane's bin are added with one aggregated atomic, the others one by one.  The whole
// kernel is bound by the instruction issue of its single compute unit, so the later passes — whose digits are
// mantissa bits and spread over the bins — use the plain form.
template <bool AGGREGATE>
__device__ __forceinline__ void hist_add(unsigned int* hist, uint32_t bin, bool active) {
  if constexpr (AGGREGATE) {
    const unsigned long long todo = __ballot(active);
    if (todo == 0) return;
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lb = __shfl(bin, leader, 64);
    const bool same = active && bin == lb;
    const unsigned long long peers = __ballot(same);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (unsigned int)__popcll(peers));
    active = active && !same;
  }
  if (active) atomicAdd(&hist[bin], 1u);
}

__global__ __launch_bounds__(kSelThreads) void select_kernel(const float* __restrict__ x, int n, int k, int largest,
                                                             int64_t* __restrict__ out) {
  __shared__ unsigned int hist[kSelBins];
  __shared__ unsigned int wave_tot[kSelThreads / 64];
  __shared__ unsigned int s_bin, s_before;
  extern __shared__ unsigned int table[];               // [chunks][16 waves][2]: per-wave counts, then offsets
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lg = largest != 0;
  // f(key, i, valid) for every position of every 1024-key chunk, chunk by chunk; all threads call f the same
  // number of times (f may use wave-wide operations)
  auto for_each = [&](auto&& f) {
    for (int base = 0; base < n; base += kSelBatch * kSelThreads) {
      uint32_t kk[kSelBatch];
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) {
        const int i = base + u * kSelThreads + tid;
        kk[u] = i < n ? select_key(x[i], lg) : 0xffffffffu;
      }
#pragma unroll
      for (int u = 0; u < kSelBatch; ++u) {
        if (base + u * kSelThreads < n) {                 // uniform
          const int i = base + u * kSelThreads + tid;
          f(kk[u], i, i < n);
        }
      }
    }
  };
  uint32_t prefix = 0, mask = 0;
  unsigned int remaining = (unsigned int)k;             // rank (1-based) of the wanted key among the candidates
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const uint32_t bins = 1u << widths[pass];
    for (int b = tid; b < kSelBins; b += kSelThreads) hist[b] = 0;
    __syncthreads();
    if (pass == 0) {
      for_each([&](uint32_t key, int, bool valid) { hist_add<true>(hist, key >> 21, valid); });
    } else {
      for_each([&](uint32_t key, int, bool valid) {
        hist_add<false>(hist, (key >> shift) & (bins - 1), valid && (key & mask) == prefix);
      });
    }
    __syncthreads();
    // inclusive scan of the 2048 bins: two bins per thread, wave scan, wave totals
    const unsigned int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    unsigned int incl = h0 + h1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const unsigned int end1 = before + incl, end0 = end1 - h1, start0 = end0 - h0;
    if (start0 < remaining && remaining <= end0) { s_bin = 2 * tid; s_before = start0; }
    else if (end0 < remaining && remaining <= end1) { s_bin = 2 * tid + 1; s_before = end0; }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= (bins - 1) << shift;
    remaining -= s_before;
    __syncthreads();
  }
  // prefix: key of the k-th element; remaining: how many elements equal to it are taken.  The output is in
  // ascending index order: (1) per (chunk, wave) counts of "below" and "equal" keys, (2) one exclusive scan
  // of that table in (chunk, wave) order, (3) every lane places its element from its wave's offsets.
  const uint32_t T = prefix;
  const int entries = ((n + kSelThreads - 1) / kSelThreads) * (kSelThreads / 64);
  for_each([&](uint32_t key, int i, bool valid) {
    const unsigned long long blt = __ballot(valid && key < T), beq = __ballot(valid && key == T);
    if (lane == 0) {
      const int e = (i / kSelThreads) * (kSelThreads / 64) + wave;
      table[2 * e] = __popcll(blt);
      table[2 * e + 1] = __popcll(beq);
    }
  });
  __syncthreads();
  unsigned int run_lt = 0, run_eq = 0;                  // totals of the entries before the current batch
  for (int e0 = 0; e0 < entries; e0 += kSelThreads) {
    const int e = e0 + tid;
    const unsigned int c_lt = e < entries ? table[2 * e] : 0u, c_eq = e < entries ? table[2 * e + 1] : 0u;
    unsigned int i_lt = c_lt, i_eq = c_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int a = __shfl_up(i_lt, o, 64), b = __shfl_up(i_eq, o, 64);
      if (lane >= o) { i_lt += a; i_eq += b; }
    }
    __syncthreads();
    if (lane == 63) { hist[2 * wave] = i_lt; hist[2 * wave + 1] = i_eq; }
    __syncthreads();
    unsigned int b_lt = run_lt, b_eq = run_eq, t_lt = 0, t_eq = 0;
    for (int w = 0; w < kSelThreads / 64; ++w) {
      if (w < wave) { b_lt += hist[2 * w]; b_eq += hist[2 * w + 1]; }
      t_lt += hist[2 * w]; t_eq += hist[2 * w + 1];
    }
    if (e < entries) { table[2 * e] = b_lt + i_lt - c_lt; table[2 * e + 1] = b_eq + i_eq - c_eq; }   // exclusive
    run_lt += t_lt; run_eq += t_eq;
  }
  __syncthreads();
  for_each([&](uint32_t key, int i, bool valid) {
    const bool lt = valid && key < T, eq = valid && key == T;
    const unsigned long long blt = __ballot(lt), beq = __ballot(eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int e = (i / kSelThreads) * (kSelThreads / 64) + wave;
    const unsigned int my_lt = table[2 * e] + __popcll(blt & below);
    const unsigned int my_eq = table[2 * e + 1] + __popcll(beq & below);
    if (lt) out[my_lt + min(my_eq, remaining)] = i;
    else if (eq && my_eq < remaining) out[my_lt + my_eq] = i;
  });
}

// The same selection with the keys held in registers: up to 65 536 keys are 64 per lane of a 1024-thread workgroup,
// so memory is read once instead of five times and every pass runs at the issue rate of the compute unit.  Wave w
// owns the contiguous keys [w*64*per, (w+1)*64*per), lane l its positions u*64 + l: loads are coalesced, and the
// ascending index order of the output is (wave, u, lane), which needs only a running per-wave count (a ballot and a
// scalar population count per step) and one 16-entry scan over the waves — no per-chunk table.
// The second and third histogram pass only see the keys that share the prefix found so far; a step of a wave with no
// such key costs a compare and a branch.
template <int E>
__global__ __launch_bounds__(kSelThreads) void select_reg_kernel(const float* __restrict__ x, int n, int per, int k,
                                                                 int largest, int64_t* __restrict__ out) {
  __shared__ unsigned int hist[kSelBins];
  __shared__ unsigned int wave_tot[kSelThreads / 64];
  __shared__ unsigned int wcnt[kSelThreads / 64][2];
  __shared__ unsigned int s_bin, s_before;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lg = largest != 0;
  const int first = wave * 64 * per + lane;              // index of this lane's key at step 0
  const int left = n - first;                            // step u holds a key iff u*64 < left
  uint32_t kk[E];
#pragma unroll
  for (int u = 0; u < E; ++u) {
    kk[u] = 0xffffffffu;
    if (u < per) {                                       // uniform
      const int i = first + u * 64;
      if (i < n) kk[u] = select_key(x[i], lg);
    }
  }
  uint32_t prefix = 0, mask = 0;
  unsigned int remaining = (unsigned int)k;              // rank (1-based) of the wanted key among the candidates
#pragma nounroll
  for (int pass = 0; pass < 3; ++pass) {                 // not unrolled: the last two passes share their code
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t bins = pass == 2 ? 1024u : 2048u;
    for (int b = tid; b < kSelBins; b += kSelThreads) hist[b] = 0;
    __syncthreads();
    if (pass == 0) {
#pragma unroll
      for (int u = 0; u < E; ++u)
        if (u < per) hist_add<true>(hist, kk[u] >> 21, u * 64 < left);
    } else {
#pragma unroll
      for (int u = 0; u < E; ++u)
        if (u < per) {
          const bool cand = u * 64 < left && (kk[u] & mask) == prefix;
          if (__ballot(cand)) hist_add<true>(hist, (kk[u] >> shift) & (bins - 1), cand);
        }
    }
    __syncthreads();
    const unsigned int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    unsigned int incl = h0 + h1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const unsigned int end1 = before + incl, end0 = end1 - h1, start0 = end0 - h0;
    if (start0 < remaining && remaining <= end0) { s_bin = 2 * tid; s_before = start0; }
    else if (end0 < remaining && remaining <= end1) { s_bin = 2 * tid + 1; s_before = end0; }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= (bins - 1) << shift;
    remaining -= s_before;
    __syncthreads();
  }
  // prefix: key of the k-th element; remaining: how many elements equal to it are taken
  const uint32_t T = prefix;
  unsigned int c_lt = 0, c_eq = 0;                       // wave-uniform
#pragma unroll
  for (int u = 0; u < E; ++u)
    if (u < per) {
      const bool valid = u * 64 < left;
      c_lt += __popcll(__ballot(valid && kk[u] < T));
      c_eq += __popcll(__ballot(valid && kk[u] == T));
    }
  if (lane == 0) { wcnt[wave][0] = c_lt; wcnt[wave][1] = c_eq; }
  __syncthreads();
  unsigned int off_lt = 0, off_eq = 0;
  for (int w = 0; w < wave; ++w) { off_lt += wcnt[w][0]; off_eq += wcnt[w][1]; }
#pragma unroll
  for (int u = 0; u < E; ++u)
    if (u < per) {
      const bool valid = u * 64 < left;
      const bool lt = valid && kk[u] < T, eq = valid && kk[u] == T;
      const unsigned long long blt = __ballot(lt), beq = __ballot(eq);
      if (blt | beq) {
        const unsigned int my_lt = off_lt + __builtin_amdgcn_mbcnt_hi((unsigned int)(blt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)blt, 0u));
        const unsigned int my_eq = off_eq + __builtin_amdgcn_mbcnt_hi((unsigned int)(beq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)beq, 0u));
        const int i = first + u * 64;
        if (lt) out[my_lt + min(my_eq, remaining)] = i;
        else if (eq && my_eq < remaining) out[my_lt + my_eq] = i;
        off_lt += __popcll(blt);
        off_eq += __popcll(beq);
      }
    }
}

}  // namespace

// evox_hip_select for n <= 65 536 with the keys resident in registers (one read of the keys).
EVOX_API int evox_hip_select_reg(const float* x, int n, int k, int largest, int64_t* out, void* stream) {
  if (n <= 0 || k < 1 || k > n || n > (1 << 16)) return (int)hipErrorInvalidValue;
  const int per = (n + kSelThreads - 1) / kSelThreads;
  hipStream_t s = (hipStream_t)stream;
  if (per <= 8) select_reg_kernel<8><<<1, kSelThreads, 0, s>>>(x, n, per, k, largest, out);
  else if (per <= 16) select_reg_kernel<16><<<1, kSelThreads, 0, s>>>(x, n, per, k, largest, out);
  else if (per <= 32) select_reg_kernel<32><<<1, kSelThreads, 0, s>>>(x, n, per, k, largest, out);
  else select_reg_kernel<64><<<1, kSelThreads, 0, s>>>(x, n, per, k, largest, out);
  return evox::last_launch_status();
}

// out: k indices (int64, ascending) of the k smallest (largest != 0: largest) of the n keys; 1 <= k <= n <= 2^20.
EVOX_API int evox_hip_select(const float* x, int n, int k, int largest, int64_t* out, void* stream) {
  if (n <= 0 || k < 1 || k > n || n > (1 << 20)) return (int)hipErrorInvalidValue;
  const int chunks = (n + kSelThreads - 1) / kSelThreads;
  const size_t lds = (size_t)chunks * (kSelThreads / 64) * 2 * sizeof(unsigned int);    // ≤ 128 KB at n = 2^20
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  select_kernel<<<1, kSelThreads, lds, (hipStream_t)stream>>>(x, n, k, largest, out);
  return evox::last_launch_status();
}

// The same selection spread over the chip for long key vectors (a single workgroup needs 1.3 ns per key): every
// pass is one launch of many workgroups that histogram their slice in LDS and add it to a global histogram; the
// workgroup that finishes last (a ticket counter) scans the histogram and stores the refined prefix for the next
// launch.  Two more launches count the selected keys per block (the last block scans the counts) and write the
// indices out in ascending order.  Five launches and one small zeroed workspace per call.
namespace {

constexpr int kMsThreads = 256;
constexpr int kMsPerBlock = 4096;      // keys per workgroup: 16 per thread, in chunks of 256 consecutive keys

struct SelState {
  unsigned int prefix, mask, remaining, ticket;
  unsigned int hist[3][kSelBins];
};

__device__ __forceinline__ unsigned int agent_load(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kMsThreads) void ms_hist_kernel(const float* __restrict__ x, int n, int largest, int k, int pass,
                                                             SelState* __restrict__ st) {
  __shared__ unsigned int hist[kSelBins];
  __shared__ unsigned int wave_tot[kMsThreads / 64];
  __shared__ unsigned int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lg = largest != 0;
  const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
  const uint32_t bins = pass == 2 ? 1024u : 2048u;
  const uint32_t prefix = pass == 0 ? 0u : st->prefix, mask = pass == 0 ? 0u : st->mask;
  for (int b = tid; b < kSelBins; b += kMsThreads) hist[b] = 0;
  __syncthreads();
  const int base = blockIdx.x * kMsPerBlock;
  uint32_t kk[kMsPerBlock / kMsThreads];
#pragma unroll
  for (int u = 0; u < kMsPerBlock / kMsThreads; ++u) {
    const int i = base + u * kMsThreads + tid;
    kk[u] = i < n ? select_key(x[i], lg) : 0xffffffffu;
  }
#pragma unroll
  for (int u = 0; u < kMsPerBlock / kMsThreads; ++u) {
    const bool valid = base + u * kMsThreads + tid < n;
    if (pass == 0) hist_add<true>(hist, kk[u] >> 21, valid);
    else hist_add<false>(hist, (kk[u] >> shift) & (bins - 1), valid && (kk[u] & mask) == prefix);
  }
  __syncthreads();
  for (int b = tid; b < kSelBins; b += kMsThreads)
    if (hist[b]) atomicAdd(&st->hist[pass][b], hist[b]);
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  // last workgroup: scan the complete histogram (eight bins per thread) and refine the prefix
  __threadfence();
  const unsigned int remaining = pass == 0 ? (unsigned int)k : st->remaining;
  unsigned int h[kSelBins / kMsThreads], incl = 0;
#pragma unroll
  for (int q = 0; q < kSelBins / kMsThreads; ++q) { h[q] = agent_load(&st->hist[pass][tid * (kSelBins / kMsThreads) + q]); incl += h[q]; }
  const unsigned int mine = incl;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned int start = incl - mine;
  for (int w = 0; w < wave; ++w) start += wave_tot[w];
#pragma unroll
  for (int q = 0; q < kSelBins / kMsThreads; ++q) {
    if (start < remaining && remaining <= start + h[q]) {
      st->prefix = prefix | ((uint32_t)(tid * (kSelBins / kMsThreads) + q) << shift);
      st->mask = mask | ((bins - 1) << shift);
      st->remaining = remaining - start;
    }
    start += h[q];
  }
  if (tid == 0) st->ticket = 0;
}

// blockcnt[2b], [2b+1]: keys of block b below / equal to the threshold; the last block turns them into exclusive
// offsets (blocks in index order).
__global__ __launch_bounds__(kMsThreads) void ms_count_kernel(const float* __restrict__ x, int n, int largest,
                                                              SelState* __restrict__ st, unsigned int* __restrict__ blockcnt) {
  __shared__ unsigned int red[kMsThreads / 64][2];
  __shared__ unsigned int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lg = largest != 0;
  const uint32_t T = st->prefix;
  const int base = blockIdx.x * kMsPerBlock;
  unsigned int lt = 0, eq = 0;
#pragma unroll
  for (int u = 0; u < kMsPerBlock / kMsThreads; ++u) {
    const int i = base + u * kMsThreads + tid;
    if (i < n) { const uint32_t key = select_key(x[i], lg); lt += key < T; eq += key == T; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); eq += __shfl_xor(eq, o, 64); }
  if (lane == 0) { red[wave][0] = lt; red[wave][1] = eq; }
  __syncthreads();
  if (tid == 0) {
    unsigned int a = 0, b = 0;
    for (int w = 0; w < kMsThreads / 64; ++w) { a += red[w][0]; b += red[w][1]; }
    __hip_atomic_store(&blockcnt[2 * blockIdx.x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&blockcnt[2 * blockIdx.x + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // exclusive scan of the per-block counts, kMsThreads blocks at a time
  unsigned int run_lt = 0, run_eq = 0;
  const int blocks = gridDim.x;
  for (int b0 = 0; b0 < blocks; b0 += kMsThreads) {
    const int b = b0 + tid;
    const unsigned int c_lt = b < blocks ? agent_load(&blockcnt[2 * b]) : 0u, c_eq = b < blocks ? agent_load(&blockcnt[2 * b + 1]) : 0u;
    unsigned int i_lt = c_lt, i_eq = c_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int a = __shfl_up(i_lt, o, 64), e = __shfl_up(i_eq, o, 64);
      if (lane >= o) { i_lt += a; i_eq += e; }
    }
    __syncthreads();
    if (lane == 63) { red[wave][0] = i_lt; red[wave][1] = i_eq; }
    __syncthreads();
    unsigned int b_lt = run_lt, b_eq = run_eq, t_lt = 0, t_eq = 0;
    for (int w = 0; w < kMsThreads / 64; ++w) {
      if (w < wave) { b_lt += red[w][0]; b_eq += red[w][1]; }
      t_lt += red[w][0]; t_eq += red[w][1];
    }
    if (b < blocks) { blockcnt[2 * b] = b_lt + i_lt - c_lt; blockcnt[2 * b + 1] = b_eq + i_eq - c_eq; }
    run_lt += t_lt; run_eq += t_eq;
  }
  if (tid == 0) st->ticket = 0;
}

__global__ __launch_bounds__(kMsThreads) void ms_emit_kernel(const float* __restrict__ x, int n, int largest,
                                                             const SelState* __restrict__ st,
                                                             const unsigned int* __restrict__ blockoff, int64_t* __restrict__ out) {
  __shared__ unsigned int cnt[kMsPerBlock / kMsThreads][kMsThreads / 64][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lg = largest != 0;
  const uint32_t T = st->prefix;
  const unsigned int remaining = st->remaining;
  const int base = blockIdx.x * kMsPerBlock;
  uint32_t kk[kMsPerBlock / kMsThreads];
  unsigned long long blt[kMsPerBlock / kMsThreads], beq[kMsPerBlock / kMsThreads];
#pragma unroll
  for (int u = 0; u < kMsPerBlock / kMsThreads; ++u) {
    const int i = base + u * kMsThreads + tid;
    kk[u] = i < n ? select_key(x[i], lg) : 0xffffffffu;
    blt[u] = __ballot(i < n && kk[u] < T);
    beq[u] = __ballot(i < n && kk[u] == T);
    if (lane == 0) { cnt[u][wave][0] = __popcll(blt[u]); cnt[u][wave][1] = __popcll(beq[u]); }
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned int off_lt = blockoff[2 * blockIdx.x], off_eq = blockoff[2 * blockIdx.x + 1];
#pragma unroll
  for (int u = 0; u < kMsPerBlock / kMsThreads; ++u) {
    unsigned int w_lt = 0, w_eq = 0, t_lt = 0, t_eq = 0;
    for (int w = 0; w < kMsThreads / 64; ++w) {
      if (w < wave) { w_lt += cnt[u][w][0]; w_eq += cnt[u][w][1]; }
      t_lt += cnt[u][w][0]; t_eq += cnt[u][w][1];
    }
    const int i = base + u * kMsThreads + tid;
    const bool lt = i < n && kk[u] < T, eq = i < n && kk[u] == T;
    const unsigned int my_lt = off_lt + w_lt + __popcll(blt[u] & below);
    const unsigned int my_eq = off_eq + w_eq + __popcll(beq[u] & below);
    if (lt) out[my_lt + min(my_eq, remaining)] = i;
    else if (eq && my_eq < remaining) out[my_lt + my_eq] = i;
    off_lt += t_lt; off_eq += t_eq;
  }
}

}  // namespace

// The chip-wide selection in ONE launch for up to 262 144 keys: at most 64 workgroups of 256 threads, each with its
// 4096 keys in registers, meet at four grid-wide barriers (after each histogram pass has been added to the global
// histogram, and after the per-block counts are stored).  Five launches and their gaps cost ≈45 µs on the stream for
// work that takes a few µs; a barrier is one atomic and a poll per workgroup.  All workgroups of so small a grid
// are resident together, so the barrier completes as soon as the slowest one arrives; the wait is nevertheless
// bounded (spin_limit polls): a workgroup that gives up counts itself in `timeouts`, zero-fills its share of the
// output and leaves, and so do all others when their own wait runs out.
namespace {

constexpr int kGsThreads = 256;
constexpr int kGsE = 16;                                 // keys per lane
constexpr int kGsPerBlock = kGsThreads * kGsE;
constexpr int kGsMaxBlocks = 64;                         // one lane per preceding workgroup in the offset sum

struct GridSel {
  unsigned int arrive;                                   // barrier arrivals, monotonic over the barriers of a call
  unsigned int timeouts;                                 // workgroups that gave up waiting
  unsigned int pad[2];
  unsigned int hist[3][kSelBins];
  unsigned int blockcnt[kGsMaxBlocks][2];
};

// true when all workgroups have arrived; the same value in every thread of the workgroup
__device__ __forceinline__ bool grid_wait(GridSel* g, unsigned int target, unsigned int* s_ok, int spin_limit) {
  // Everything the workgroups exchange is written with device-scope atomics, which are performed where all of them
  // see it; what the arrival needs is that this workgroup's own atomics have been acknowledged, so every thread
  // waits for its outstanding memory operations — no cache write-back, which a fence of device scope would add.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    int budget = spin_limit;
    if (atomicAdd(&g->arrive, 1u) + 1u < target)
      while (agent_load(&g->arrive) < target && --budget > 0) __builtin_amdgcn_s_sleep(2);
    if (budget <= 0) atomicAdd(&g->timeouts, 1u);
    *s_ok = budget > 0 ? 1u : 0u;
  }
  __syncthreads();
  return *s_ok != 0u;
}

__global__ __launch_bounds__(kGsThreads) void select_grid_kernel(const float* __restrict__ x, int n, int k, int largest,
                                                                 int64_t* __restrict__ out, GridSel* __restrict__ g,
                                                                 int spin_limit) {
  __shared__ unsigned int hist[kSelBins];
  __shared__ unsigned int wave_tot[kGsThreads / 64];
  __shared__ unsigned int wcnt[kGsThreads / 64][2];
  __shared__ unsigned int s_bin, s_before, s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blocks = gridDim.x;
  const bool lg = largest != 0;
  // wave w of block b owns the contiguous keys [b*4096 + w*1024, +1024): ascending index = (block, wave, u, lane)
  const int first = blockIdx.x * kGsPerBlock + wave * (64 * kGsE) + lane;
  const int left = n - first;
  uint32_t kk[kGsE];
#pragma unroll
  for (int u = 0; u < kGsE; ++u) {
    const int i = first + u * 64;
    kk[u] = i < n ? select_key(x[i], lg) : 0xffffffffu;
  }
  auto give_up = [&]() {
    for (int j = blockIdx.x * kGsThreads + tid; j < k; j += blocks * kGsThreads) out[j] = 0;
  };
  uint32_t prefix = 0, mask = 0;
  unsigned int remaining = (unsigned int)k, arrivals = 0;
#pragma nounroll
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const uint32_t bins = pass == 2 ? 1024u : 2048u;
    for (int b = tid; b < kSelBins; b += kGsThreads) hist[b] = 0;
    __syncthreads();
    if (pass == 0) {
#pragma unroll
      for (int u = 0; u < kGsE; ++u) hist_add<true>(hist, kk[u] >> 21, u * 64 < left);
    } else {
#pragma unroll
      for (int u = 0; u < kGsE; ++u) {
        const bool cand = u * 64 < left && (kk[u] & mask) == prefix;
        if (__ballot(cand)) hist_add<true>(hist, (kk[u] >> shift) & (bins - 1), cand);
      }
    }
    __syncthreads();
    for (int b = tid; b < kSelBins; b += kGsThreads)
      if (hist[b]) atomicAdd(&g->hist[pass][b], hist[b]);
    arrivals += blocks;
    if (!grid_wait(g, arrivals, &s_ok, spin_limit)) { give_up(); return; }
    // every workgroup scans the complete histogram (eight bins per thread) and refines the prefix the same way
    unsigned int h[kSelBins / kGsThreads], incl = 0;
#pragma unroll
    for (int q = 0; q < kSelBins / kGsThreads; ++q) { h[q] = agent_load(&g->hist[pass][tid * (kSelBins / kGsThreads) + q]); incl += h[q]; }
    const unsigned int mine = incl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned int start = incl - mine;
    for (int w = 0; w < wave; ++w) start += wave_tot[w];
#pragma unroll
    for (int q = 0; q < kSelBins / kGsThreads; ++q) {
      if (start < remaining && remaining <= start + h[q]) { s_bin = tid * (kSelBins / kGsThreads) + q; s_before = start; }
      start += h[q];
    }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= (bins - 1) << shift;
    remaining -= s_before;
    __syncthreads();
  }
  const uint32_t T = prefix;
  unsigned int c_lt = 0, c_eq = 0;                       // wave-uniform
#pragma unroll
  for (int u = 0; u < kGsE; ++u) {
    const bool valid = u * 64 < left;
    c_lt += __popcll(__ballot(valid && kk[u] < T));
    c_eq += __popcll(__ballot(valid && kk[u] == T));
  }
  if (lane == 0) { wcnt[wave][0] = c_lt; wcnt[wave][1] = c_eq; }
  __syncthreads();
  if (tid == 0) {
    unsigned int a = 0, b = 0;
    for (int w = 0; w < kGsThreads / 64; ++w) { a += wcnt[w][0]; b += wcnt[w][1]; }
    __hip_atomic_store(&g->blockcnt[blockIdx.x][0], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&g->blockcnt[blockIdx.x][1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  arrivals += blocks;
  if (!grid_wait(g, arrivals, &s_ok, spin_limit)) { give_up(); return; }
  // keys selected before this wave: the preceding workgroups (one per lane), then the preceding waves
  unsigned int off_lt = lane < (int)blockIdx.x ? agent_load(&g->blockcnt[lane][0]) : 0u;
  unsigned int off_eq = lane < (int)blockIdx.x ? agent_load(&g->blockcnt[lane][1]) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { off_lt += __shfl_xor(off_lt, o, 64); off_eq += __shfl_xor(off_eq, o, 64); }
  for (int w = 0; w < wave; ++w) { off_lt += wcnt[w][0]; off_eq += wcnt[w][1]; }
#pragma unroll
  for (int u = 0; u < kGsE; ++u) {
    const bool valid = u * 64 < left;
    const bool lt = valid && kk[u] < T, eq = valid && kk[u] == T;
    const unsigned long long blt = __ballot(lt), beq = __ballot(eq);
    if (blt | beq) {
      const unsigned int my_lt = off_lt + __builtin_amdgcn_mbcnt_hi((unsigned int)(blt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)blt, 0u));
      const unsigned int my_eq = off_eq + __builtin_amdgcn_mbcnt_hi((unsigned int)(beq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)beq, 0u));
      const int i = first + u * 64;
      if (lt) out[my_lt + min(my_eq, remaining)] = i;
      else if (eq && my_eq < remaining) out[my_lt + my_eq] = i;
      off_lt += __popcll(blt);
      off_eq += __popcll(beq);
    }
  }
}

}  // namespace

// Bytes of zero-initialised workspace of evox_hip_select_grid, and the most keys it takes.
EVOX_API int64_t evox_hip_select_grid_ws() { return (int64_t)sizeof(GridSel); }
EVOX_API int evox_hip_select_grid_max() { return kGsMaxBlocks * kGsPerBlock; }

// Like evox_hip_select, one launch of ceil(n / 4096) <= 64 co-operating workgroups; ws: zeroed workspace (its second
// 32-bit word counts workgroups whose bounded wait ran out: the output is then zero-filled, not a selection).
// spin_limit <= 0: the default bound (about a second).
EVOX_API int evox_hip_select_grid(const float* x, int n, int k, int largest, int64_t* out, void* ws, int spin_limit,
                                  void* stream) {
  if (n <= 0 || k < 1 || k > n || ws == nullptr || n > kGsMaxBlocks * kGsPerBlock) return (int)hipErrorInvalidValue;
  const int blocks = (n + kGsPerBlock - 1) / kGsPerBlock;
  select_grid_kernel<<<blocks, kGsThreads, 0, (hipStream_t)stream>>>(x, n, k, largest, out, (GridSel*)ws,
                                                                    spin_limit > 0 ? spin_limit : (1 << 20));
  return evox::last_launch_status();
}

// Bytes of zero-initialised workspace evox_hip_select_wide needs for n keys.
EVOX_API int64_t evox_hip_select_wide_ws(int n) {
  const int blocks = (n + kMsPerBlock - 1) / kMsPerBlock;
  return (int64_t)sizeof(SelState) + (int64_t)blocks * 2 * sizeof(unsigned int);
}

// Like evox_hip_select, over many workgroups; ws: zeroed workspace of evox_hip_select_wide_ws(n) bytes.
EVOX_API int evox_hip_select_wide(const float* x, int n, int k, int largest, int64_t* out, void* ws, void* stream) {
  if (n <= 0 || k < 1 || k > n || ws == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  SelState* st = (SelState*)ws;
  unsigned int* blockcnt = (unsigned int*)((char*)ws + sizeof(SelState));
  const int blocks = (n + kMsPerBlock - 1) / kMsPerBlock;
  for (int pass = 0; pass < 3; ++pass) ms_hist_kernel<<<blocks, kMsThreads, 0, s>>>(x, n, largest, k, pass, st);
  ms_count_kernel<<<blocks, kMsThreads, 0, s>>>(x, n, largest, st, blockcnt);
  ms_emit_kernel<<<blocks, kMsThreads, 0, s>>>(x, n, largest, st, blockcnt, out);
  return evox::last_launch_status();
}
