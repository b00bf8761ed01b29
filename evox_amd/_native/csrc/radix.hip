// Stable key-index radix sort for long key vectors (the counting ranks of sort.hip stop at 16 384 keys).
//
// Least-significant-digit passes of 8 bits over 32-bit keys, two launches per pass:
//
//   radix_hist     every workgroup counts the digits of its tile of consecutive keys: table[tile][digit].
//                  The first pass of a sort also builds the keys: fp32 values (optionally gathered through an incoming
//                  permutation) mapped to order-preserving unsigned integers — -0 = +0, NaN after +inf — or raw
//                  integers, so a multi-column (lexicographic, segmented) sort is a chain of single-column sorts from the
//                  least significant column on, each one permuting the result of the one before.
//   radix_scatter  every workgroup derives its own write offsets from the table (the table is small: 256 x tiles
//                  counters, every workgroup sums the part it needs — no separate scan launch, no waiting on other
//                  workgroups), then moves its tile: wavefront-private digit counters in LDS, the 64 keys of a step
//                  ranked against each other with eight ballots, steps, wavefronts and tiles in index order — the pass
//                  is stable.
//
// Reference call sites: sort_utils.py:5-12 (argsort + gather), non_dominate.py:118-158 (per-objective sorts of the
// crowding distance), non_dominate.py:218 (lexsort truncation), find_pbest.py:14-15 (top-p).
#include "hip_common.h"

namespace {

constexpr int kRadixThreads = 256;
constexpr int kRadixWaves = kRadixThreads / 64;
constexpr int kDigits = 256;
// keys per thread: a tile is KPT * 256 consecutive keys, KPT steps of 64 per wavefront.  Small tiles for short vectors
// (more workgroups), large ones for long vectors (every scatter workgroup walks the tile table once)

__device__ __forceinline__ uint32_t sortable_f32(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFEu;      // NaN: after +inf
  if (u == 0x80000000u) u = 0u;                                  // -0 -> +0
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// mode 0: keys are ready (key_in / perm_in are the double buffer of the previous pass)
// mode 1: first pass over fp32 values src_f[perm[i] * stride] (perm = identity when perm_src is null)
// mode 2: first pass over int32 values (non-negative, or any when all 32 bits are sorted: the sign bit is flipped)
struct RadixSrc {
  const uint32_t* key_in;
  const uint32_t* perm_in;
  const void* src;
  const int64_t* perm_src;     // optional incoming permutation (int64 indices)
  long long stride;            // element stride of src
  int mode;
};

__device__ __forceinline__ void fetch(const RadixSrc& s, int i, uint32_t& key, uint32_t& perm) {
  if (s.mode == 0) {
    key = s.key_in[i];
    perm = s.perm_in[i];
    return;
  }
  perm = s.perm_src != nullptr ? (uint32_t)s.perm_src[i] : (uint32_t)i;
  if (s.mode == 1) key = sortable_f32(reinterpret_cast<const float*>(s.src)[(long long)perm * s.stride]);
  else key = (uint32_t)reinterpret_cast<const int32_t*>(s.src)[(long long)perm * s.stride] ^ 0x80000000u;
}

template <int KPT>
__global__ __launch_bounds__(kRadixThreads) void radix_hist_kernel(RadixSrc s, int n, int shift, int tiles,
                                                                   uint32_t* __restrict__ key_out,
                                                                   uint32_t* __restrict__ perm_out,
                                                                   uint32_t* __restrict__ table) {
  __shared__ unsigned int hist[kDigits];
  hist[threadIdx.x] = 0;
  __syncthreads();
  constexpr int kTile = KPT * kRadixThreads;
  const int base = blockIdx.x * kTile;
#pragma unroll 4
  for (int k = threadIdx.x; k < kTile; k += kRadixThreads) {
    const int i = base + k;
    if (i < n) {
      uint32_t key, perm;
      fetch(s, i, key, perm);
      if (s.mode != 0) {                       // materialise the keys and the permutation for the passes to come
        key_out[i] = key;
        perm_out[i] = perm;
      }
      atomicAdd(&hist[(key >> shift) & 0xFFu], 1u);
    }
  }
  __syncthreads();
  table[(size_t)blockIdx.x * kDigits + threadIdx.x] = hist[threadIdx.x];
}

template <int KPT>
__global__ __launch_bounds__(kRadixThreads) void radix_scatter_kernel(const uint32_t* __restrict__ key_in,
                                                                      const uint32_t* __restrict__ perm_in, int n, int shift,
                                                                      int tiles, const uint32_t* __restrict__ table,
                                                                      uint32_t* __restrict__ key_out,
                                                                      uint32_t* __restrict__ perm_out,
                                                                      int64_t* __restrict__ perm64_out,
                                                                      int64_t* __restrict__ rank_out) {
  __shared__ unsigned int wbase[kRadixWaves][kDigits];
  __shared__ unsigned int scan[kDigits];
  __shared__ unsigned int wtot[kRadixWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- write offset of (this tile, digit tid): all smaller digits everywhere, this digit in the tiles in front ----
  constexpr int kTile = KPT * kRadixThreads;
  unsigned int total = 0, before = 0;
  {
    // one coalesced 1 KB row of the table per tile, eight rows in flight
    const uint32_t* col = table + tid;
    int b = 0;
    for (; b + 8 <= tiles; b += 8) {
      unsigned int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = col[(size_t)(b + u) * kDigits];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        total += c[u];
        before += b + u < (int)blockIdx.x ? c[u] : 0u;
      }
    }
    for (; b < tiles; ++b) {
      const unsigned int c = col[(size_t)b * kDigits];
      total += c;
      before += b < (int)blockIdx.x ? c : 0u;
    }
  }
  unsigned int incl = total;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wtot[wave] = incl;
#pragma unroll
  for (int w = 0; w < kRadixWaves; ++w) wbase[w][tid] = 0;
  __syncthreads();
  unsigned int front = 0;
  for (int w = 0; w < wave; ++w) front += wtot[w];
  scan[tid] = front + incl - total + before;          // exclusive over the digits + this digit's share of earlier tiles
  // ---- wavefront-private digit counts of the tile ----
  const int base = blockIdx.x * kTile + wave * (kTile / kRadixWaves);
  uint32_t keys[KPT];
#pragma unroll
  for (int g = 0; g < KPT; ++g) {
    const int i = base + g * 64 + lane;
    keys[g] = i < n ? key_in[i] : 0u;
    if (i < n) atomicAdd(&wbase[wave][(keys[g] >> shift) & 0xFFu], 1u);
  }
  __syncthreads();
  {
    // exclusive prefix over the wavefronts, on top of the tile's offset
    unsigned int run = scan[tid];
#pragma unroll
    for (int w = 0; w < kRadixWaves; ++w) {
      const unsigned int c = wbase[w][tid];
      wbase[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  // ---- move the keys: 64 at a time, in order ----
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int g = 0; g < KPT; ++g) {
    const int i = base + g * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (keys[g] >> shift) & 0xFFu;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long b = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? b : ~b;
    }
    if (valid) {
      const unsigned int rank = __popcll(peers & lt);
      const unsigned int pos = wbase[wave][d] + rank;
      const uint32_t p = perm_in[i];
      if (perm64_out != nullptr) {
        perm64_out[pos] = (int64_t)p;
        if (rank_out != nullptr) rank_out[p] = (int64_t)pos;
      } else {
        key_out[pos] = keys[g];
        perm_out[pos] = p;
      }
      if (rank == 0) wbase[wave][d] += (unsigned int)__popcll(peers);      // after every peer's read: LDS is in order
    }
  }
}

}  // namespace

namespace {
inline int keys_per_thread(int64_t n) { return n <= (1 << 19) ? 8 : 32; }

template <int KPT>
int radix_passes(const void* src, int mode, int64_t stride, const int64_t* perm_src, int n, int bit_lo, int bit_hi,
                 int64_t* order, int64_t* rank_out, uint32_t* ws, hipStream_t st) {
  constexpr int kTile = KPT * kRadixThreads;
  const int tiles = (n + kTile - 1) / kTile;
  uint32_t* kbuf[2] = {ws, ws + n};
  uint32_t* pbuf[2] = {ws + 2 * (size_t)n, ws + 3 * (size_t)n};
  uint32_t* table = ws + 4 * (size_t)n;
  int cur = 0;
  for (int shift = bit_lo; shift < bit_hi; shift += 8) {
    const bool first = shift == bit_lo, last = shift + 8 >= bit_hi;
    RadixSrc s{};
    if (first) {
      s.mode = mode;
      s.src = src;
      s.perm_src = perm_src;
      s.stride = stride;
    } else {
      s.mode = 0;
      s.key_in = kbuf[cur];
      s.perm_in = pbuf[cur];
    }
    radix_hist_kernel<KPT><<<tiles, kRadixThreads, 0, st>>>(s, n, shift, tiles, kbuf[cur], pbuf[cur], table);
    radix_scatter_kernel<KPT><<<tiles, kRadixThreads, 0, st>>>(kbuf[cur], pbuf[cur], n, shift, tiles, table, kbuf[cur ^ 1],
                                                              pbuf[cur ^ 1], last ? order : nullptr,
                                                              last ? rank_out : nullptr);
    cur ^= 1;
  }
  return evox::last_launch_status();
}
}  // namespace

EVOX_API int64_t evox_hip_radix_ws_words(int64_t n) {
  const int64_t tile = (int64_t)keys_per_thread(n) * kRadixThreads;
  return 4 * n + kDigits * ((n + tile - 1) / tile);
}

// order[r] = index (into the source) of the r-th smallest key, stable; rank_out (optional) its inverse.
//   mode 1: src = fp32 values, element i at src[i * stride]; mode 2: src = int32 values.
//   perm_src (optional, int64): the sort permutes this sequence of indices instead of 0..n-1 — the result of a sort by a
//   less significant column.
//   bit_lo, bit_hi: the key bits that take part (multiples of 8; floats: 0, 32).
//   ws: evox_hip_radix_ws_words(n) 32-bit words.
EVOX_API int evox_hip_radix_argsort(const void* src, int mode, int64_t stride, const int64_t* perm_src, int n, int bit_lo,
                                    int bit_hi, int64_t* order, int64_t* rank_out, uint32_t* ws, void* stream) {
  if (n <= 0) return 0;
  if ((mode != 1 && mode != 2) || bit_lo < 0 || bit_hi > 32 || bit_lo >= bit_hi || (bit_lo & 7) || (bit_hi & 7) ||
      src == nullptr || order == nullptr || ws == nullptr)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (keys_per_thread(n) == 8) return radix_passes<8>(src, mode, stride, perm_src, n, bit_lo, bit_hi, order, rank_out, ws, st);
  return radix_passes<32>(src, mode, stride, perm_src, n, bit_lo, bit_hi, order, rank_out, ws, st);
}
