// One-at-a-time environmental-selection loops as single persistent launches.
//
// IBEA removal, SPEA2 truncation and SRA's stochastic bubble sort update the scores of the
// survivors after every step, so their length is inherent.  Launching one kernel per step would
// cost more in launch boundaries than in work; instead ONE workgroup (1024 threads = 16
// wavefronts) runs the whole loop, synchronising with workgroup barriers only: the working set
// (a score vector, one matrix row per step) lives in LDS / L2 and no other workgroup is
// involved, so no inter-workgroup hand-off is needed.  Every loop has a fixed trip count.
#include "hip_common.h"
#include "sr_coins.h"

using namespace evox;

namespace {

constexpr int NT = 1024;

struct ArgMin { float v; int i; };

__device__ __forceinline__ ArgMin block_argmin(float v, int i, float* s_v, int* s_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                        // protect s_v / s_i of the previous call
  if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
  __syncthreads();
  ArgMin r{s_v[0], s_i[0]};
  for (int w = 1; w < NT / 64; ++w)
    if (s_v[w] < r.v || (s_v[w] == r.v && s_i[w] < r.i)) { r.v = s_v[w]; r.i = s_i[w]; }
  return r;
}

// IBEA removal: drop the member with the smallest fitness, add back the indicator loss it caused to
// everyone else (one matrix row), repeat.  Fitness and the live mask stay in LDS when they fit; the row
// is read with 16-byte loads when rows are 16-byte aligned.
template <bool kLds>
__global__ __launch_bounds__(NT) void ibea_remove_kernel(float* __restrict__ fit_g, const float* __restrict__ I,
                                                         const float* __restrict__ C, float kappa, int n, int n_remove,
                                                         uint8_t* __restrict__ alive_g) {
  extern __shared__ __attribute__((aligned(16))) int s_dyn[];
  __shared__ float s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  float* fit = fit_g;
  uint8_t* alive = alive_g;
  const int n4 = (n + 3) & ~3;
  if constexpr (kLds) {
    fit = reinterpret_cast<float*>(s_dyn);
    alive = reinterpret_cast<uint8_t*>(s_dyn + n4);
    for (int j = threadIdx.x; j < n; j += NT) fit[j] = fit_g[j];
  }
  for (int j = threadIdx.x; j < n; j += NT) alive[j] = 1;
  __syncthreads();
  const bool vec = (n & 3) == 0;
  for (int r = 0; r < n_remove; ++r) {
    float v = INFINITY; int idx = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += NT)
      if (alive[j]) { const float f = fit[j]; if (f < v || (f == v && j < idx)) { v = f; idx = j; } }
    const ArgMin m = block_argmin(v, idx, s_v, s_i);
    if (m.i == 0x7fffffff) break;                       // uniform: nothing left
    const float s = 1.0f / (C[m.i] * kappa);
    const float* row = I + (int64_t)m.i * n;
    if (vec) {
      const float4* row4 = reinterpret_cast<const float4*>(row);
      float4* fit4 = reinterpret_cast<float4*>(fit);
      for (int q = threadIdx.x; q < (n >> 2); q += NT) {
        const float4 d = row4[q];
        float4 f = fit4[q];
        f.x += expf(-d.x * s); f.y += expf(-d.y * s); f.z += expf(-d.z * s); f.w += expf(-d.w * s);
        fit4[q] = f;
      }
    } else {
      for (int j = threadIdx.x; j < n; j += NT) fit[j] += expf(-row[j] * s);
    }
    if (threadIdx.x == 0) alive[m.i] = 0;
    __syncthreads();
  }
  if constexpr (kLds) {
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += NT) { fit_g[j] = fit[j]; alive_g[j] = alive[j]; }
  }
}

// Niche-crowding removal (the Pareto-criterion archive of BCE): the crowding degree of member j is
// 1 − Π_{i alive} min(d_ij / r, 1), kept as a sum of logarithms plus a count of zero factors.  Each step
// drops the most crowded member x and takes its factors out of every product: row x of ``logs``, where a
// NaN entry stands for a zero factor.
template <bool kLds>
__global__ __launch_bounds__(NT) void crowd_remove_kernel(const float* __restrict__ logs, int n, int n_remove,
                                                          float* __restrict__ log_sum_g, int* __restrict__ zero_cnt_g,
                                                          uint8_t* __restrict__ alive_g) {
  extern __shared__ __attribute__((aligned(16))) int s_dyn[];
  __shared__ float s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  float* log_sum = log_sum_g;
  int* zero_cnt = zero_cnt_g;
  uint8_t* alive = alive_g;
  const int n4 = (n + 3) & ~3;
  if constexpr (kLds) {
    log_sum = reinterpret_cast<float*>(s_dyn);
    zero_cnt = s_dyn + n4;
    alive = reinterpret_cast<uint8_t*>(s_dyn + 2 * n4);
    for (int j = threadIdx.x; j < n; j += NT) { log_sum[j] = log_sum_g[j]; zero_cnt[j] = zero_cnt_g[j]; }
  }
  for (int j = threadIdx.x; j < n; j += NT) alive[j] = 1;
  __syncthreads();
  for (int r = 0; r < n_remove; ++r) {
    float v = INFINITY; int idx = 0x7fffffff;               // argmin of the negated crowding degree
    for (int j = threadIdx.x; j < n; j += NT)
      if (alive[j]) {
        const float f = zero_cnt[j] > 0 ? -1.0f : expf(log_sum[j]) - 1.0f;
        if (f < v || (f == v && j < idx)) { v = f; idx = j; }
      }
    const ArgMin m = block_argmin(v, idx, s_v, s_i);
    if (m.i == 0x7fffffff) break;
    const float* row = logs + (int64_t)m.i * n;
    for (int j = threadIdx.x; j < n; j += NT) {
      const float d = row[j];
      if (d != d) zero_cnt[j] -= 1; else log_sum[j] -= d;
    }
    if (threadIdx.x == 0) alive[m.i] = 0;
    __syncthreads();
  }
  if constexpr (kLds) {
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += NT) alive_g[j] = alive[j];
  }
}

// SPEA2 truncation.  The distance matrix is read-only (rows padded to a multiple of four floats, +inf on
// the diagonal); which points are gone is a byte mask.  Per point the kernel tracks the distance to the
// nearest live neighbour and who that is; removing x only invalidates the rows whose neighbour was x,
// and each of those is rescanned by one wave with 16-byte loads.  The three per-point arrays live in
// LDS (9 bytes a point) when they fit.
template <bool kLds>
__global__ __launch_bounds__(NT) void spea2_truncate_kernel(const float* __restrict__ dis, int n, int ld, int n_remove,
                                                            uint8_t* __restrict__ alive_g, float* __restrict__ rowmin_g,
                                                            int* __restrict__ nn_g) {
  extern __shared__ __attribute__((aligned(16))) int s_dyn[];
  __shared__ float s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  float* rowmin = rowmin_g;
  int* nn = nn_g;
  uint8_t* alive = alive_g;
  if constexpr (kLds) {
    rowmin = reinterpret_cast<float*>(s_dyn);
    nn = s_dyn + ld;
    alive = reinterpret_cast<uint8_t*>(s_dyn + 2 * ld);
    for (int j = threadIdx.x; j < ld; j += NT) {
      rowmin[j] = j < n ? rowmin_g[j] : INFINITY;
      nn[j] = j < n ? nn_g[j] : -1;
      alive[j] = alive_g[j];
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto scan_row = [&](int i) {           // one wavefront rescans one row among the live columns
    float v = INFINITY; int a = 0x7fffffff;
    const float4* row = reinterpret_cast<const float4*>(dis + (int64_t)i * ld);
    const uint32_t* al4 = reinterpret_cast<const uint32_t*>(alive);
    for (int q = lane; q < (ld >> 2); q += 64) {
      const float4 d = row[q];
      const uint32_t al = al4[q];
      if ((al & 0x000000ffu) && d.x < v) { v = d.x; a = 4 * q; }
      if ((al & 0x0000ff00u) && d.y < v) { v = d.y; a = 4 * q + 1; }
      if ((al & 0x00ff0000u) && d.z < v) { v = d.z; a = 4 * q + 2; }
      if ((al & 0xff000000u) && d.w < v) { v = d.w; a = 4 * q + 3; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64); const int oa = __shfl_xor(a, o, 64);
      if (ov < v || (ov == v && oa < a)) { v = ov; a = oa; }
    }
    if (lane == 0) { rowmin[i] = v; nn[i] = (v == INFINITY) ? -1 : a; }
  };
  for (int r = 0; r < n_remove; ++r) {
    float v = INFINITY; int idx = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += NT) { const float f = rowmin[j]; if (f < v || (f == v && j < idx)) { v = f; idx = j; } }
    const ArgMin m = block_argmin(v, idx, s_v, s_i);
    const int x = m.i == 0x7fffffff ? 0 : m.i;
    if (threadIdx.x == 0) { alive[x] = 0; rowmin[x] = INFINITY; nn[x] = -1; }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += NT) {
      const int i = i0 + threadIdx.x;
      unsigned long long hit = __ballot(i < n && nn[i] == x);
      while (hit) {                                     // wave-uniform
        const int bit = __ffsll((long long)hit) - 1;
        hit &= hit - 1;
        scan_row(i0 + wave * 64 + bit);
      }
    }
    __syncthreads();
  }
  if constexpr (kLds) {
    for (int j = threadIdx.x; j < n; j += NT) alive_g[j] = alive[j];
  }
}

// Stochastic bubble sort.  Inside one sweep every compare-swap depends on the previous one, but sweep
// s+1 may work on position j as soon as sweep s has left position j+1: the threads of the workgroup run
// consecutive sweeps as a systolic pipeline, thread g trailing thread g−1 by two positions (thread g at
// time t handles position t − 2g).  Each thread carries the element it is bubbling; in one time step the
// threads touch distinct entries of the permutation (LDS) and a thread reads what its predecessor wrote
// one step earlier, so a workgroup barrier separates the steps (a wave barrier when one wave suffices).
// The result is exactly that of the sequential sort, in ⌈sweeps / L⌉ passes of n − 1 + 2(L − 1) steps for
// L threads instead of n/2 passes of n steps by a single lane.
// KEYS: the two key vectors and the coin flips are staged in LDS too (each instantiation addresses one memory
// space only, so the loop compiles to LDS / global instructions rather than generic ones).
// FRESH: every comparison draws its own coin (sr_coins.h: a hash of seed, sweep and position; a thread refreshes its
// hash word every fourth position) instead of reading the per-position vector u.
template <int L, bool KEYS, bool FRESH>
__global__ __launch_bounds__(L) void stochastic_ranking_kernel(const float* __restrict__ I1, const float* __restrict__ I2,
                                                               const float* __restrict__ u, float pc, int n,
                                                               int64_t* __restrict__ rank, const int32_t* __restrict__ seed_ptr,
                                                               const float* __restrict__ pc_dev) {
  if (pc_dev) pc = *pc_dev;                                // the probability on the device (a recorded graph reads it here)
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  int* rk = s_mem;                                       // the permutation, n entries
  float* l1 = reinterpret_cast<float*>(s_mem + n);       // copies of I1, I2, u when KEYS
  float* l2 = l1 + n;
  float* lu = l2 + n;
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += L) {
    rk[i] = i;
    if constexpr (KEYS) {
      l1[i] = I1[i]; l2[i] = I2[i];
      if constexpr (!FRESH) lu[i] = u[i];
    }
  }
  __syncthreads();
  const uint32_t seed = FRESH ? (uint32_t)*seed_ptr : 0u, pc7 = sr_pc7(pc);
  uint32_t mask = 0;
  auto key1 = [&](int i) { if constexpr (KEYS) return l1[i]; else return I1[i]; };
  auto key2 = [&](int i) { if constexpr (KEYS) return l2[i]; else return I2[i]; };
  auto coin = [&](int j) { if constexpr (KEYS) return lu[j]; else return u[j]; };
  const int sweeps = (n + 1) / 2;
  const int steps = (n - 1) + 2 * (L - 1);
  for (int s0 = 0; s0 < sweeps; s0 += L) {
    const bool mine = s0 + lane < sweeps;
    int a = 0;
    float a1 = 0.f, a2 = 0.f;
    for (int t = 0; t < steps; ++t) {
      const int j = t - 2 * lane;
      if (mine && j >= 0 && j < n - 1) {
        if (j == 0) { a = rk[0]; a1 = key1(a); a2 = key2(a); }
        const int b = rk[j + 1];
        const float b1 = key1(b), b2 = key2(b);
        bool first;
        if constexpr (FRESH) {
          if ((j & 3) == 0) mask = sr_coin_mask(sr_coin_group(seed, (uint32_t)(s0 + lane), (uint32_t)j >> 2), pc7);
          first = (mask >> (8 * (j & 3) + 7)) & 1u;
        } else {
          first = coin(j) < pc;
        }
        if ((first ? a1 : a2) < (first ? b1 : b2)) {
          rk[j] = b;                                       // b moves left, a keeps bubbling right
        } else {
          rk[j] = a;
          a = b; a1 = b1; a2 = b2;
        }
        if (j == n - 2) rk[n - 1] = a;
      }
      __syncthreads();                                     // the next step reads what this one wrote
    }
  }
  for (int i = lane; i < n; i += L) rank[i] = rk[i];
}

// The same sort pipelined over POSITIONS instead of sweeps, for long permutations.  Thread t keeps B consecutive
// entries (element index and both keys) in registers and owns the comparators between them and the one to its right
// neighbour's first entry.  A sweep enters a thread as the element `a` that its left neighbour carried out, walks the
// B comparators in registers, and leaves as the carried element for the right neighbour.  Sweep s may run in thread t
// once thread t-1 has finished sweep s and thread t+1 sweep s-1 (whose first entry it reads): thread t runs sweep s at
// macro step t + 2s.  A macro step is B compare-swaps in registers, six LDS words exchanged and one barrier, so the
// whole sort takes T + 2·sweeps barriers instead of sweeps·n/T·... one per single compare-swap.  The array is padded
// to T·B entries with sentinels (both keys -inf) that no element can pass, which keeps every thread's range full.
// Exactly the sequential result (a sweep without swaps changes nothing, so running all ⌈n/2⌉ sweeps is the same as
// stopping early).
// FRESH: the coins of a thread's B comparators in sweep s are B/4 hash words (B is a multiple of 4), turned into
// four coins each by one carry-free byte-wise comparison.
template <int B, bool FRESH>
__global__ __launch_bounds__(1024) void stochastic_ranking_blocked_kernel(const float* __restrict__ I1,
                                                                          const float* __restrict__ I2,
                                                                          const float* __restrict__ u, float pc, int n,
                                                                          int64_t* __restrict__ rank,
                                                                          const int32_t* __restrict__ seed_ptr,
                                                                          const float* __restrict__ pc_dev) {
  static_assert(B % 4 == 0, "coin groups hold four positions");
  if (pc_dev) pc = *pc_dev;
  constexpr int T = 1024;
  __shared__ int c_i[T], f_i[T];
  __shared__ float c_1[T], c_2[T], f_1[T], f_2[T];
  // Logical thread t works at macro steps of the parity of t: threads of equal parity share wavefronts (even waves
  // hold the even t, odd waves the odd t), so that a wavefront is either fully at work or skips the step.
  const int t = 2 * ((threadIdx.x >> 7) * 64 + (threadIdx.x & 63)) + ((threadIdx.x >> 6) & 1);
  const int threads = (n + B - 1) / B;                   // threads that hold real entries
  int xi[B];
  float x1[B], x2[B];
  unsigned use_first = 0;                                // bit q: comparator between entries q and q+1 ranks by I1
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const int p = t * B + q;
    xi[q] = p < n ? p : -1;
    x1[q] = p < n ? I1[p] : -INFINITY;
    x2[q] = p < n ? I2[p] : -INFINITY;
    if constexpr (!FRESH) {
      if (p < n - 1 && u[p] < pc) use_first |= 1u << q;
    }
  }
  const uint32_t seed = FRESH ? (uint32_t)*seed_ptr : 0u, pc7 = sr_pc7(pc);
  f_i[t] = xi[0]; f_1[t] = x1[0]; f_2[t] = x2[0];
  __syncthreads();
  const int sweeps = (n + 1) / 2;
  const int steps = threads + 2 * sweeps;
  const bool has_work = t < threads;
  for (int step = 0; step < steps; ++step) {
    const int d = step - t;
    if (has_work && d >= 0 && !(d & 1) && (d >> 1) < sweeps) {
      int ai; float a1, a2;
      if (t == 0) { ai = xi[0]; a1 = x1[0]; a2 = x2[0]; }
      else { ai = c_i[t - 1]; a1 = c_1[t - 1]; a2 = c_2[t - 1]; }
      // the right neighbour's first entry closes this thread's range (a sentinel behind the last thread)
      const bool last = t == threads - 1;
      const int ni = last ? -1 : f_i[t + 1];
      const float n1 = last ? -INFINITY : f_1[t + 1], n2 = last ? -INFINITY : f_2[t + 1];
      uint32_t masks[B / 4];
      if constexpr (FRESH) {
#pragma unroll
        for (int w = 0; w < B / 4; ++w)
          masks[w] = sr_coin_mask(sr_coin_group(seed, (uint32_t)(d >> 1), (uint32_t)(t * (B / 4) + w)), pc7);
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int bi = q + 1 < B ? xi[q + 1] : ni;
        const float b1 = q + 1 < B ? x1[q + 1] : n1, b2 = q + 1 < B ? x2[q + 1] : n2;
        const bool first = FRESH ? (masks[q >> 2] >> (8 * (q & 3) + 7)) & 1u : (use_first >> q) & 1u;
        if ((first ? a1 : a2) < (first ? b1 : b2)) {
          xi[q] = bi; x1[q] = b1; x2[q] = b2;             // b moves left, a keeps bubbling right
        } else {
          xi[q] = ai; x1[q] = a1; x2[q] = a2;
          ai = bi; a1 = b1; a2 = b2;
        }
      }
      c_i[t] = ai; c_1[t] = a1; c_2[t] = a2;
      f_i[t] = xi[0]; f_1[t] = x1[0]; f_2[t] = x2[0];
    }
    __syncthreads();
  }
  // Entry q of thread t is position t·B + q, except that every sweep left its carried element one slot to the right
  // of the thread's range: position (t+1)·B holds what thread t carried out last — that is thread t+1's entry 0 as far
  // as its own later sweeps saw it, and they all ran (thread t+1 finishes after thread t), so the registers are final.
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const int p = t * B + q;
    if (p < n) rank[p] = xi[q];
  }
}

// The blocked sort on PACKED entries.  What a compare-swap moves decides its cost: (index, key 1, key 2) is three
// conditional moves per side and two more to pick the key.  Here an entry is ONE word, (r1 << 16) | r2, with r1 and r2
// the dense ranks of the element's two keys (equal keys share a rank, so `<` on ranks is `<` on keys; 0 is the sentinel
// below every element): a comparison by key 1 looks at the upper halves, by key 2 at the lower halves, and a swap moves
// one register per side (one bit-field extract per side picks the deciding rank).  The kernel returns the packed words in their final order; which element each word stands for
// is recovered outside (elements with both keys equal are indistinguishable to the sort and never pass each other, so
// the k-th occurrence of a word in the result is the k-th such element in index order).  n <= 65 535.
template <int B, bool FRESH>
__global__ __launch_bounds__(1024) void stochastic_ranking_packed_kernel(const uint32_t* __restrict__ packed,
                                                                         const float* __restrict__ u, float pc, int n,
                                                                         uint32_t* __restrict__ out,
                                                                         const int32_t* __restrict__ seed_ptr,
                                                                         const float* __restrict__ pc_dev) {
  static_assert(B % 4 == 0, "coin groups hold four positions");
  if (pc_dev) pc = *pc_dev;
  constexpr int T = 1024;
  __shared__ uint32_t c_w[T], f_w[T];
  const int t = 2 * ((threadIdx.x >> 7) * 64 + (threadIdx.x & 63)) + ((threadIdx.x >> 6) & 1);
  const int threads = (n + B - 1) / B;
  uint32_t x[B];
  uint32_t offs[FRESH ? 1 : B];                          // per-position coins: the bit offset of each comparator's key
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const int p = t * B + q;
    x[q] = p < n ? packed[p] : 0u;
    if constexpr (!FRESH) offs[q] = p < n - 1 && u[p] < pc ? 16u : 0u;
  }
  const uint32_t seed = FRESH ? (uint32_t)*seed_ptr : 0u, pc7 = sr_pc7(pc);
  f_w[t] = x[0];
  __syncthreads();
  const int sweeps = (n + 1) / 2;
  const int steps = threads + 2 * sweeps;
  const bool has_work = t < threads;
  for (int step = 0; step < steps; ++step) {
    const int d = step - t;
    if (has_work && d >= 0 && !(d & 1) && (d >> 1) < sweeps) {
      uint32_t a = t == 0 ? x[0] : c_w[t - 1];
      const uint32_t nx = t == threads - 1 ? 0u : f_w[t + 1];
      uint32_t masks[B / 4];
      if constexpr (FRESH) {
#pragma unroll
        for (int w = 0; w < B / 4; ++w)
          masks[w] = sr_coin_mask(sr_coin_group(seed, (uint32_t)(d >> 1), (uint32_t)(t * (B / 4) + w)), pc7);
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const uint32_t b = q + 1 < B ? x[q + 1] : nx;
        // where the deciding rank starts in the word: bit 16 (key 1) when the coin is set, bit 0 (key 2) otherwise
        const uint32_t off = FRESH ? (masks[q >> 2] >> (8 * (q & 3) + 3)) & 16u : offs[q];
        const bool lt = __builtin_amdgcn_ubfe(a, off, 16u) < __builtin_amdgcn_ubfe(b, off, 16u);
        x[q] = lt ? b : a;                                 // b moves left, a keeps bubbling right
        a = lt ? a : b;
      }
      c_w[t] = a;
      f_w[t] = x[0];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const int p = t * B + q;
    if (p < n) out[p] = x[q];
  }
}

// The packed sort over SEVERAL workgroups.  One workgroup is bound by the instruction issue of its compute unit
// (2·10⁸ compare-swaps at n = 20 000); the systolic schedule — logical thread t runs sweep s at macro step t + 2s, reading
// the carry of t−1 and the first entry of t+1 from the step before — lets information travel one thread per step, so a
// window of 1024 logical threads computes its middle 1024 − 2·halo threads EXACTLY for `halo` steps without hearing
// from outside (the outer threads go stale one per step from each end, and are discarded).  Every launch advances all
// windows by up to `halo` macro steps: a workgroup loads its window (owned threads plus a halo of the neighbours'
// state), steps, and stores its owned threads into the OTHER state buffer, which the next launch reads.  The overlap costs
// 2·halo / 1024 redundant work and buys ⌈threads / own⌉ compute units instead of one; the launch boundary is the only
// synchronisation between workgroups.  State per logical thread: its B entries and the element it last carried out
// (its first entry, which the left neighbour reads, is x[0]).  Same coins, same schedule, same permutation.
template <int B, bool FRESH, int T>
__global__ __launch_bounds__(T) void stochastic_ranking_window_kernel(
    const uint32_t* __restrict__ src_x, int n_src, const uint32_t* __restrict__ src_c, uint32_t* __restrict__ dst_x, int n_dst,
    uint32_t* __restrict__ dst_c, const float* __restrict__ u, float pc, int n, const int32_t* __restrict__ seed_ptr,
    const float* __restrict__ pc_dev, int step0, int nsteps, int halo) {
  static_assert(B % 4 == 0, "coin groups hold four positions");
  if (pc_dev) pc = *pc_dev;
  __shared__ uint32_t c_w[T], f_w[T];
  const int own = T - 2 * halo;
  // local index l: threads of equal parity share wavefronts; the window starts at an even logical thread
  const int l = 2 * ((threadIdx.x >> 7) * 64 + (threadIdx.x & 63)) + ((threadIdx.x >> 6) & 1);
  // LDS slot of local thread i: the lanes of a wave hold every second thread, so the two parities are kept in separate
  // halves and a wave's reads of its neighbours' words are consecutive (no bank conflicts)
  auto slot = [](int i) { return (i & 1) * (T / 2) + (i >> 1); };
  const int t = (int)blockIdx.x * own - halo + l;
  const int threads = (n + B - 1) / B;
  const bool real = t >= 0 && t < threads;
  uint32_t x[B];
  uint32_t offs[FRESH ? 1 : B];
#pragma unroll
  for (int q = 0; q < B; ++q) {
    const int p = t * B + q;
    x[q] = real && p < n_src ? src_x[p] : 0u;
    if constexpr (!FRESH) offs[q] = real && p < n - 1 && u[p] < pc ? 16u : 0u;
  }
  const uint32_t seed = FRESH ? (uint32_t)*seed_ptr : 0u, pc7 = sr_pc7(pc);
  c_w[slot(l)] = real && src_c ? src_c[t] : 0u;
  f_w[slot(l)] = x[0];
  __syncthreads();
  const int sweeps = (n + 1) / 2;
  for (int step = step0; step < step0 + nsteps; ++step) {
    const int d = step - t;
    if (real && d >= 0 && !(d & 1) && (d >> 1) < sweeps) {
      uint32_t a = t == 0 ? x[0] : (l > 0 ? c_w[slot(l - 1)] : 0u);
      const uint32_t nx = (t == threads - 1 || l == T - 1) ? 0u : f_w[slot(l + 1)];
      uint32_t masks[B / 4];
      if constexpr (FRESH) {
#pragma unroll
        for (int w = 0; w < B / 4; ++w)
          masks[w] = sr_coin_mask(sr_coin_group(seed, (uint32_t)(d >> 1), (uint32_t)(t * (B / 4) + w)), pc7);
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const uint32_t b = q + 1 < B ? x[q + 1] : nx;
        const uint32_t off = FRESH ? (masks[q >> 2] >> (8 * (q & 3) + 3)) & 16u : offs[q];
        const bool lt = __builtin_amdgcn_ubfe(a, off, 16u) < __builtin_amdgcn_ubfe(b, off, 16u);
        x[q] = lt ? b : a;
        a = lt ? a : b;
      }
      c_w[slot(l)] = a;
      f_w[slot(l)] = x[0];
    }
    __syncthreads();
  }
  if (real && l >= halo && l < halo + own) {
#pragma unroll
    for (int q = 0; q < B; ++q) {
      const int p = t * B + q;
      if (p < n_dst) dst_x[p] = x[q];
    }
    if (dst_c) dst_c[t] = c_w[slot(l)];
  }
}

// ws: 2 * (threads * B + threads) words, threads = ceil(n / B); T: logical threads per window; halo: even, 2 * halo < T.
template <int B, int T>
int launch_ranking_window(const uint32_t* packed, const float* u, float pc, int n, uint32_t* out, const int32_t* seed,
                          const float* pc_dev, uint32_t* ws, int halo, hipStream_t st) {
  const int threads = (n + B - 1) / B, own = T - 2 * halo, groups = (threads + own - 1) / own;
  const int cap = threads * B;
  uint32_t* xs[2] = {ws, ws + cap};
  uint32_t* cs[2] = {ws + 2 * cap, ws + 2 * cap + threads};
  const int steps = threads + 2 * ((n + 1) / 2);
  int cur = 0;
  for (int step0 = 0; step0 < steps; step0 += halo) {
    const int nsteps = steps - step0 < halo ? steps - step0 : halo;
    const bool first = step0 == 0, last = step0 + halo >= steps;
    const uint32_t* sx = first ? packed : xs[cur];
    const uint32_t* sc = first ? nullptr : cs[cur];
    uint32_t* dx = last ? out : xs[cur ^ 1];
    uint32_t* dc = last ? nullptr : cs[cur ^ 1];
    if (seed)
      stochastic_ranking_window_kernel<B, true, T><<<groups, T, 0, st>>>(sx, first ? n : cap, sc, dx, last ? n : cap, dc, u, pc, n,
                                                                        seed, pc_dev, step0, nsteps, halo);
    else
      stochastic_ranking_window_kernel<B, false, T><<<groups, T, 0, st>>>(sx, first ? n : cap, sc, dx, last ? n : cap, dc, u, pc, n,
                                                                         seed, pc_dev, step0, nsteps, halo);
    cur ^= 1;
  }
  return last_launch_status();
}

template <int B>
int launch_ranking_packed(const uint32_t* packed, const float* u, float pc, int n, uint32_t* out, const int32_t* seed,
                          const float* pc_dev, hipStream_t st) {
  if (seed) stochastic_ranking_packed_kernel<B, true><<<1, 1024, 0, st>>>(packed, u, pc, n, out, seed, pc_dev);
  else stochastic_ranking_packed_kernel<B, false><<<1, 1024, 0, st>>>(packed, u, pc, n, out, seed, pc_dev);
  return last_launch_status();
}

template <int B>
int launch_ranking_blocked(const float* I1, const float* I2, const float* u, float pc, int n, int64_t* rank,
                           const int32_t* seed, const float* pc_dev, hipStream_t st) {
  if (seed) stochastic_ranking_blocked_kernel<B, true><<<1, 1024, 0, st>>>(I1, I2, u, pc, n, rank, seed, pc_dev);
  else stochastic_ranking_blocked_kernel<B, false><<<1, 1024, 0, st>>>(I1, I2, u, pc, n, rank, seed, pc_dev);
  return last_launch_status();
}

// sequential form for permutations that do not fit the LDS (n > 40 000)
__global__ __launch_bounds__(64) void stochastic_ranking_serial_kernel(const float* __restrict__ I1,
                                                                       const float* __restrict__ I2,
                                                                       const float* __restrict__ u, float pc, int n,
                                                                       int64_t* __restrict__ rank, int* __restrict__ rk,
                                                                       const int32_t* __restrict__ seed_ptr,
                                                                       const float* __restrict__ pc_dev) {
  if (pc_dev) pc = *pc_dev;
  for (int i = threadIdx.x; i < n; i += 64) rk[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int sweeps = (n + 1) / 2;
    const bool fresh = seed_ptr != nullptr;
    const uint32_t seed = fresh ? (uint32_t)*seed_ptr : 0u, pc7 = sr_pc7(pc);
    for (int s = 0; s < sweeps; ++s) {
      bool swapped = fresh;                                 // fresh coins: every sweep runs
      int a = rk[0];
      for (int j = 0; j + 1 < n; ++j) {
        const int b = rk[j + 1];
        const bool first = fresh ? sr_coin(seed, (uint32_t)s, (uint32_t)j, pc7) : u[j] < pc;
        const float* key = first ? I1 : I2;
        if (key[a] < key[b]) { rk[j] = b; swapped = true; }
        else { rk[j] = a; a = b; }
      }
      rk[n - 1] = a;
      if (!swapped) break;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 64) rank[i] = rk[i];
}


__device__ __forceinline__ long long block_sum_ll(long long v, long long* s_w) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  long long r = 0;
  for (int w = 0; w < NT / 64; ++w) r += s_w[w];
  return r;
}

// NSGA-III niche water-filling: the smallest level L with Σ_j min(c_j, max(0, L − ρ_j)) ≥ k by bisection,
// then t_j = granted(L − 1) plus one more for the first (k − Σ granted(L − 1)) niches that can still take
// one at level L, in index order.  One workgroup: the bisection is ~log2(population) block reductions
// over a few thousand niches, which as separate tensor operations costs a hundred launches.
__global__ __launch_bounds__(NT) void niche_fill_kernel(const int64_t* __restrict__ rho, const int64_t* __restrict__ cnt,
                                                        int nref, const int64_t* __restrict__ kptr,
                                                        int64_t* __restrict__ out) {
  __shared__ long long s_w[NT / 64];
  __shared__ int s_scan[NT / 64];
  const long long k = *kptr;
  long long mr = 0, mc = 0;
  for (int j = threadIdx.x; j < nref; j += NT) { mr = max(mr, (long long)rho[j]); mc = max(mc, (long long)cnt[j]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mr = max(mr, __shfl_xor(mr, o, 64)); mc = max(mc, __shfl_xor(mc, o, 64)); }
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = mr + mc;      // ≥ every ρ_j + c_j after the max below
  __syncthreads();
  long long hi = 0;
  for (int w = 0; w < NT / 64; ++w) hi = max(hi, s_w[w]);
  hi += 1;
  long long lo = 0;
  auto granted = [&](int j, long long level) -> long long {
    const long long room = level - (long long)rho[j];
    return min((long long)cnt[j], room > 0 ? room : 0ll);
  };
  while (lo < hi) {                                                   // uniform: every thread holds the same lo, hi
    const long long mid = (lo + hi) / 2;
    long long part = 0;
    for (int j = threadIdx.x; j < nref; j += NT) part += granted(j, mid);
    if (block_sum_ll(part, s_w) >= k) hi = mid; else lo = mid + 1;
  }
  const long long level = hi;
  long long part = 0;
  for (int j = threadIdx.x; j < nref; j += NT) part += granted(j, level - 1);
  const long long need = k - block_sum_ll(part, s_w);
  // running count of niches that can take one more, in index order, chunk by chunk
  long long before = 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j0 = 0; j0 < nref; j0 += NT) {
    const int j = j0 + threadIdx.x;
    long long base = 0; int ok = 0;
    if (j < nref) { base = granted(j, level - 1); ok = granted(j, level) > base; }
    int inc = ok;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    int wave_before = 0, total = 0;
    for (int w = 0; w < NT / 64; ++w) { if (w < wave) wave_before += s_scan[w]; total += s_scan[w]; }
    if (j < nref) out[j] = base + ((ok && before + wave_before + inc <= need) ? 1 : 0);
    before += total;
  }
}

}  // namespace

EVOX_API int evox_hip_ibea_remove(float* fit, const float* I, const float* C, float kappa, int n, int n_remove,
                                  uint8_t* alive, void* stream) {
  if (n <= 0) return 0;
  const size_t lds = (size_t)((n + 3) & ~3) * 5;
  if (lds <= 150 * 1024) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)ibea_remove_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    ibea_remove_kernel<true><<<1, NT, lds, (hipStream_t)stream>>>(fit, I, C, kappa, n, n_remove, alive);
  } else {
    ibea_remove_kernel<false><<<1, NT, 0, (hipStream_t)stream>>>(fit, I, C, kappa, n, n_remove, alive);
  }
  return last_launch_status();
}

// logs: (n, n), NaN where the factor is zero; log_sum / zero_cnt: column sums of the finite entries / counts
// of the NaN entries (scratch, modified); alive: out (n).
EVOX_API int evox_hip_crowd_remove(const float* logs, int n, int n_remove, float* log_sum, int* zero_cnt, uint8_t* alive,
                                   void* stream) {
  if (n <= 0) return 0;
  const size_t lds = (size_t)((n + 3) & ~3) * 9;
  if (lds <= 150 * 1024) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)crowd_remove_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    crowd_remove_kernel<true><<<1, NT, lds, (hipStream_t)stream>>>(logs, n, n_remove, log_sum, zero_cnt, alive);
  } else {
    crowd_remove_kernel<false><<<1, NT, 0, (hipStream_t)stream>>>(logs, n, n_remove, log_sum, zero_cnt, alive);
  }
  return last_launch_status();
}

// dis: (n, ld) with ld % 4 == 0; alive: ld bytes in/out (padding zero); rowmin / nn: each row's minimum over
// the live columns and its position, computed by the caller with a full-chip reduction.
EVOX_API int evox_hip_spea2_truncate(const float* dis, int n, int ld, int n_remove, uint8_t* alive, float* rowmin, int* nn,
                                     void* stream) {
  if (n <= 0 || n_remove <= 0) return 0;
  if (ld < n || (ld & 3)) return (int)hipErrorInvalidValue;
  const size_t lds = (size_t)ld * 9;
  if (lds <= 150 * 1024) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)spea2_truncate_kernel<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    spea2_truncate_kernel<true><<<1, NT, lds, (hipStream_t)stream>>>(dis, n, ld, n_remove, alive, rowmin, nn);
  } else {
    spea2_truncate_kernel<false><<<1, NT, 0, (hipStream_t)stream>>>(dis, n, ld, n_remove, alive, rowmin, nn);
  }
  return last_launch_status();
}

template <int L, bool KEYS, bool FRESH>
int launch_ranking_as(const float* I1, const float* I2, const float* u, float pc, int n, int64_t* rank, size_t lds,
                      const int32_t* seed, const float* pc_dev, hipStream_t st) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)stochastic_ranking_kernel<L, KEYS, FRESH>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  stochastic_ranking_kernel<L, KEYS, FRESH><<<1, L, lds, st>>>(I1, I2, u, pc, n, rank, seed, pc_dev);
  return last_launch_status();
}

template <int L>
int launch_ranking(const float* I1, const float* I2, const float* u, float pc, int n, int64_t* rank, size_t lds,
                   int keys_in_lds, const int32_t* seed, const float* pc_dev, hipStream_t st) {
  if (seed) {
    return keys_in_lds ? launch_ranking_as<L, true, true>(I1, I2, u, pc, n, rank, lds, seed, pc_dev, st)
                       : launch_ranking_as<L, false, true>(I1, I2, u, pc, n, rank, lds, seed, pc_dev, st);
  }
  return keys_in_lds ? launch_ranking_as<L, true, false>(I1, I2, u, pc, n, rank, lds, seed, pc_dev, st)
                     : launch_ranking_as<L, false, false>(I1, I2, u, pc, n, rank, lds, seed, pc_dev, st);
}

// u: n-1 per-position coins in [0, 1) (the comparison at position j ranks by I1 when u[j] < pc, in every sweep), or —
// seed != null, u unused — a fresh coin for every comparison, drawn from a hash of (*seed, sweep, position).
static int stochastic_ranking_any(const float* I1, const float* I2, const float* u, float pc, int n, int64_t* rank,
                                  int* scratch, const int32_t* seed, const float* pc_dev, void* stream) {
  if (n <= 0) return 0;
  if (n > 4096 && n <= 1024 * 24) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 1024 * 8) return launch_ranking_blocked<8>(I1, I2, u, pc, n, rank, seed, pc_dev, st);
    if (n <= 1024 * 16) return launch_ranking_blocked<16>(I1, I2, u, pc, n, rank, seed, pc_dev, st);
    return launch_ranking_blocked<24>(I1, I2, u, pc, n, rank, seed, pc_dev, st);
  }
  const size_t perm = (size_t)n * sizeof(int);
  if (perm <= 160 * 1024) {
    const size_t with_keys = perm * 4;
    const int keys_in_lds = with_keys <= 160 * 1024;
    const size_t lds = keys_in_lds ? with_keys : perm;
    const int sweeps = (n + 1) / 2;
    hipStream_t st = (hipStream_t)stream;
    // as many pipeline stages as there are sweeps, up to the 1024 threads of a workgroup
    if (sweeps <= 64) return launch_ranking<64>(I1, I2, u, pc, n, rank, lds, keys_in_lds, seed, pc_dev, st);
    if (sweeps <= 256) return launch_ranking<256>(I1, I2, u, pc, n, rank, lds, keys_in_lds, seed, pc_dev, st);
    return launch_ranking<1024>(I1, I2, u, pc, n, rank, lds, keys_in_lds, seed, pc_dev, st);
  }
  if (scratch == nullptr) return (int)hipErrorInvalidValue;
  stochastic_ranking_serial_kernel<<<1, 64, 0, (hipStream_t)stream>>>(I1, I2, u, pc, n, rank, scratch, seed, pc_dev);
  return last_launch_status();
}

EVOX_API int evox_hip_stochastic_ranking(const float* I1, const float* I2, const float* u, float pc, int n,
                                         int64_t* rank, int* scratch, void* stream) {
  if (u == nullptr && n > 1) return (int)hipErrorInvalidValue;
  return stochastic_ranking_any(I1, I2, u, pc, n, rank, scratch, nullptr, nullptr, stream);
}

EVOX_API int evox_hip_stochastic_ranking_fresh(const float* I1, const float* I2, const int32_t* seed, float pc, int n,
                                               int64_t* rank, int* scratch, void* stream) {
  if (seed == nullptr) return (int)hipErrorInvalidValue;
  return stochastic_ranking_any(I1, I2, nullptr, pc, n, rank, scratch, seed, nullptr, stream);
}

// The same with the probability read from the device (one float): nothing of the call depends on a host value.
EVOX_API int evox_hip_stochastic_ranking_fresh_dev(const float* I1, const float* I2, const int32_t* seed, const float* pc,
                                                   int n, int64_t* rank, int* scratch, void* stream) {
  if (seed == nullptr || pc == nullptr) return (int)hipErrorInvalidValue;
  return stochastic_ranking_any(I1, I2, nullptr, 0.5f, n, rank, scratch, seed, pc, stream);
}

// The stochastic ranking on packed entries (see stochastic_ranking_packed_kernel): packed[i] = (r1[i] << 16) | r2[i],
// out = the packed words in their final order.  u / seed / pc as above (pc_dev optional).  4096 < n <= 24 576.
EVOX_API int evox_hip_stochastic_ranking_packed(const uint32_t* packed, const float* u, const int32_t* seed, float pc,
                                                const float* pc_dev, int n, uint32_t* out, void* stream) {
  if (n <= 1 || n > 1024 * 24 || packed == nullptr || out == nullptr || (u == nullptr && seed == nullptr))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 1024 * 8) return launch_ranking_packed<8>(packed, u, pc, n, out, seed, pc_dev, st);
  if (n <= 1024 * 16) return launch_ranking_packed<16>(packed, u, pc, n, out, seed, pc_dev, st);
  return launch_ranking_packed<24>(packed, u, pc, n, out, seed, pc_dev, st);
}

// The packed ranking over several workgroups (see stochastic_ranking_window_kernel): a chain of launches, each advancing
// every window by `halo` macro steps.  ws: evox_hip_stochastic_ranking_window_ws(n) words of scratch (any content).
EVOX_API int64_t evox_hip_stochastic_ranking_window_ws(int n) {
  const int threads = (n + 3) / 4;                       // the finer of the two block sizes
  return 2 * ((int64_t)threads * 4 + 4 + threads);
}

// block: entries per logical thread, 4 or 8; window: logical threads per workgroup, 256, 512 or 1024 (2 * halo < window).
EVOX_API int evox_hip_stochastic_ranking_window(const uint32_t* packed, const float* u, const int32_t* seed, float pc,
                                                const float* pc_dev, int n, uint32_t* out, uint32_t* ws, int halo, int block,
                                                int window, void* stream) {
  if (n <= 1 || n > 65535 || packed == nullptr || out == nullptr || ws == nullptr || (u == nullptr && seed == nullptr))
    return (int)hipErrorInvalidValue;
  if ((window != 256 && window != 512 && window != 1024) || halo < 2 || (halo & 1) || 2 * halo >= window ||
      (block != 4 && block != 8))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (window == 256) {
    if (block == 4) return launch_ranking_window<4, 256>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
    return launch_ranking_window<8, 256>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
  }
  if (window == 512) {
    if (block == 4) return launch_ranking_window<4, 512>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
    return launch_ranking_window<8, 512>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
  }
  if (block == 4) return launch_ranking_window<4, 1024>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
  return launch_ranking_window<8, 1024>(packed, u, pc, n, out, seed, pc_dev, ws, halo, st);
}

// MOEA/D replacement as a scatter: entry e = (row, slot) offers values[e] to sub-problem target[e]; a sub-problem takes
// the smallest offer below its current value, ties to the smallest row.  One 64-bit atomic minimum on (order-preserving
// value bits, row) per IMPROVING offer — most offers do not improve and issue nothing — instead of two atomic scatter
// passes over all N·T offers.  keys: n_sub words preset to all ones; finish: best value and winning row (-1: none).
namespace {

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long* __restrict__ p, int n, unsigned long long v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t b) {
  return __uint_as_float((b & 0x80000000u) ? (b & 0x7fffffffu) : ~b);
}

__global__ __launch_bounds__(256) void scatter_best_kernel(const float* __restrict__ values, const int64_t* __restrict__ target,
                                                           const float* __restrict__ current, int64_t entries, int slots,
                                                           int n_sub, unsigned long long* __restrict__ keys) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < entries; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[e];
    if (t < 0 || t >= n_sub) continue;
    const float v = values[e];
    if (v < current[t]) {
      const unsigned long long key = ((unsigned long long)ordered_bits(v) << 32) | (uint32_t)(e / slots);
      atomicMin(&keys[t], key);
    }
  }
}

__global__ __launch_bounds__(256) void scatter_best_finish_kernel(const unsigned long long* __restrict__ keys,
                                                                  const float* __restrict__ current, int n_sub,
                                                                  float* __restrict__ best, int64_t* __restrict__ winner) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_sub) return;
  const unsigned long long k = keys[j];
  const bool none = k == ~0ull;
  best[j] = none ? current[j] : from_ordered_bits((uint32_t)(k >> 32));
  winner[j] = none ? -1 : (int64_t)(uint32_t)k;
}

}  // namespace

// values / target: entries = rows x slots (row-major); current, best: n_sub floats; winner: n_sub int64; keys: n_sub
// 64-bit words of scratch (set here).
EVOX_API int evox_hip_scatter_best(const float* values, const int64_t* target, const float* current, int64_t entries,
                                   int slots, int n_sub, float* best, int64_t* winner, unsigned long long* keys,
                                   void* stream) {
  if (n_sub <= 0) return 0;
  if (slots <= 0 || entries < 0 || entries / slots > 0xFFFFFFFEll) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  fill_u64_kernel<<<(n_sub + 255) / 256, 256, 0, st>>>(keys, n_sub, ~0ull);
  if (entries > 0)
    scatter_best_kernel<<<stream_grid(entries, 256), 256, 0, st>>>(values, target, current, entries, slots, n_sub, keys);
  scatter_best_finish_kernel<<<(n_sub + 255) / 256, 256, 0, st>>>(keys, current, n_sub, best, winner);
  return last_launch_status();
}

EVOX_API int evox_hip_niche_fill(const int64_t* rho, const int64_t* cnt, int nref, const int64_t* k, int64_t* out,
                                 void* stream) {
  if (nref <= 0) return 0;
  niche_fill_kernel<<<1, NT, 0, (hipStream_t)stream>>>(rho, cnt, nref, k, out);
  return last_launch_status();
}
