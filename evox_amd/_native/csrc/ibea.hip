// IBEA's ε-indicator fitness and its removal loop straight from the objective vectors.
//
// I(i, j) = max_k (f_i[k] − f_j[k]) is a three-flop function of two short vectors, so neither the
// fitness  F_j = 1 − Σ_i exp(−I(i, j) / (C_j κ)),  C_j = max_i |I(i, j)|,  nor the removal loop needs the
// n × n indicator matrix (nor the n × n × m difference tensor behind it): two streaming passes compute
// C and F, and the removal loop keeps objectives, C and F of the whole merged population in the registers
// of one workgroup, so one removal is an arg-min and one register update between workgroup barriers,
// with no global-memory latency on its critical path.
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int NT = 1024;

template <int M>
__device__ __forceinline__ float eps_ind(const float (&a)[M], const float (&b)[M]) {
  float v = a[0] - b[0];
#pragma unroll
  for (int k = 1; k < M; ++k) v = fmaxf(v, a[k] - b[k]);
  return v;
}

// 256 threads = 64 columns j × 4 slices of the rows i; PASS 0: C_j, PASS 1: F_j (reads C).
template <int M, int PASS>
__global__ __launch_bounds__(256) void ibea_fitness_kernel(const float* __restrict__ obj, int n, float kappa,
                                                           float* __restrict__ C, float* __restrict__ fit) {
  __shared__ float tile[M][256];
  __shared__ float part[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + r;
  float mine[M];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = j < n ? obj[(int64_t)j * M + k] : 0.f;
  const float cj = (PASS == 1 && j < n) ? C[j] : 1.f;
  const float sj = -1.0f / (cj * kappa);                  // the scale of the removal loop's increments: one product per pair
  float acc = 0.f;
  for (int i0 = 0; i0 < n; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? obj[(int64_t)ii * M + k] : 0.f;
    __syncthreads();
    const int lim = min(64, n - i0 - s * 64);             // rows of this slice that exist
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      float other[M];
#pragma unroll
      for (int k = 0; k < M; ++k) other[k] = tile[k][c];
      const float v = eps_ind<M>(other, mine);
      if (PASS == 0) acc = fmaxf(acc, fabsf(v));
      else acc += __expf(v * sj);
    }
  }
  part[s][r] = acc;
  __syncthreads();
  if (s == 0 && j < n) {
    if (PASS == 0) C[j] = fmaxf(fmaxf(acc, part[1][r]), fmaxf(part[2][r], part[3][r]));
    else fit[j] = 1.0f - (acc + part[1][r] + part[2][r] + part[3][r]);
  }
}

struct ArgMin { float v; int i; };

__device__ __forceinline__ ArgMin block_argmin(float v, int i, float* s_v, int* s_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
  __syncthreads();
  // every wave folds the 16 partial results with four shuffle steps
  float rv = s_v[lane & 15];
  int ri = s_i[lane & 15];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float ov = __shfl_xor(rv, o, 64);
    const int oi = __shfl_xor(ri, o, 64);
    if (ov < rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
  }
  return ArgMin{rv, ri};
}

// The same loop with every thread's share of the population (E members: fitness, scale and objectives)
// held in registers; the thread that owns the removed member publishes its objectives and scale through
// a few words of LDS.  One workgroup is bound by instruction issue of its single compute unit, so the
// loop body is kept to the E indicator evaluations (hardware exp) and a running arg-min.
template <int M, int E>
__global__ __launch_bounds__(NT) void ibea_remove_reg_kernel(const float* __restrict__ obj, int n, const float* __restrict__ C,
                                                             float kappa, const float* __restrict__ fit_in, int n_remove,
                                                             uint8_t* __restrict__ alive_out) {
  __shared__ float s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  __shared__ float s_x[M + 1];
  float f[E], cj[E], oj[E][M];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = threadIdx.x + e * NT;
    f[e] = INFINITY; cj[e] = 1.f;
#pragma unroll
    for (int k = 0; k < M; ++k) oj[e][k] = 0.f;
    if (j < n) {
      f[e] = fit_in[j]; cj[e] = C[j];
#pragma unroll
      for (int k = 0; k < M; ++k) oj[e][k] = obj[(int64_t)j * M + k];
    }
  }
  for (int r = 0; r < n_remove; ++r) {
    float v = INFINITY; int idx = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (f[e] < v) { v = f[e]; idx = threadIdx.x + e * NT; }
    const ArgMin m = block_argmin(v, idx, s_v, s_i);
    if (m.i == 0x7fffffff) break;                       // uniform: nothing left
    const int x = m.i;
    if ((int)threadIdx.x == (x & (NT - 1))) {           // the owner of x
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (e == x / NT) {
#pragma unroll
          for (int k = 0; k < M; ++k) s_x[k] = oj[e][k];
          s_x[M] = -1.0f / (cj[e] * kappa);
          f[e] = INFINITY;
        }
    }
    __syncthreads();
    float ox[M];
#pragma unroll
    for (int k = 0; k < M; ++k) ox[k] = s_x[k];
    const float s = s_x[M];
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] += __expf(eps_ind<M>(ox, oj[e]) * s);      // +inf stays +inf
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = threadIdx.x + e * NT;
    if (j < n) alive_out[j] = f[e] < INFINITY ? 1 : 0;
  }
}

// The removal loop spread over several compute units.  One workgroup is bound by the instruction issue of its compute
// unit (about 12 operations per member and removal); here G workgroups of 256 threads each hold a slice of the members in
// registers.  Per removal every workgroup publishes its local arg-min as one 64-bit word
//     [ order-preserving fitness bits : 32 | removal number mod 4096 : 12 | member index : 20 ]
// in its slot (two sets of slots, alternating with the parity of the removal number: a workgroup can run at most one
// removal ahead of the slowest, so a slot is never overwritten before everyone has read it), waits until the G slots of
// this removal are there, and takes their minimum — the same member on every workgroup, value first, index second, as in
// the single-workgroup loop, so the survivors are identical.  The removed member's objectives and scale are read from
// the (immutable) input arrays.  The wait is bounded: after spin_limit polls a workgroup raises *error and leaves, and so
// do the others when their own waits run out.  Only every kGroupStride-th workgroup of the launch takes part: consecutive
// workgroups go to different dies of the package, the ones that work then share one die's L2.
constexpr int kGroupStride = 8;
constexpr int kMultiThreads = 256;
constexpr unsigned kIdxMask = 0xFFFFFu;

__device__ __forceinline__ unsigned ordered_bits32(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int M, int E>
__global__ __launch_bounds__(kMultiThreads) void ibea_remove_multi_kernel(const float* __restrict__ obj, int n,
                                                                          const float* __restrict__ C, float kappa,
                                                                          const float* __restrict__ fit_in, int n_remove,
                                                                          uint8_t* __restrict__ alive_out,
                                                                          unsigned long long* slots, int* error, int G,
                                                                          int spin_limit) {
  if (blockIdx.x % kGroupStride) return;
  const int g = blockIdx.x / kGroupStride;
  if (g >= G) return;
  __shared__ float s_v[kMultiThreads / 64];
  __shared__ int s_i[kMultiThreads / 64];
  __shared__ unsigned long long s_win;
  const int base = g * kMultiThreads * E;
  float f[E], cj[E], oj[E][M];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = base + threadIdx.x + e * kMultiThreads;
    f[e] = INFINITY; cj[e] = 1.f;
#pragma unroll
    for (int k = 0; k < M; ++k) oj[e][k] = 0.f;
    if (j < n) {
      f[e] = fit_in[j]; cj[e] = C[j];
#pragma unroll
      for (int k = 0; k < M; ++k) oj[e][k] = obj[(int64_t)j * M + k];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = 0; r < n_remove; ++r) {
    float v = INFINITY; int idx = (int)kIdxMask;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (f[e] < v) { v = f[e]; idx = base + threadIdx.x + e * kMultiThreads; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) { s_v[wave] = v; s_i[wave] = idx; }
    __syncthreads();
    const unsigned tag = (unsigned)r & 0xFFFu;
    unsigned long long* set = slots + (size_t)(r & 1) * G;
    if (wave == 0) {
      if (lane == 0) {
#pragma unroll
        for (int w = 1; w < kMultiThreads / 64; ++w)
          if (s_v[w] < v || (s_v[w] == v && s_i[w] < idx)) { v = s_v[w]; idx = s_i[w]; }
        const unsigned long long mine = ((unsigned long long)ordered_bits32(v) << 32) | ((unsigned long long)tag << 20) |
                                        (unsigned)idx;
        __hip_atomic_store(&set[g], mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      // lane q < G waits for workgroup q's word of this removal
      unsigned long long word = ~0ull;
      int late = 0;
      if (lane < G) {
        int polls = 0;
        for (;;) {
          word = __hip_atomic_load(&set[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if ((((unsigned)(word >> 20)) & 0xFFFu) == tag) break;
          if (++polls >= spin_limit) { late = 1; word = ~0ull; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      unsigned hi = (unsigned)(word >> 32), lo = (unsigned)word;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned ohi = __shfl_xor(hi, o, 64), olo = __shfl_xor(lo, o, 64);
        if (ohi < hi || (ohi == hi && olo < lo)) { hi = ohi; lo = olo; }
      }
      const bool any_late = __any(late);
      if (lane == 0) {
        if (any_late) *error = 1;
        s_win = any_late ? ~0ull : (((unsigned long long)hi << 32) | lo);
      }
    }
    __syncthreads();
    const unsigned long long win = s_win;
    const int x = (int)((unsigned)win & kIdxMask);
    if (win == ~0ull || x == (int)kIdxMask) break;        // a wait ran out, or nothing is left: uniform over the launch
    float ox[M];
#pragma unroll
    for (int k = 0; k < M; ++k) ox[k] = obj[(int64_t)x * M + k];
    const float s = -1.0f / (C[x] * kappa);
    const int local = x - base;
    if (local >= 0 && local < kMultiThreads * E && (local & (kMultiThreads - 1)) == (int)threadIdx.x) {
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (e == local / kMultiThreads) f[e] = INFINITY;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) f[e] += __expf(eps_ind<M>(ox, oj[e]) * s);      // +inf stays +inf
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = base + threadIdx.x + e * kMultiThreads;
    if (j < n) alive_out[j] = f[e] < INFINITY ? 1 : 0;
  }
}

__global__ __launch_bounds__(64) void ibea_multi_reset_kernel(unsigned long long* slots, int count, int* error) {
  const int i = threadIdx.x;
  for (int k = i; k < count; k += 64) slots[k] = ~0ull;
  if (i == 0) *error = 0;
}

// SRA's two indicators for point j (thread), streamed over all points i like the fitness above:
//   I1_j = 1 − Σ_i exp(−I(i, j) / 0.05)                                   (ε-indicator fitness, unscaled)
//   I2_j = min_i ‖f_j − max(f_i, f_j)‖ = min_i sqrt(Σ_k max(f_i[k] − f_j[k], 0)²)     (shift-based density)
// with i < j only (LOWER, the reference's form: an individual is compared with those in front of it in the merged
// array, the first one gets +inf) or over all i ≠ j.
template <int M, bool LOWER>
__global__ __launch_bounds__(256) void sra_indicators_kernel(const float* __restrict__ obj, int n, float* __restrict__ I1,
                                                             float* __restrict__ I2) {
  __shared__ float tile[M][256];
  __shared__ float part1[4][64], part2[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + r;
  float mine[M];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = j < n ? obj[(int64_t)j * M + k] : 0.f;
  float acc = 0.f, best = INFINITY;
  for (int i0 = 0; i0 < n; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? obj[(int64_t)ii * M + k] : 0.f;
    __syncthreads();
    const int lim = min(64, n - i0 - s * 64);
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      const int i = i0 + c;
      float eps = -INFINITY, d2 = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const float d = tile[k][c] - mine[k];
        eps = fmaxf(eps, d);
        const float p = fmaxf(d, 0.f);
        d2 = fmaf(p, p, d2);
      }
      acc += __expf(eps * -20.0f);
      if (LOWER ? i < j : i != j) best = fminf(best, d2);
    }
  }
  part1[s][r] = acc;
  part2[s][r] = best;
  __syncthreads();
  if (s == 0 && j < n) {
    I1[j] = 1.0f - (acc + part1[1][r] + part1[2][r] + part1[3][r]);
    I2[j] = sqrtf(fminf(fminf(best, part2[1][r]), fminf(part2[2][r], part2[3][r])));
  }
}

// BiGE's crowding degree from the normalised objective vectors, without the n x n distance and sharing matrices:
//   cd_j = sqrt( Σ_{i ≠ j, d_ij < r} ( 0.5 · (1 + [pr_j >= pr_i] + [pr_j > pr_i]) · (1 − d_ij / r) )² ),
// d_ij the Euclidean distance, pr the proximity (sum of the normalised objectives), r the niche radius (read from the
// device: it depends on the size of the front being ranked).
template <int M>
__global__ __launch_bounds__(256) void bige_crowding_kernel(const float* __restrict__ f, const float* __restrict__ pr, int n,
                                                            const float* __restrict__ radius, float* __restrict__ cd) {
  __shared__ float tile[M + 1][256];
  __shared__ float part[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + r;
  const float rad = *radius, inv_rad = 1.f / rad;
  float mine[M];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = j < n ? f[(int64_t)j * M + k] : 0.f;
  const float pj = j < n ? pr[j] : 0.f;
  float acc = 0.f;
  for (int i0 = 0; i0 < n; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? f[(int64_t)ii * M + k] : 0.f;
    tile[M][threadIdx.x] = ii < n ? pr[ii] : 0.f;
    __syncthreads();
    const int lim = min(64, n - i0 - s * 64);
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const float d = tile[k][c] - mine[k];
        d2 = fmaf(d, d, d2);
      }
      const float d = sqrtf(d2);
      if (d < rad && i0 + c != j) {
        const float pi = tile[M][c];
        const float w = 0.5f * (1.f + (float)(pj >= pi) + (float)(pj > pi)) * (1.f - d * inv_rad);
        acc = fmaf(w, w, acc);
      }
    }
  }
  part[s][r] = acc;
  __syncthreads();
  if (s == 0 && j < n) cd[j] = sqrtf(fmaxf(acc + part[1][r] + part[2][r] + part[3][r], 0.f));
}

// KnEA's density estimate needs the K nearest neighbours of every point: K smallest distances (self excluded) streamed
// from the objective vectors; four lanes share a point, each keeps the K smallest of its quarter of the tiles in
// registers, lane group 0 merges.  out: n x K ascending distances.
template <int M, int K>
__global__ __launch_bounds__(256) void knn_dist_kernel(const float* __restrict__ f, int n, float* __restrict__ out) {
  __shared__ float tile[M][256];
  __shared__ float part[3][64][K];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + r;
  float mine[M], best[K];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = j < n ? f[(int64_t)j * M + k] : 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) best[k] = INFINITY;
  auto insert = [&](float d2) {
    if (d2 < best[K - 1]) {
      best[K - 1] = d2;
#pragma unroll
      for (int k = K - 1; k > 0; --k)
        if (best[k] < best[k - 1]) { const float t = best[k]; best[k] = best[k - 1]; best[k - 1] = t; }
    }
  };
  for (int i0 = 0; i0 < n; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? f[(int64_t)ii * M + k] : 0.f;
    __syncthreads();
    const int lim = min(64, n - i0 - s * 64);
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const float d = tile[k][c] - mine[k];
        d2 = fmaf(d, d, d2);
      }
      if (i0 + c != j) insert(d2);
    }
  }
  if (s > 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) part[s - 1][r][k] = best[k];
  }
  __syncthreads();
  if (s == 0 && j < n) {
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int k = 0; k < K; ++k) insert(part[w][r][k]);
#pragma unroll
    for (int k = 0; k < K; ++k) out[(int64_t)j * K + k] = sqrtf(best[k]);
  }
}

template <int M>
int launch_knn(const float* f, int n, int k, float* out, hipStream_t st) {
  const int blocks = (n + 63) / 64;
  switch (k) {
    case 1: knn_dist_kernel<M, 1><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 2: knn_dist_kernel<M, 2><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 3: knn_dist_kernel<M, 3><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 4: knn_dist_kernel<M, 4><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 5: knn_dist_kernel<M, 5><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 6: knn_dist_kernel<M, 6><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 7: knn_dist_kernel<M, 7><<<blocks, 256, 0, st>>>(f, n, out); break;
    case 8: knn_dist_kernel<M, 8><<<blocks, 256, 0, st>>>(f, n, out); break;
    default: return (int)hipErrorInvalidValue;
  }
  return last_launch_status();
}

// One round of KnEA's knee suppression as a fixed-point iteration.  The points are given in walking order (the point
// closest to the ideal side of the hyperplane first); a point stays a knee candidate unless an EARLIER point that is
// itself still selected lies in its neighbourhood (|f_p - f_q| < R in every objective):
//   keep_out[q] = not exists p < q : keep_in[p] and close(p, q).
// Iterated from all-ones this converges to the result of the sequential walk (position q is final after q rounds at
// the latest; in practice after a handful).  Tiles of earlier points only: half the pair tests.
template <int M>
__global__ __launch_bounds__(256) void knee_round_kernel(const float* __restrict__ f, const float* __restrict__ R, int n,
                                                         const uint8_t* __restrict__ keep_in, uint8_t* __restrict__ keep_out) {
  __shared__ float tile[M][256];
  __shared__ uint8_t tkeep[256];
  __shared__ int hit[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + r;
  float mine[M], rad[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    mine[k] = q < n ? f[(int64_t)q * M + k] : 0.f;
    rad[k] = R[k];
  }
  int found = 0;
  const int end = min(n, blockIdx.x * 64 + 64);            // points at or after this block's last row cannot precede it
  for (int i0 = 0; i0 < end; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? f[(int64_t)ii * M + k] : 0.f;
    tkeep[threadIdx.x] = ii < n ? keep_in[ii] : 0;
    __syncthreads();
    const int lim = min(64, end - i0 - s * 64);
    for (int t = 0; t < lim; ++t) {
      const int c = s * 64 + t;
      const int p = i0 + c;
      bool close = tkeep[c] && p < q;
#pragma unroll
      for (int k = 0; k < M; ++k) close = close && fabsf(tile[k][c] - mine[k]) < rad[k];
      found |= (int)close;
    }
  }
  hit[s][r] = found;
  __syncthreads();
  if (s == 0 && q < n) keep_out[q] = !(hit[0][r] | hit[1][r] | hit[2][r] | hit[3][r]);
}

// The same fixed point with three states per point — 0 undecided, 1 knee candidate for good, 2 suppressed for good — so
// that a round only looks at the undecided points (listed, ascending, in `todo`):
//   some earlier close point is 1            -> 2
//   no earlier close point is 0 or 1         -> 1      (every earlier close point is suppressed, or there is none)
//   otherwise                                -> stays 0.
// Both rules only use final states, so states never flip and the result is the sequential walk's whatever the number of
// rounds; the undecided set shrinks geometrically (a point is decided once the chain of close predecessors it hangs
// on is), and with it the work of a round.  state_out must hold a copy of state_in on entry.  suppress_only: apply the
// first rule alone (state_out may then be state_in itself) — afterwards no undecided point has a final-kept close
// predecessor, which is what the tail kernel below relies on.
template <int M>
__global__ __launch_bounds__(256) void knee_state_kernel(const float* __restrict__ f, const float* __restrict__ R, int n,
                                                         const int64_t* __restrict__ todo, int n_todo,
                                                         const uint8_t* state_in, uint8_t* state_out,   // one buffer when suppress_only
                                                         int suppress_only) {
  __shared__ float tile[M][256];
  __shared__ uint8_t tstate[256];
  __shared__ int hit[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + r;
  const int q = row < n_todo ? (int)todo[row] : -1;
  float mine[M], rad[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    mine[k] = q >= 0 ? f[(int64_t)q * M + k] : 0.f;
    rad[k] = R[k];
  }
  int found = 0;                                            // bit 0: a close earlier point is final-kept, bit 1: undecided
  const int last = min(n_todo, blockIdx.x * 64 + 64) - 1;
  const int end = (int)todo[last];                          // the largest point of this block: later points cannot precede
  for (int i0 = 0; i0 < end; i0 += 256) {
    __syncthreads();
    const int ii = i0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = ii < n ? f[(int64_t)ii * M + k] : 0.f;
    tstate[threadIdx.x] = ii < n ? state_in[ii] : 2;
    __syncthreads();
    const int lim = min(64, end - i0 - s * 64);
    for (int t = 0; t < lim; ++t) {
      const int c = s * 64 + t;
      const int p = i0 + c;
      const int st = tstate[c];
      bool close = st != 2 && p < q;
#pragma unroll
      for (int k = 0; k < M; ++k) close = close && fabsf(tile[k][c] - mine[k]) < rad[k];
      found |= close ? (st == 1 ? 1 : 2) : 0;
    }
  }
  hit[s][r] = found;
  __syncthreads();
  if (s == 0 && q >= 0) {
    const int all = hit[0][r] | hit[1][r] | hit[2][r] | hit[3][r];
    if (all & 1) state_out[q] = 2;
    else if (!(all & 2) && !suppress_only) state_out[q] = 1;
  }
}

// The tail of the three-state iteration in one workgroup: once at most 1024 points are undecided, their fate only depends
// on each other (a close earlier point that is final-kept would have decided them; final-suppressed ones do not count).
// Thread i takes the i-th undecided point, lists its close undecided predecessors once (up to kTailPred of them in LDS; a
// thread with more re-tests all its predecessors every round) and the rounds then run on LDS states without the host.
constexpr int kTail = 1024;
constexpr int kTailPred = 16;

template <int M>
__global__ __launch_bounds__(kTail) void knee_tail_kernel(const float* __restrict__ f, const float* __restrict__ R,
                                                          const int64_t* __restrict__ todo, int n_todo,
                                                          uint8_t* __restrict__ state) {
  __shared__ float s_obj[M][kTail];
  __shared__ uint16_t s_pred[kTailPred][kTail];
  __shared__ uint8_t s_state[kTail];
  const int i = threadIdx.x;
  const bool have = i < n_todo;
  const int q = have ? (int)todo[i] : 0;
  float mine[M], rad[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    mine[k] = have ? f[(int64_t)q * M + k] : 0.f;
    rad[k] = R[k];
    s_obj[k][i] = mine[k];
  }
  int mystate = have ? (int)state[q] : 2;                   // 0, or 2 from the suppress-only pass before this launch
  s_state[i] = (uint8_t)mystate;
  __syncthreads();
  int cnt = 0;
  if (have) {
    for (int j = 0; j < i; ++j) {                           // todo is ascending: the predecessors are the entries before i
      bool close = true;
#pragma unroll
      for (int k = 0; k < M; ++k) close = close && fabsf(s_obj[k][j] - mine[k]) < rad[k];
      if (close) {
        if (cnt < kTailPred) s_pred[cnt][i] = (uint16_t)j;
        ++cnt;
      }
    }
  }
  for (int round = 0; round < kTail; ++round) {            // a round decides at least the first undecided point
    int next = mystate;
    if (mystate == 0) {
      bool kept = false, open = false;
      if (cnt <= kTailPred) {
        for (int t = 0; t < cnt; ++t) {
          const int st = s_state[s_pred[t][i]];
          kept = kept || st == 1;
          open = open || st == 0;
        }
      } else {
        for (int j = 0; j < i; ++j) {
          const int st = s_state[j];
          if (st == 2) continue;
          bool close = true;
#pragma unroll
          for (int k = 0; k < M; ++k) close = close && fabsf(s_obj[k][j] - mine[k]) < rad[k];
          kept = kept || (close && st == 1);
          open = open || (close && st == 0);
        }
      }
      next = kept ? 2 : (open ? 0 : 1);
    }
    __syncthreads();                                        // everyone has read the old states
    s_state[i] = (uint8_t)next;
    mystate = next;
    if (!__syncthreads_or(mystate == 0)) break;
  }
  if (have) state[q] = (uint8_t)mystate;
}

template <int M>
int launch_knee_tail(const float* f, const float* R, const int64_t* todo, int n_todo, uint8_t* state, hipStream_t st) {
  knee_tail_kernel<M><<<1, kTail, 0, st>>>(f, R, todo, n_todo, state);
  return last_launch_status();
}

template <int M>
int launch_knee_state(const float* f, const float* R, int n, const int64_t* todo, int n_todo, const uint8_t* state_in,
                      uint8_t* state_out, int suppress_only, hipStream_t st) {
  knee_state_kernel<M><<<(n_todo + 63) / 64, 256, 0, st>>>(f, R, n, todo, n_todo, state_in, state_out, suppress_only);
  return last_launch_status();
}

template <int M>
int launch_knee(const float* f, const float* R, int n, const uint8_t* keep_in, uint8_t* keep_out, hipStream_t st) {
  knee_round_kernel<M><<<(n + 63) / 64, 256, 0, st>>>(f, R, n, keep_in, keep_out);
  return last_launch_status();
}

template <int M>
int launch_bige(const float* f, const float* pr, int n, const float* radius, float* cd, hipStream_t st) {
  bige_crowding_kernel<M><<<(n + 63) / 64, 256, 0, st>>>(f, pr, n, radius, cd);
  return last_launch_status();
}

template <int M>
int launch_sra(const float* obj, int n, int lower, float* I1, float* I2, hipStream_t st) {
  const int blocks = (n + 63) / 64;
  if (lower) sra_indicators_kernel<M, true><<<blocks, 256, 0, st>>>(obj, n, I1, I2);
  else sra_indicators_kernel<M, false><<<blocks, 256, 0, st>>>(obj, n, I1, I2);
  return last_launch_status();
}

template <int M>
int launch_fitness(const float* obj, int n, float kappa, float* C, float* fit, hipStream_t st) {
  const int blocks = (n + 63) / 64;
  ibea_fitness_kernel<M, 0><<<blocks, 256, 0, st>>>(obj, n, kappa, C, fit);
  ibea_fitness_kernel<M, 1><<<blocks, 256, 0, st>>>(obj, n, kappa, C, fit);
  return last_launch_status();
}

template <int M, int E>
int launch_remove_reg(const float* obj, int n, const float* C, float kappa, const float* fit, int n_remove, uint8_t* alive,
                      hipStream_t st) {
  ibea_remove_reg_kernel<M, E><<<1, NT, 0, st>>>(obj, n, C, kappa, fit, n_remove, alive);
  return last_launch_status();
}

template <int M, int E>
int launch_remove_multi(const float* obj, int n, const float* C, float kappa, const float* fit, int n_remove, uint8_t* alive,
                        unsigned long long* slots, int* error, int G, int spin_limit, hipStream_t st) {
  ibea_multi_reset_kernel<<<1, 64, 0, st>>>(slots, 2 * G, error);
  ibea_remove_multi_kernel<M, E><<<G * kGroupStride, kMultiThreads, 0, st>>>(obj, n, C, kappa, fit, n_remove, alive, slots,
                                                                           error, G, spin_limit);
  return last_launch_status();
}

constexpr int kMultiGroups = 8;
// members per thread of the multi-workgroup form
constexpr int multi_capacity(int m) { return m <= 4 ? 12 : 6; }

template <int M>
int launch_remove_multi_any(const float* obj, int n, const float* C, float kappa, const float* fit, int n_remove,
                            uint8_t* alive, unsigned long long* slots, int* error, int spin_limit, hipStream_t st) {
  const int per = kMultiGroups * kMultiThreads;
  const int e = (n + per - 1) / per;
  if (e <= 3) return launch_remove_multi<M, 3>(obj, n, C, kappa, fit, n_remove, alive, slots, error, kMultiGroups, spin_limit, st);
  if (e <= 6) return launch_remove_multi<M, 6>(obj, n, C, kappa, fit, n_remove, alive, slots, error, kMultiGroups, spin_limit, st);
  if constexpr (M <= 4) {
    if (e <= 12)
      return launch_remove_multi<M, 12>(obj, n, C, kappa, fit, n_remove, alive, slots, error, kMultiGroups, spin_limit, st);
  }
  return (int)hipErrorInvalidValue;
}

// members per thread that the register form is built for, given m (0: none — use the matrix form)
constexpr int reg_capacity(int m) { return m <= 3 ? 20 : (m <= 4 ? 16 : 8); }

template <int M>
int launch_remove(const float* obj, int n, const float* C, float kappa, const float* fit, int n_remove, uint8_t* alive,
                  hipStream_t st) {
  if (n <= 4 * NT) return launch_remove_reg<M, 4>(obj, n, C, kappa, fit, n_remove, alive, st);
  if (n <= 8 * NT) return launch_remove_reg<M, 8>(obj, n, C, kappa, fit, n_remove, alive, st);
  if constexpr (M <= 4) {
    if (n <= 16 * NT) return launch_remove_reg<M, 16>(obj, n, C, kappa, fit, n_remove, alive, st);
  }
  if constexpr (M <= 3) {
    if (n <= 20 * NT) return launch_remove_reg<M, 20>(obj, n, C, kappa, fit, n_remove, alive, st);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace

#define EVOX_IBEA_DISPATCH(FN, ...)                \
  switch (m) {                                     \
    case 2: return FN<2>(__VA_ARGS__);             \
    case 3: return FN<3>(__VA_ARGS__);             \
    case 4: return FN<4>(__VA_ARGS__);             \
    case 5: return FN<5>(__VA_ARGS__);             \
    case 6: return FN<6>(__VA_ARGS__);             \
    case 7: return FN<7>(__VA_ARGS__);             \
    case 8: return FN<8>(__VA_ARGS__);             \
    default: return (int)hipErrorInvalidValue;     \
  }

// obj: (n, m) row-major fp32 (already normalised), 2 ≤ m ≤ 8.  C, fit: out (n).
EVOX_API int evox_hip_ibea_fitness(const float* obj, int n, int m, float kappa, float* C, float* fit, void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_fitness, obj, n, kappa, C, fit, (hipStream_t)stream)
}

// obj: (n, m) row-major fp32.  I1, I2: out (n).  lower != 0: I2 over the points in front of j only.
EVOX_API int evox_hip_sra_indicators(const float* obj, int n, int m, int lower, float* I1, float* I2, void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_sra, obj, n, lower, I1, I2, (hipStream_t)stream)
}

// f: (n, m) normalised objectives, pr: (n) proximity, radius: one float on the device.  cd: out (n).
EVOX_API int evox_hip_bige_crowding(const float* f, const float* pr, int n, int m, const float* radius, float* cd, void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_bige, f, pr, n, radius, cd, (hipStream_t)stream)
}

// f: (n, m); out: (n, k) ascending Euclidean distances to the k nearest other points, 1 <= k <= 8.
EVOX_API int evox_hip_knn_dist(const float* f, int n, int m, int k, float* out, void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_knn, f, n, k, out, (hipStream_t)stream)
}

// f: (n, m) in walking order, R: (m) neighbourhood half-widths on the device, keep_in / keep_out: (n) bytes.
EVOX_API int evox_hip_knee_round(const float* f, const float* R, int n, int m, const uint8_t* keep_in, uint8_t* keep_out,
                                 void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_knee, f, R, n, keep_in, keep_out, (hipStream_t)stream)
}

// One round of the three-state form: todo (n_todo, ascending int64) lists the undecided points; state_out holds a copy of
// state_in on entry.
EVOX_API int evox_hip_knee_state_round(const float* f, const float* R, int n, int m, const int64_t* todo, int n_todo,
                                       const uint8_t* state_in, uint8_t* state_out, int suppress_only, void* stream) {
  if (n <= 0 || n_todo <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_knee_state, f, R, n, todo, n_todo, state_in, state_out, suppress_only, (hipStream_t)stream)
}

// The rest of the three-state iteration in one launch: todo (n_todo <= 1024, ascending) lists the undecided points, whose
// final states are written to `state`.  No listed point may have a final-kept close predecessor any more (run a
// suppress-only round over the list first); listed points that round suppressed are skipped.
EVOX_API int evox_hip_knee_state_tail(const float* f, const float* R, int m, const int64_t* todo, int n_todo, uint8_t* state,
                                      void* stream) {
  if (n_todo <= 0) return 0;
  if (n_todo > kTail) return (int)hipErrorInvalidValue;
  EVOX_IBEA_DISPATCH(launch_knee_tail, f, R, todo, n_todo, state, (hipStream_t)stream)
}

// Largest population the register-resident removal loop takes for m objectives (the caller uses the matrix
// form beyond it).
EVOX_API int64_t evox_hip_ibea_remove_obj_max(int m) { return m >= 2 && m <= 8 ? (int64_t)reg_capacity(m) * NT : 0; }

// Largest population of the multi-workgroup removal loop for m objectives (member indices take 20 bits).
EVOX_API int64_t evox_hip_ibea_remove_multi_max(int m) {
  return m >= 2 && m <= 8 ? (int64_t)multi_capacity(m) * kMultiGroups * kMultiThreads : 0;
}

// The removal loop on several compute units (see ibea_remove_multi_kernel).  slots: 2 x 8 64-bit words of scratch,
// error: one int, set to 1 when a bounded wait ran out (the survivor mask is then not valid); both are reset here.
EVOX_API int evox_hip_ibea_remove_obj_multi(const float* obj, int n, int m, const float* C, float kappa, const float* fit,
                                            int n_remove, uint8_t* alive, unsigned long long* slots, int* error,
                                            int spin_limit, void* stream) {
  if (n <= 0) return 0;
  if (n >= (int)kIdxMask || spin_limit <= 0 || slots == nullptr || error == nullptr) return (int)hipErrorInvalidValue;
  EVOX_IBEA_DISPATCH(launch_remove_multi_any, obj, n, C, kappa, fit, n_remove, alive, slots, error, spin_limit,
                     (hipStream_t)stream)
}

EVOX_API int evox_hip_ibea_remove_obj(const float* obj, int n, int m, const float* C, float kappa, const float* fit,
                                      int n_remove, uint8_t* alive, void* stream) {
  if (n <= 0) return 0;
  EVOX_IBEA_DISPATCH(launch_remove, obj, n, C, kappa, fit, n_remove, alive, (hipStream_t)stream)
}
