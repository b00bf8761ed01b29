// SPEA2 fitness without n × n intermediates.
//
// The fitness of a point is (sum of the strengths of its dominators) + 1 / (distance to its 2nd nearest
// neighbour + 2), where a point's strength is the number of points it dominates.  Written with tensor
// operations that is a boolean n × n dominance matrix, an n × n distance matrix and a row-wise
// selection on it; here two passes stream tiles of the objective vectors through LDS instead and keep
// a count and the two smallest distances per row in registers.
//
// Layout of a workgroup: 256 threads = 64 rows × 4 column slices; wave s handles columns
// [64 s, 64 s + 64) of every 256-point tile, so the LDS reads of a wave are broadcasts.
#include "hip_common.h"

using namespace evox;

namespace {

template <int M>
__global__ __launch_bounds__(256) void spea2_density_kernel(const float* __restrict__ obj, int n,
                                                            float* __restrict__ strength, float* __restrict__ kth) {
  __shared__ float tile[M][256];
  __shared__ float p_cnt[4][64], p_d1[4][64], p_d2[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;
  float mine[M];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = i < n ? obj[(int64_t)i * M + k] : NAN;
  float cnt = 0.f, d1 = INFINITY, d2 = INFINITY;
  for (int j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    const int jj = j0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = jj < n ? obj[(int64_t)jj * M + k] : NAN;
    __syncthreads();
    for (int q = 0; q < 64; ++q) {
      const int c = s * 64 + q;
      bool le = true, lt = false;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const float o = tile[k][c];
        le = le && (mine[k] <= o);
        lt = lt || (mine[k] < o);
        const float diff = mine[k] - o;
        d += diff * diff;
      }
      cnt += (le && lt) ? 1.f : 0.f;
      if (j0 + c != i) {                                  // padding columns give NaN, which fails both tests
        if (d < d1) { d2 = d1; d1 = d; }
        else if (d < d2) d2 = d;
      }
    }
  }
  p_cnt[s][r] = cnt; p_d1[s][r] = d1; p_d2[s][r] = d2;
  __syncthreads();
  if (s == 0 && i < n) {
    for (int t = 1; t < 4; ++t) {
      cnt += p_cnt[t][r];
      const float a = p_d1[t][r], b = p_d2[t][r];         // a ≤ b
      if (a < d1) { d2 = fminf(d1, b); d1 = a; }
      else if (a < d2) d2 = a;
    }
    strength[i] = cnt;
    kth[i] = sqrtf(d2);
  }
}

template <int M>
__global__ __launch_bounds__(256) void spea2_raw_kernel(const float* __restrict__ obj, int n,
                                                        const float* __restrict__ strength,
                                                        const float* __restrict__ kth, float* __restrict__ fit) {
  __shared__ float tile[M + 1][256];
  __shared__ float p_raw[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;
  float mine[M];
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] = i < n ? obj[(int64_t)i * M + k] : NAN;
  float raw = 0.f;
  for (int j0 = 0; j0 < n; j0 += 256) {
    __syncthreads();
    const int jj = j0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = jj < n ? obj[(int64_t)jj * M + k] : NAN;
    tile[M][threadIdx.x] = jj < n ? strength[jj] : 0.f;
    __syncthreads();
    for (int q = 0; q < 64; ++q) {
      const int c = s * 64 + q;
      bool le = true, lt = false;                         // does column c dominate row i?
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const float o = tile[k][c];
        le = le && (o <= mine[k]);
        lt = lt || (o < mine[k]);
      }
      raw += (le && lt) ? tile[M][c] : 0.f;
    }
  }
  p_raw[s][r] = raw;
  __syncthreads();
  if (s == 0 && i < n) fit[i] = 1.0f / (kth[i] + 2.0f) + (raw + p_raw[1][r] + p_raw[2][r] + p_raw[3][r]);
}

template <int M>
int launch(const float* obj, int n, float* strength, float* kth, float* fit, hipStream_t stream) {
  const int blocks = (n + 63) / 64;
  spea2_density_kernel<M><<<blocks, 256, 0, stream>>>(obj, n, strength, kth);
  spea2_raw_kernel<M><<<blocks, 256, 0, stream>>>(obj, n, strength, kth, fit);
  return last_launch_status();
}

}  // namespace

// obj: (n, m) row-major fp32, 2 ≤ m ≤ 8.  strength, kth: scratch (n); fit: out (n).
EVOX_API int evox_hip_spea2_fitness(const float* obj, int n, int m, float* strength, float* kth, float* fit, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  switch (m) {
    case 2: return launch<2>(obj, n, strength, kth, fit, st);
    case 3: return launch<3>(obj, n, strength, kth, fit, st);
    case 4: return launch<4>(obj, n, strength, kth, fit, st);
    case 5: return launch<5>(obj, n, strength, kth, fit, st);
    case 6: return launch<6>(obj, n, strength, kth, fit, st);
    case 7: return launch<7>(obj, n, strength, kth, fit, st);
    case 8: return launch<8>(obj, n, strength, kth, fit, st);
    default: return (int)hipErrorInvalidValue;
  }
}
