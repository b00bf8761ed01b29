// Pairwise distances by direct differences: out[i][j] = ‖a_i − b_j‖ (L2, squared L2, L1 or L∞).
//
// The GEMM form (‖a‖² + ‖b‖² − 2 a·b) loses the small distances to cancellation — an absolute error of
// 1e-7 in d² is 3e-4 in d near zero — and nearest-neighbour selections (SPEA2 truncation, niche radii,
// IGD) live exactly there.  The work is bound by writing the p × r result, so the exact form costs
// nothing extra: a workgroup owns 32 rows × 256 columns, each thread keeps its column's vector in
// registers (short vectors) or streams both operands through LDS in chunks of 32 components.
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int kRows = 32;

// mode 0: Euclidean, 1: squared Euclidean, 2: Manhattan, 3: Chebyshev (uniform per launch)
__device__ __forceinline__ float accumulate(float acc, float d, int mode) {
  return mode == 2 ? acc + fabsf(d) : (mode == 3 ? fmaxf(acc, fabsf(d)) : acc + d * d);
}

template <int M>
__global__ __launch_bounds__(256) void cdist_small_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          int p, int r, float* __restrict__ out, int mode) {
  __shared__ float at[kRows][M];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * kRows;
  for (int e = threadIdx.x; e < kRows * M; e += 256) {
    const int i = i0 + e / M;
    at[e / M][e % M] = i < p ? a[(int64_t)i * M + e % M] : 0.f;
  }
  float bj[M];
#pragma unroll
  for (int k = 0; k < M; ++k) bj[k] = j < r ? b[(int64_t)j * M + k] : 0.f;
  __syncthreads();
  if (j >= r) return;
  const int rows = min(kRows, p - i0);
  for (int i = 0; i < rows; ++i) {
    float acc;
    if (mode <= 1) {
      // (squared) Euclidean: the fixed rounding sequence that the streaming nearest-neighbour kernels share
      float ai[M];
#pragma unroll
      for (int k = 0; k < M; ++k) ai[k] = at[i][k];
      acc = squared_distance<M>(ai, bj);
    } else {
      acc = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) acc = accumulate(acc, at[i][k] - bj[k], mode);
    }
    out[(int64_t)(i0 + i) * r + j] = mode == 0 ? sqrtf(acc) : acc;
  }
}

__global__ __launch_bounds__(256) void cdist_kernel(const float* __restrict__ a, const float* __restrict__ b, int p, int r,
                                                    int m, float* __restrict__ out, int mode) {
  __shared__ float at[kRows][32];
  __shared__ float bt[32][257];                         // component-major, padded: conflict-free both ways
  const int j0 = blockIdx.x * 256, i0 = blockIdx.y * kRows;
  float acc[kRows];
#pragma unroll
  for (int i = 0; i < kRows; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < m; k0 += 32) {
    const int kc = min(32, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < kRows * 32; e += 256) {
      const int i = i0 + (e >> 5), k = e & 31;
      at[e >> 5][k] = (i < p && k < kc) ? a[(int64_t)i * m + k0 + k] : 0.f;
    }
    for (int e = threadIdx.x; e < 256 * 32; e += 256) {
      const int c = e >> 5, k = e & 31;                 // consecutive threads read consecutive components of a row
      bt[k][c] = (j0 + c < r && k < kc) ? b[(int64_t)(j0 + c) * m + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int k = 0; k < kc; ++k) {
      const float bv = bt[k][threadIdx.x];
#pragma unroll
      for (int i = 0; i < kRows; ++i) acc[i] = accumulate(acc[i], at[i][k] - bv, mode);
    }
  }
  const int j = j0 + threadIdx.x;
  if (j >= r) return;
#pragma unroll
  for (int i = 0; i < kRows; ++i)
    if (i0 + i < p) out[(int64_t)(i0 + i) * r + j] = mode == 0 ? sqrtf(acc[i]) : acc[i];
}

template <int M>
int launch_small(const float* a, const float* b, int p, int r, float* out, int mode, hipStream_t st) {
  cdist_small_kernel<M><<<dim3((r + 255) / 256, (p + kRows - 1) / kRows), 256, 0, st>>>(a, b, p, r, out, mode);
  return last_launch_status();
}

}  // namespace

// a: (p, m), b: (r, m), out: (p, r); all row-major fp32.  mode: 0 L2, 1 squared L2, 2 L1, 3 L∞.  p / 32 must fit the grid's y extent (p ≤ 2 097 120).
EVOX_API int evox_hip_cdist(const float* a, const float* b, int p, int r, int m, float* out, int mode, void* stream) {
  if (p <= 0 || r <= 0) return 0;
  if (m <= 0 || (p + kRows - 1) / kRows > 65535) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (m) {
    case 1: return launch_small<1>(a, b, p, r, out, mode, st);
    case 2: return launch_small<2>(a, b, p, r, out, mode, st);
    case 3: return launch_small<3>(a, b, p, r, out, mode, st);
    case 4: return launch_small<4>(a, b, p, r, out, mode, st);
    case 5: return launch_small<5>(a, b, p, r, out, mode, st);
    case 6: return launch_small<6>(a, b, p, r, out, mode, st);
    case 7: return launch_small<7>(a, b, p, r, out, mode, st);
    case 8: return launch_small<8>(a, b, p, r, out, mode, st);
    default: break;
  }
  cdist_kernel<<<dim3((r + 255) / 256, (p + kRows - 1) / kRows), 256, 0, st>>>(a, b, p, r, m, out, mode);
  return last_launch_status();
}
