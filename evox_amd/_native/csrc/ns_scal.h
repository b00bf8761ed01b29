// Scalar block, per-block partial results and small helpers shared by the Newton-Schulz factorisation kernels
// (blas.hip, cma_fast.hip).
#pragma once
#include "hip_common.h"

namespace evox {
namespace nsk {

// scal: float[kScalFloats].  [13] scale s, [14] residual, [15] 1 if the guarded outputs were
// replaced; from [16] on, rows of 256 per-block partial results (no atomics: a thousand waves
// adding into one address serialise for tens of µs, while every later kernel can re-reduce
// 256 partials in its prologue for nothing):
//   rows 0..7 ‖w_j‖² of the power steps, row 8 max row-abs-sum, row 9 Σ diag, row 10 Σ (T−I)²
constexpr int kMaxPow = 8, kPartRow = 256, kRowInf = 8, kRowDiag = 9, kRowRes = 10;
constexpr int kScalFloats = 16 + 11 * kPartRow;
__device__ __forceinline__ float* part_row(float* scal, int k) { return scal + 16 + k * kPartRow; }
__device__ __forceinline__ const float* part_row(const float* scal, int k) { return scal + 16 + k * kPartRow; }

// every thread of a 256-thread block receives the sum / max of `count` (≤ 256) partials
__device__ __forceinline__ float block_reduce(const float* parts, int count, bool take_max, float* red) {
  float v = (int)threadIdx.x < count ? parts[threadIdx.x] : 0.f;
  v = take_max ? evox::wave_max(v) : evox::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return take_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}


// 1 + e/2 + 3e²/8 + 5e³/16 + 35e⁴/128 with e = I − M, in powers of M
constexpr float kP4c0 = 315.f / 128.f, kP4c1 = -420.f / 128.f, kP4c2 = 378.f / 128.f, kP4c3 = -180.f / 128.f, kP4c4 = 35.f / 128.f;

// fixed pseudo-random probe vector of the residual estimate
__device__ __forceinline__ float probe_v(int i) {
  const uint32_t h = (uint32_t)i * 2654435761u + 0x9E3779B9u;
  return (float)((h >> 9) & 0xFFFFu) * (1.f / 32768.f) - 1.f;
}

}  // namespace nsk
}  // namespace evox
