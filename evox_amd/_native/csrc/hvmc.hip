// Monte-Carlo hypervolume credit (HypE).  S samples are drawn in the box [f_min, ref]; a sample
// dominated by c points credits each of them alpha[c−1].  Two scans, each lane holding its own vector in
// registers and walking the other set in lock-step with its wave (so the walked rows are wave-uniform
// and come through the scalar cache), replace the (n × S) boolean matrices of the tensor formulation:
//   count   per sample:  c_s = #{ i : p_i ≤ sample_s }                (points split over blockIdx.y, int atomics)
//   credit  per point :  credit_i = Σ_s [p_i ≤ sample_s] · alpha[c_s − 1]   (samples split over blockIdx.y, one
//                        partial row per slice — summed by the caller in a fixed order)
#include "hip_common.h"

using namespace evox;

namespace {

template <int M>
__global__ __launch_bounds__(256) void hv_count_kernel(const float* __restrict__ points, int n,
                                                       const float* __restrict__ samples, int S, int* __restrict__ cnt) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  float v[M];
#pragma unroll
  for (int k = 0; k < M; ++k) v[k] = s < S ? samples[(size_t)s * M + k] : -INFINITY;
  const int per = (n + gridDim.y - 1) / gridDim.y;
  const int lo = blockIdx.y * per, hi = min(n, lo + per);
  int c = 0;
#pragma unroll 4
  for (int q = lo; q < hi; ++q) {
    bool dom = true;
#pragma unroll
    for (int k = 0; k < M; ++k) dom = dom & (points[(size_t)q * M + k] <= v[k]);
    c += dom ? 1 : 0;
  }
  if (s < S && c) atomicAdd(&cnt[s], c);
}

template <int M>
__global__ __launch_bounds__(256) void hv_credit_kernel(const float* __restrict__ points, int n,
                                                        const float* __restrict__ samples, int S,
                                                        const int* __restrict__ cnt, const float* __restrict__ alpha,
                                                        float* __restrict__ partial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float p[M];
#pragma unroll
  for (int k = 0; k < M; ++k) p[k] = i < n ? points[(size_t)i * M + k] : INFINITY;
  const int per = (S + gridDim.y - 1) / gridDim.y;
  const int lo = blockIdx.y * per, hi = min(S, lo + per);
  float acc = 0.f;
#pragma unroll 4
  for (int q = lo; q < hi; ++q) {
    bool dom = true;
#pragma unroll
    for (int k = 0; k < M; ++k) dom = dom & (p[k] <= samples[(size_t)q * M + k]);
    const int c = cnt[q];
    const float a = c > 0 ? alpha[c - 1] : 0.f;
    acc += dom ? a : 0.f;
  }
  if (i < n) partial[(size_t)blockIdx.y * n + i] = acc;
}

template <int M>
int launch_hv(const float* points, int n, const float* samples, int S, int* cnt, const float* alpha, float* partial,
              int ysplit, hipStream_t s) {
  hv_count_kernel<M><<<dim3((S + 255) / 256, ysplit), 256, 0, s>>>(points, n, samples, S, cnt);
  hv_credit_kernel<M><<<dim3((n + 255) / 256, ysplit), 256, 0, s>>>(points, n, samples, S, cnt, alpha, partial);
  return last_launch_status();
}

}  // namespace

// points (n, m) and samples (S, m) row-major fp32; cnt: S ints zeroed by the caller; alpha: n floats;
// partial: ysplit × n floats (credit per slice of the samples)
EVOX_API int evox_hip_hv_credit(const float* points, int n, const float* samples, int S, int m, int* cnt,
                                const float* alpha, float* partial, int ysplit, void* stream) {
  if (n <= 0 || S <= 0 || ysplit < 1) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  switch (m) {
    case 2: return launch_hv<2>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 3: return launch_hv<3>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 4: return launch_hv<4>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 5: return launch_hv<5>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 6: return launch_hv<6>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 7: return launch_hv<7>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    case 8: return launch_hv<8>(points, n, samples, S, cnt, alpha, partial, ysplit, s);
    default: return (int)hipErrorInvalidValue;
  }
}
