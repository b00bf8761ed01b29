// Elementwise companions of the CMA-ES GEMMs.
//
//   cma_gather_scale   S[i, :] = (pop[order[i], :] - mean) * rowscale[i] / sigma      (i < mu)
//       With rowscale = sqrt(w) the rank-mu term of the covariance update is the Gram matrix S^T S and
//       the weighted mean shift is sigma * S^T sqrt(w): the selected rows are read once, centred,
//       scaled and written once (fp32 or bf16) instead of going through gather / subtract / multiply
//       passes with temporaries.
//   cma_affine         X = mean + sigma * G   (in place on the GEMM output, sigma read on the device)
#include "hip_common.h"

using namespace evox;

namespace {

__device__ __forceinline__ uint16_t to_bf16(float v) {
  const uint32_t u = __float_as_uint(v);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// colsum[b, c] = sum over the rows i handled by block b of rowscale[i]^2 * (x - mean)[c]  (the block's
// share of the weighted mean shift, added up by the caller in a fixed order)
template <typename T>
__global__ __launch_bounds__(256) void gather_scale_kernel(const float* __restrict__ pop, int64_t ldpop,
                                                           const int64_t* __restrict__ order, const float* __restrict__ mean,
                                                           const float* __restrict__ rowscale, const float* __restrict__ sigma,
                                                           T* __restrict__ out, float* __restrict__ colsum, int mu, int n) {
  const float inv_sigma = 1.0f / sigma[0];
  // rows on 16-byte boundaries: four columns per lane (one 16-byte load; a 16- or 8-byte store), two rows in flight
  auto store4 = [&](int64_t row, int c, float4 v) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>((float*)out + row * n + c) = v;
    } else {
      const uint32_t lo = (uint32_t)to_bf16(v.x) | ((uint32_t)to_bf16(v.y) << 16);
      const uint32_t hi = (uint32_t)to_bf16(v.z) | ((uint32_t)to_bf16(v.w) << 16);
      *reinterpret_cast<uint2*>((uint16_t*)out + row * n + c) = make_uint2(lo, hi);
    }
  };
  constexpr uintptr_t kOutAlign = sizeof(T) == 4 ? 15 : 7;
  if ((n & 3) == 0 && (ldpop & 3) == 0 && (((uintptr_t)pop | (uintptr_t)mean | (uintptr_t)colsum) & 15) == 0 &&
      ((uintptr_t)out & kOutAlign) == 0) {
    for (int c = threadIdx.x * 4; c < n; c += blockDim.x * 4) {
      const float4 m = *reinterpret_cast<const float4*>(mean + c);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int i = blockIdx.x;
      for (; i + (int)gridDim.x < mu; i += 2 * gridDim.x) {
        const int i2 = i + gridDim.x;
        const float r1 = rowscale[i], r2 = rowscale[i2];
        const float4 x1 = *reinterpret_cast<const float4*>(pop + order[i] * ldpop + c);
        const float4 x2 = *reinterpret_cast<const float4*>(pop + order[i2] * ldpop + c);
        const float4 d1 = make_float4(x1.x - m.x, x1.y - m.y, x1.z - m.z, x1.w - m.w);
        const float4 d2 = make_float4(x2.x - m.x, x2.y - m.y, x2.z - m.z, x2.w - m.w);
        const float s1 = r1 * inv_sigma, s2 = r2 * inv_sigma;
        store4(i, c, make_float4(d1.x * s1, d1.y * s1, d1.z * s1, d1.w * s1));
        store4(i2, c, make_float4(d2.x * s2, d2.y * s2, d2.z * s2, d2.w * s2));
        const float q1 = r1 * r1, q2 = r2 * r2;
        acc.x = fmaf(q1, d1.x, acc.x); acc.y = fmaf(q1, d1.y, acc.y); acc.z = fmaf(q1, d1.z, acc.z); acc.w = fmaf(q1, d1.w, acc.w);
        acc.x = fmaf(q2, d2.x, acc.x); acc.y = fmaf(q2, d2.y, acc.y); acc.z = fmaf(q2, d2.z, acc.z); acc.w = fmaf(q2, d2.w, acc.w);
      }
      for (; i < mu; i += gridDim.x) {
        const float r1 = rowscale[i];
        const float4 x1 = *reinterpret_cast<const float4*>(pop + order[i] * ldpop + c);
        const float4 d1 = make_float4(x1.x - m.x, x1.y - m.y, x1.z - m.z, x1.w - m.w);
        const float s1 = r1 * inv_sigma;
        store4(i, c, make_float4(d1.x * s1, d1.y * s1, d1.z * s1, d1.w * s1));
        const float q1 = r1 * r1;
        acc.x = fmaf(q1, d1.x, acc.x); acc.y = fmaf(q1, d1.y, acc.y); acc.z = fmaf(q1, d1.z, acc.z); acc.w = fmaf(q1, d1.w, acc.w);
      }
      *reinterpret_cast<float4*>(colsum + (int64_t)blockIdx.x * n + c) = acc;
    }
    return;
  }
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    const float m = mean[c];
    float acc = 0.f;
    for (int i = blockIdx.x; i < mu; i += gridDim.x) {
      const float r = rowscale[i];
      const float d = pop[order[i] * ldpop + c] - m;
      const float v = d * (r * inv_sigma);
      if constexpr (sizeof(T) == 2) out[(int64_t)i * n + c] = to_bf16(v); else out[(int64_t)i * n + c] = v;
      acc = fmaf(r * r, d, acc);
    }
    colsum[(int64_t)blockIdx.x * n + c] = acc;
  }
}

__global__ __launch_bounds__(256) void affine_kernel(float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ sigma, int64_t rows, int n) {
  const float s = sigma[0];
  const int64_t total = rows * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    x[e] = fmaf(s, x[e], mean[e % n]);
}

// All O(n) pieces of the CMA-ES path / step-size update in one single-workgroup launch
// (the matrix-vector product C^{-1/2} step is done before, by the library):
//   mean' = mean + cm step,  delta = cm step
//   ps'   = (1-cs) ps + sqrt(cs (2-cs) mueff) * (whitened * cm / sigma)
//   hsig  = |ps'| / sqrt(1 - (1-cs)^(2 t)) < (1.4 + 2/(n+1)) chiN
//   pc'   = (1-cc) pc + hsig sqrt(cc (2-cc) mueff) delta / sigma
//   sigma'= sigma exp((cs/damps) (|ps'|/chiN - 1)),   alpha = (1-c1-cmu) + c1 (1-hsig) cc (2-cc)
struct PathConst { float cs, cc, mueff, cm, damps, chiN, c1, cmu; };

__global__ __launch_bounds__(1024) void cma_paths_kernel(const float* __restrict__ mean, const float* __restrict__ step,
                                                         const float* __restrict__ whitened, const float* __restrict__ ps,
                                                         const float* __restrict__ pc, const float* __restrict__ sigma,
                                                         const float* __restrict__ iter, PathConst k, int n,
                                                         float* __restrict__ mean_out, float* __restrict__ ps_out,
                                                         float* __restrict__ pc_out, float* __restrict__ scalars) {
  __shared__ float red[16];
  __shared__ float s_norm;
  const float sg = sigma[0];
  const float a = sqrtf(k.cs * (2.f - k.cs) * k.mueff) * k.cm / sg;
  float sq = 0.f;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    const float p = (1.f - k.cs) * ps[c] + a * whitened[c];
    ps_out[c] = p;
    sq = fmaf(p, p, sq);
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    s_norm = sqrtf(t);
  }
  __syncthreads();
  const float norm = s_norm;
  const float denom = sqrtf(1.f - powf(1.f - k.cs, 2.f * iter[0]));
  const float hsig = (norm / denom < (1.4f + 2.f / (float)(n + 1)) * k.chiN) ? 1.f : 0.f;
  const float b = hsig * sqrtf(k.cc * (2.f - k.cc) * k.mueff) * k.cm / sg;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    const float st = step[c];
    mean_out[c] = fmaf(k.cm, st, mean[c]);
    pc_out[c] = (1.f - k.cc) * pc[c] + b * st;
  }
  const float sigma_new = sg * expf((k.cs / k.damps) * (norm / k.chiN - 1.f));
  // scalars[4 .. 4+n): sigma' repeated — the per-column scale vector the sampling GEMM's epilogue reads
  for (int c = threadIdx.x; c < n; c += blockDim.x) scalars[4 + c] = sigma_new;
  if (threadIdx.x == 0) {
    scalars[0] = sigma_new;                                                                   // sigma'
    scalars[1] = (1.f - k.c1 - k.cmu) + k.c1 * (1.f - hsig) * k.cc * (2.f - k.cc);            // alpha
    scalars[2] = hsig;
    scalars[3] = norm;
  }
}

}  // namespace

EVOX_API int evox_hip_cma_paths(const float* mean, const float* step, const float* whitened, const float* ps,
                                const float* pc, const float* sigma, const float* iter, const float* consts, int n,
                                float* mean_out, float* ps_out, float* pc_out, float* scalars, void* stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  PathConst k{consts[0], consts[1], consts[2], consts[3], consts[4], consts[5], consts[6], consts[7]};
  cma_paths_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(mean, step, whitened, ps, pc, sigma, iter, k, n, mean_out, ps_out,
                                                        pc_out, scalars);
  return last_launch_status();
}

// colsum: blocks x n partial sums; blocks is chosen by the caller (grid size).
EVOX_API int evox_hip_cma_gather_scale(const float* pop, int64_t ldpop, const int64_t* order, const float* mean,
                                       const float* rowscale, const float* sigma, void* out, int out_dtype, float* colsum,
                                       int blocks, int mu, int n, void* stream) {
  if (mu <= 0 || n <= 0 || blocks <= 0) return (int)hipErrorInvalidValue;
  if (out_dtype == 1)
    gather_scale_kernel<uint16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(pop, ldpop, order, mean, rowscale, sigma,
                                                                            (uint16_t*)out, colsum, mu, n);
  else
    gather_scale_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(pop, ldpop, order, mean, rowscale, sigma,
                                                                         (float*)out, colsum, mu, n);
  return last_launch_status();
}

namespace {
// base = alpha·C + c1·pc pcᵀ (alpha on the device: it depends on the stall indicator)
__global__ __launch_bounds__(256) void cov_base_kernel(const float* __restrict__ C, const float* __restrict__ pc,
                                                       const float* __restrict__ alpha, float c1,
                                                       float* __restrict__ out, int n) {
  const float a = alpha[0];
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    out[i] = a * C[i] + c1 * pc[r] * pc[c];
  }
}
}  // namespace

EVOX_API int evox_hip_cma_cov_base(const float* C, const float* pc, const float* alpha, float c1, float* out, int n,
                                   void* stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  cov_base_kernel<<<evox::stream_grid((int64_t)n * n, 256), 256, 0, (hipStream_t)stream>>>(C, pc, alpha, c1, out, n);
  return evox::last_launch_status();
}

EVOX_API int evox_hip_cma_affine(float* x, const float* mean, const float* sigma, int64_t rows, int n, void* stream) {
  if (rows <= 0 || n <= 0) return 0;
  affine_kernel<<<stream_grid(rows * n, 256), 256, 0, (hipStream_t)stream>>>(x, mean, sigma, rows, n);
  return last_launch_status();
}
