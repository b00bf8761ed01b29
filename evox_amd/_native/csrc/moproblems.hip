// DTLZ1–7 evaluated in one pass: a lane owns one individual, accumulates the distance function g
// over the last d−m+1 variables and unrolls the triangular product over the first m−1.
// (The tensor formulation is a dozen launches with a cumulative-product scan in the middle.)
#include "hip_common.h"

namespace {

constexpr int kMaxObj = 16;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kPi = 3.14159265358979323846f;

// kind: 1..7 = DTLZ1..DTLZ7
__global__ __launch_bounds__(256) void dtlz_kernel(const float* __restrict__ x, int64_t ldx, int n, int d, int m,
                                                   int kind, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (int64_t)i * ldx;
  const int k = d - m + 1;                       // distance variables x[m−1 .. d)
  float g = 0.f;
  if (kind == 1 || kind == 3) {
    for (int j = m - 1; j < d; ++j) {
      const float t = r[j] - 0.5f;
      g += t * t - cosf(20.f * kPi * t);
    }
    g = 100.f * ((float)k + g);
  } else if (kind == 6) {
    for (int j = m - 1; j < d; ++j) g += powf(r[j], 0.1f);
  } else if (kind == 7) {
    for (int j = m - 1; j < d; ++j) g += r[j];
    g = 1.f + 9.f * g / (float)k;
  } else {
    for (int j = m - 1; j < d; ++j) {
      const float t = r[j] - 0.5f;
      g += t * t;
    }
  }
  float* o = out + (int64_t)i * m;
  if (kind == 7) {
    float s = 0.f;
    for (int j = 0; j < m - 1; ++j) {
      const float h = r[j];
      o[j] = h;
      s += h / (1.f + g) * (1.f + sinf(3.f * kPi * h));
    }
    o[m - 1] = (1.f + g) * ((float)m - s);
    return;
  }
  // f_j = scale · Π_{i < m−1−j} head_i · tail_{m−1−j}  (no tail factor for j = 0)
  float head[kMaxObj], tail[kMaxObj];
  for (int j = 0; j < m - 1; ++j) {
    float p = r[j];
    if (kind == 1) {
      head[j] = p;
      tail[j] = 1.f - p;
    } else {
      if (kind == 4) p = powf(p, 100.f);
      if ((kind == 5 || kind == 6) && j > 0) p = (1.f + 2.f * g * p) / (2.f + 2.f * g);
      const float a = p * kHalfPi;
      head[j] = fmaxf(cosf(a), 0.f);
      tail[j] = sinf(a);
    }
  }
  const float scale = (kind == 1 ? 0.5f : 1.f) * (1.f + g);
  float prod = 1.f;
  // walk j from m−1 down to 0: prefix products grow one head factor at a time
  for (int q = 0; q < m; ++q) {
    const int j = m - 1 - q;                      // objective index; uses heads 0..q−1 and tail_q
    float f = prod;
    if (j > 0) f *= tail[q];
    o[j] = scale * f;
    if (q < m - 1) prod *= head[q];
  }
}

}  // namespace

EVOX_API int evox_hip_dtlz(const float* x, int64_t ldx, int n, int d, int m, int kind, float* out, void* stream) {
  if (n <= 0 || m < 2 || m > kMaxObj || d < m || kind < 1 || kind > 7) return (int)hipErrorInvalidValue;
  dtlz_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, ldx, n, d, m, kind, out);
  return evox::last_launch_status();
}

// ZDT1, 2, 3, 4, 6 (kind): f1 = f1(x_0), f2 = g(x_1..) · h(f1, g), one thread per row.
namespace {
__global__ __launch_bounds__(256) void zdt_kernel(const float* __restrict__ x, int64_t ldx, int n, int d, int kind,
                                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (int64_t)i * ldx;
  const float x0 = r[0];
  float acc = 0.f;
  if (kind == 4) {
    for (int j = 1; j < d; ++j) acc += r[j] * r[j] - 10.f * cosf(4.f * kPi * r[j]);
  } else {
    for (int j = 1; j < d; ++j) acc += r[j];
  }
  float f1 = x0, g;
  if (kind == 4) g = 1.f + 10.f * (float)(d - 1) + acc;
  else if (kind == 6) g = 1.f + 9.f * powf(acc / 9.f, 0.25f);
  else g = 1.f + 9.f * acc / (float)(d - 1);
  if (kind == 6) { const float sn = sinf(6.f * kPi * x0); const float s2 = sn * sn; f1 = 1.f - expf(-4.f * x0) * s2 * s2 * s2; }
  const float q = f1 / g;
  float h;
  if (kind == 2 || kind == 6) h = 1.f - q * q;
  else if (kind == 3) h = 1.f - sqrtf(q) - q * sinf(10.f * kPi * f1);
  else h = 1.f - sqrtf(q);
  out[2 * (int64_t)i] = f1;
  out[2 * (int64_t)i + 1] = g * h;
}
}  // namespace

EVOX_API int evox_hip_zdt(const float* x, int64_t ldx, int n, int d, int kind, float* out, void* stream) {
  if (n <= 0 || d < 2 || !(kind == 1 || kind == 2 || kind == 3 || kind == 4 || kind == 6)) return (int)hipErrorInvalidValue;
  zdt_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, ldx, n, d, kind, out);
  return evox::last_launch_status();
}

// CEC-style prologue z = M·((x − o)·rate), optionally with the output coordinates permuted
// (z'_j = z_{perm[j]}): three tensor ops and a K = d ≤ 128 GEMM that the library runs far below its
// rate, as one pass with the rotation matrix staged in LDS.
namespace {

constexpr int kMaxRotDim = 128;

__global__ __launch_bounds__(256) void shift_rotate_kernel(const float* __restrict__ x, int64_t ldx,
                                                           const float* __restrict__ o, const float* __restrict__ M,
                                                           const int64_t* __restrict__ perm, float rate, int n, int d,
                                                           float* __restrict__ out) {
  extern __shared__ float lds[];                // M (d × (d+1), padded rows) then o·rate pre-scaled? no: o (d)
  float* sm = lds;
  float* so = lds + d * (d + 1);
  for (int t = threadIdx.x; t < d * d; t += blockDim.x) sm[(t / d) * (d + 1) + (t % d)] = M[t];
  for (int t = threadIdx.x; t < d; t += blockDim.x) so[t] = o[t];
  __syncthreads();
  const int64_t total = (int64_t)n * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / d), j = (int)(e - (int64_t)r * d);
    const int row = perm ? (int)perm[j] : j;
    const float* xr = x + (int64_t)r * ldx;
    const float* mr = sm + row * (d + 1);
    float acc = 0.f;
    for (int k = 0; k < d; ++k) acc += (xr[k] - so[k]) * rate * mr[k];
    out[e] = acc;
  }
}

}  // namespace

EVOX_API int evox_hip_shift_rotate(const float* x, int64_t ldx, const float* o, const float* M, const int64_t* perm,
                                   float rate, int n, int d, float* out, void* stream) {
  if (n <= 0 || d <= 0 || d > kMaxRotDim) return (int)hipErrorInvalidValue;
  const size_t lds = sizeof(float) * ((size_t)d * (d + 1) + d);
  shift_rotate_kernel<<<evox::stream_grid((int64_t)n * d, 256), 256, lds, (hipStream_t)stream>>>(x, ldx, o, M, perm, rate,
                                                                                               n, d, out);
  return evox::last_launch_status();
}

// A x = b for one tiny dense system (m ≤ 16) by Gaussian elimination with partial pivoting, one lane.
// The library routes (lstsq / solve / matrix_rank) cost milliseconds of host time on a 3 × 3 system;
// NSGA-III solves one per generation for its hyperplane intercepts.  flag[0] = 1 when a pivot is
// negligible against the largest entry (the matrix is numerically rank deficient).
namespace {
__global__ void small_solve_kernel(const float* __restrict__ A, const float* __restrict__ b, int m, float* __restrict__ x,
                                   int* __restrict__ flag) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float a[16][17];
  float scale = 0.f;
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < m; ++j) { a[i][j] = A[i * m + j]; scale = fmaxf(scale, fabsf(a[i][j])); }
    a[i][m] = b[i];
  }
  int singular = !(scale > 0.f) || !(scale < INFINITY);
  const float tiny = scale * 1e-6f * (float)m;
  for (int c = 0; c < m && !singular; ++c) {
    int piv = c;
    for (int r = c + 1; r < m; ++r) if (fabsf(a[r][c]) > fabsf(a[piv][c])) piv = r;
    if (!(fabsf(a[piv][c]) > tiny)) { singular = 1; break; }
    if (piv != c) for (int j = c; j <= m; ++j) { const float t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    const float inv = 1.f / a[c][c];
    for (int r = c + 1; r < m; ++r) {
      const float f = a[r][c] * inv;
      for (int j = c; j <= m; ++j) a[r][j] -= f * a[c][j];
    }
  }
  if (!singular) {
    for (int r = m - 1; r >= 0; --r) {
      float s = a[r][m];
      for (int j = r + 1; j < m; ++j) s -= a[r][j] * a[j][m];
      a[r][m] = s / a[r][r];
    }
  }
  for (int i = 0; i < m; ++i) x[i] = singular ? NAN : a[i][m];
  flag[0] = singular;
}
}  // namespace

EVOX_API int evox_hip_small_solve(const float* A, const float* b, int m, float* x, int* flag, void* stream) {
  if (m < 1 || m > 16) return (int)hipErrorInvalidValue;
  small_solve_kernel<<<1, 64, 0, (hipStream_t)stream>>>(A, b, m, x, flag);
  return evox::last_launch_status();
}

// CEC2022 F1, F2, F4, F5 in one launch: shift, scale and rotate (z = M·((x − o)·rate)) and the basic function
// (Zakharov, Rosenbrock, Rastrigin, Levy) per row, rotation matrix and shift staged in LDS, the row and its image
// in registers (the suite is defined for D ∈ {2, 10, 20} only).  Values below `floor` are reported as zero.
namespace {

template <int D>
__global__ __launch_bounds__(256) void cec_simple_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ o,
                                                         const float* __restrict__ M, float rate, int kind, float floor_,
                                                         int n, float* __restrict__ out) {
  __shared__ float sm[D * D];
  __shared__ float so[D];
  for (int t = threadIdx.x; t < D * D; t += blockDim.x) sm[t] = M[t];
  for (int t = threadIdx.x; t < D; t += blockDim.x) so[t] = o[t];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (int64_t)i * ldx;
  float y[D], z[D];
#pragma unroll
  for (int k = 0; k < D; ++k) y[k] = (r[k] - so[k]) * rate;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) acc = fmaf(sm[j * D + k], y[k], acc);
    z[j] = acc;
  }
  float f;
  if (kind == 1) {                                      // Zakharov
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) { t1 = fmaf(z[j], z[j], t1); t2 = fmaf(0.5f * (float)(j + 1), z[j], t2); }
    const float q = t2 * t2;
    f = t1 + q + q * q;
  } else if (kind == 2) {                               // Rosenbrock (shifted to the optimum at 1)
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j + 1 < D; ++j) {
      const float u = z[j] + 1.f, v = z[j + 1] + 1.f;
      const float d = u * u - v;
      a = fmaf(d, d, a);
      b = fmaf(1.f - u, 1.f - u, b);
    }
    f = 100.f * a + b;
  } else if (kind == 4) {                               // Rastrigin
    f = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float v = z[j] * 0.0512f;
      f += v * v - 10.f * cospif(2.f * v) + 10.f;
    }
  } else {                                              // Levy
    const float w0 = 1.f + z[0] * 0.25f, wl = 1.f + z[D - 1] * 0.25f;
    const float s0 = sinpif(w0);
    float acc = s0 * s0;
#pragma unroll
    for (int j = 0; j + 1 < D; ++j) {
      const float w = 1.f + z[j] * 0.25f;
      const float sn = sinf(kPi * w + 1.f);
      acc += (w - 1.f) * (w - 1.f) * (1.f + 10.f * sn * sn);
    }
    const float sl = sinpif(2.f * wl);
    f = acc + (wl - 1.f) * (wl - 1.f) * (1.f + sl * sl);
  }
  out[i] = f < floor_ ? 0.f : f;
}

}  // namespace

// kind: 1 Zakharov, 2 Rosenbrock, 4 Rastrigin, 5 Levy (the function numbers); d ∈ {2, 10, 20}.
EVOX_API int evox_hip_cec_simple(const float* x, int64_t ldx, const float* o, const float* M, float rate, int kind,
                                 float floor_, int n, int d, float* out, void* stream) {
  if (n <= 0) return 0;
  if (!(kind == 1 || kind == 2 || kind == 4 || kind == 5)) return (int)hipErrorInvalidValue;
  const int blocks = (n + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  switch (d) {
    case 2: cec_simple_kernel<2><<<blocks, 256, 0, st>>>(x, ldx, o, M, rate, kind, floor_, n, out); break;
    case 10: cec_simple_kernel<10><<<blocks, 256, 0, st>>>(x, ldx, o, M, rate, kind, floor_, n, out); break;
    case 20: cec_simple_kernel<20><<<blocks, 256, 0, st>>>(x, ldx, o, M, rate, kind, floor_, n, out); break;
    default: return (int)hipErrorInvalidValue;
  }
  return evox::last_launch_status();
}
