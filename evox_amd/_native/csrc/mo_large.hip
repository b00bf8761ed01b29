// One-pass evaluators of the grouped large-scale problems (LSMOP1-9, MaF14, MaF15), of the DTLZ-shaped MaF problems
// (MaF1-7, MaF13), of the WFG-based ones (MaF10-12) and of the line-distance problem MaF9.
//
// Grouped problems: the decision vector is m-1 position variables followed by m groups of nk sub-components; every
// distance variable is linked to x_0 (z = w(col)·x − 10·x_0), every sub-component is put through a landscape function
// (sphere, Griewank, Rosenbrock, Ackley, Schwefel 2.21, Rastrigin) and the group value is the mean over its variables.
// The tensor formulation is a gather, a weight vector and five to ten launches per group.  Here a wavefront owns one
// individual: its lanes stride over the variables of a sub-component (coalesced row reads), the partial sums / products
// / maxima are combined by butterfly shuffles, and lane 0 writes the m objectives.
#include "hip_common.h"

namespace {

constexpr int kMaxObj = 16;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kE = 2.71828182845904523536f;

enum Landscape { kSphere = 0, kGriewank = 1, kRosenbrock = 2, kAckley = 3, kSchwefel = 4, kRastrigin = 5 };
// front: 0 linear (1+g)·chain(x, 1−x); 1 concave (1+g_j+g_{j+1})·chain(cos, sin); 2 disconnected (DTLZ7 form);
//        3 inverted concave (1+g_j+g_{j+1})·(1 − chain(cos, sin))
struct GroupedSpec {
  int m, nk, nonlinear, front;
  int funcs[2], nfuncs;
  int sublen[kMaxObj];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_prod(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// objectives from the m-1 head / tail factors: f_j = scale_j · Π_{i < m−1−j} head_i · tail_{m−1−j} (no tail for j = 0);
// inverted: scale_j · (1 − that)
__device__ __forceinline__ void chain_out(const float* head, const float* tail, const float* scale, int m, bool inverted,
                                          float* o) {
  float prod = 1.f;
  for (int q = 0; q < m; ++q) {
    const int j = m - 1 - q;
    float f = prod;
    if (j > 0) f *= tail[q];
    o[j] = scale[j] * (inverted ? 1.f - f : f);
    if (q < m - 1) prod *= head[q];
  }
}

__global__ __launch_bounds__(256) void grouped_kernel(const float* __restrict__ x, int64_t ldx, int n, int d, GroupedSpec sp,
                                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;                                     // whole wavefronts leave together
  const float* r = x + (int64_t)i * ldx;
  const int m = sp.m;
  const float x0 = r[0] * 10.f;
  const float inv_d = 1.f / (float)d;
  auto linked = [&](int col) {
    const float idx = (float)(col + 1) * inv_d;
    const float w = 1.f + (sp.nonlinear ? cosf(idx * kHalfPi) : idx);
    return w * r[col] - x0;
  };
  float g[kMaxObj];
  int start = m - 1;
  for (int gi = 0; gi < m; ++gi) {
    const int size = sp.sublen[gi];
    const int fn = sp.funcs[gi % sp.nfuncs];
    float total = 0.f;
    for (int c = 0; c < sp.nk && size > 0; ++c) {
      const int base = start + c * size;
      float a = 0.f, b = (fn == kGriewank) ? 1.f : 0.f;
      for (int j = lane; j < size; j += 64) {
        const float z = linked(base + j);
        switch (fn) {
          case kSphere: a += z * z; break;
          case kGriewank: a += z * z; b *= cosf(z / sqrtf((float)(j + 1))); break;
          case kRosenbrock:
            if (j + 1 < size) {
              const float z1 = linked(base + j + 1);
              const float t = z * z - z1;
              a += 100.f * t * t + (z - 1.f) * (z - 1.f);
            }
            break;
          case kAckley: a += z * z; b += cosf(2.f * kPi * z); break;
          case kSchwefel: b = fmaxf(b, fabsf(z)); break;
          default: a += z * z - 10.f * cosf(2.f * kPi * z) + 10.f; break;       // Rastrigin
        }
      }
      float v;
      switch (fn) {
        case kGriewank: v = wave_sum(a) / 4000.f - wave_prod(b) + 1.f; break;
        case kAckley: {
          const float s = wave_sum(a), cs = wave_sum(b);
          v = -20.f * expf(-0.2f * sqrtf(s / (float)size)) - expf(cs / (float)size) + 20.f + kE;
          break;
        }
        case kSchwefel: v = wave_max(b); break;
        default: v = wave_sum(a); break;
      }
      total += v;
    }
    g[gi] = size > 0 ? total / (float)(size * sp.nk) : 0.f;
    start += sp.nk * size;
  }
  if (lane != 0) return;
  float* o = out + (int64_t)i * m;
  if (sp.front == 2) {
    float gs = 1.f;
    for (int j = 0; j < m; ++j) gs += g[j];
    float s = 0.f;
    for (int j = 0; j < m - 1; ++j) {
      const float h = r[j];
      o[j] = h;
      s += h / (1.f + gs) * (1.f + sinf(3.f * kPi * h));
    }
    o[m - 1] = (1.f + gs) * ((float)m - s);
    return;
  }
  float head[kMaxObj], tail[kMaxObj], scale[kMaxObj];
  for (int j = 0; j < m - 1; ++j) {
    const float p = r[j];
    if (sp.front == 0) {
      head[j] = p;
      tail[j] = 1.f - p;
    } else {
      head[j] = cosf(p * kHalfPi);
      tail[j] = sinf(p * kHalfPi);
    }
  }
  for (int j = 0; j < m; ++j) scale[j] = sp.front == 0 ? 1.f + g[j] : 1.f + g[j] + (j + 1 < m ? g[j + 1] : 0.f);
  chain_out(head, tail, scale, m, sp.front == 3, o);
}

// MaF1-7 and MaF13, one lane per individual (a dozen variables per row).  kind: 1..7, 13.
__global__ __launch_bounds__(256) void maf_kernel(const float* __restrict__ x, int64_t ldx, int n, int d, int m, int kind,
                                                  float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (int64_t)i * ldx;
  float* o = out + (int64_t)i * m;
  if (kind == 13) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    int c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    const float x0 = r[0], x1 = r[1];
    for (int j = 2; j < d; ++j) {
      const float y = r[j] - 2.f * x1 * sinf(2.f * kPi * x0 + (float)(j + 1) * kPi / (float)d);
      const float y2 = y * y;
      if (j >= 3) { s4 += y2; ++c4; }
      if (j >= 3 && (j - 3) % 3 == 0) { s1 += y2; ++c1; }
      if (j >= 4 && (j - 4) % 3 == 0) { s2 += y2; ++c2; }
      if ((j - 2) % 3 == 0) { s3 += y2; ++c3; }
    }
    const float a = x0 * kHalfPi, b = x1 * kHalfPi;
    const float f1 = sinf(a) + (c1 ? 2.f * s1 / (float)c1 : 0.f);
    const float f2 = cosf(a) * sinf(b) + (c2 ? 2.f * s2 / (float)c2 : 0.f);
    const float f3 = cosf(a) * cosf(b) + (c3 ? 2.f * s3 / (float)c3 : 0.f);
    const float f2_2 = f2 * f2, f3_2 = f3 * f3;
    const float f2_10 = f2_2 * f2_2 * f2_2 * f2_2 * f2_2, f3_10 = f3_2 * f3_2 * f3_2 * f3_2 * f3_2;
    const float rest = f1 * f1 + f2_10 + f3_10 + (c4 ? 2.f * s4 / (float)c4 : 0.f);
    o[0] = f1; o[1] = f2; o[2] = f3;
    for (int j = 3; j < m; ++j) o[j] = rest;
    return;
  }
  const int k = d - m + 1;
  float g = 0.f;
  float head[kMaxObj], tail[kMaxObj], scale[kMaxObj];
  if (kind == 3 || kind == 4) {
    for (int j = m - 1; j < d; ++j) {
      const float t = r[j] - 0.5f;
      g += t * t - cosf(20.f * kPi * t);
    }
    g = 100.f * ((float)k + g);
  } else if (kind == 7) {
    for (int j = m - 1; j < d; ++j) g += r[j];
    g = 1.f + 9.f * g / (float)k;
    float s = 0.f;
    for (int j = 0; j < m - 1; ++j) {
      const float h = r[j];
      o[j] = h;
      s += h / (1.f + g) * (1.f + sinf(3.f * kPi * h));
    }
    o[m - 1] = (1.f + g) * ((float)m - s);
    return;
  } else if (kind == 2) {
    // one distance function per objective over consecutive slices of the distance variables
    const int step = k / m;
    for (int q = 0; q < m; ++q) {
      const int lo = m - 1 + q * step, hi = q < m - 1 ? lo + step : d;
      float s = 0.f;
      for (int j = lo; j < hi; ++j) {
        const float t = (r[j] * 0.5f + 0.25f) - 0.5f;
        s += t * t;
      }
      scale[q] = 1.f + s;
    }
  } else {
    for (int j = m - 1; j < d; ++j) {
      const float t = r[j] - 0.5f;
      g += t * t;
    }
  }
  for (int j = 0; j < m - 1; ++j) {
    float p = r[j];
    if (kind == 1) {
      head[j] = p;
      tail[j] = 1.f - p;
      continue;
    }
    if (kind == 2) p = p * 0.5f + 0.25f;
    if (kind == 5) p = powf(p, 100.f);
    if (kind == 6 && j > 0) p = (1.f + 2.f * g * p) / (2.f + 2.f * g);
    head[j] = cosf(p * kHalfPi);
    tail[j] = sinf(p * kHalfPi);
  }
  if (kind != 2) {
    const float s = kind == 6 ? 1.f + 100.f * g : 1.f + g;
    for (int j = 0; j < m; ++j) scale[j] = s;
  }
  chain_out(head, tail, scale, m, false, o);
  if (kind == 1 || kind == 4) {
    // inverted: (1+g) − (1+g)·chain; MaF4 also scales objective j by 2^(j+1)
    for (int j = 0; j < m; ++j) {
      const float v = (1.f + g) - o[j];
      o[j] = kind == 4 ? v * exp2f((float)(j + 1)) : v;
    }
  } else if (kind == 3) {
    for (int j = 0; j < m; ++j) {
      const float v = o[j] * o[j];
      o[j] = j < m - 1 ? v * v : v;
    }
  } else if (kind == 5) {
    for (int j = 0; j < m; ++j) o[j] *= exp2f((float)(m - j));
  }
}

// MaF10-12 = WFG1, WFG2, WFG9 with k = m-1 position parameters: the transformation chain of the WFG toolkit
// (shift / bias / reduction), then the shape functions; one lane per individual, the transformed vector in private
// memory (d <= kMaxWfgDim).  kind: 10, 11, 12.
constexpr int kMaxWfgDim = 64;

__device__ __forceinline__ float s_linear(float y, float a) { return fabsf(y - a) / fabsf(floorf(a - y) + a); }
__device__ __forceinline__ float b_flat(float y, float a, float b, float c) {
  const float v = a + fminf(floorf(y - b), 0.f) * a * (b - y) / b - fminf(floorf(c - y), 0.f) * (1.f - a) * (y - c) / (1.f - c);
  return rintf(v * 1e4f) / 1e4f;
}
__device__ __forceinline__ float s_decept(float y, float a, float b, float c) {
  return 1.f + (fabsf(y - a) - b) * (floorf(y - a + b) * (1.f - c + (a - b) / b) / (a - b) +
                                     floorf(a + b - y) * (1.f - c + (1.f - a - b) / b) / (1.f - a - b) + 1.f / b);
}
__device__ __forceinline__ float s_multi(float y, float a, float b, float c) {
  const float t = fabsf(y - c) / 2.f / (floorf(c - y) + c);
  return (1.f + cosf((4.f * a + 2.f) * kPi * (0.5f - t)) + 4.f * b * t * t) / (b + 2.f);
}

__global__ __launch_bounds__(256) void maf_wfg_kernel(const float* __restrict__ x, int64_t ldx, int n, int d, int m, int kind,
                                                      float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = x + (int64_t)i * ldx;
  const int k = m - 1, l = d - k;
  float t[kMaxWfgDim];
  for (int j = 0; j < d; ++j) t[j] = r[j] / (float)(2 * (j + 1));
  float red[kMaxObj];                                   // the m reduced parameters
  if (kind == 10) {
    for (int j = k; j < d; ++j) t[j] = b_flat(s_linear(t[j], 0.35f), 0.8f, 0.75f, 0.85f);
    for (int j = 0; j < d; ++j) t[j] = powf(fmaxf(t[j], 0.f), 0.02f);
    for (int j = 0; j < k; ++j) red[j] = t[j];          // one weighted element each: the weight cancels
    float num = 0.f, den = 0.f;
    for (int j = k; j < d; ++j) {
      const float w = (float)(2 * (j + 1));
      num += t[j] * w;
      den += w;
    }
    red[k] = num / den;
  } else if (kind == 11) {
    for (int j = k; j < d; ++j) t[j] = s_linear(t[j], 0.35f);
    const int pairs = l / 2;
    float acc = 0.f;
    for (int q = 0; q < pairs; ++q) {
      const float a = t[k + 2 * q], b = t[k + 2 * q + 1];
      acc += (a + b + 2.f * fabsf(a - b)) / 3.f;
    }
    for (int j = 0; j < k; ++j) red[j] = t[j];
    red[k] = acc / (float)pairs;
  } else {
    // bias of every variable but the last by the mean of the variables behind it
    float tail = 0.f;
    for (int j = d - 1; j >= 0; --j) {
      const float z = t[j];
      if (j < d - 1) {
        const float y = tail / (float)(d - 1 - j);
        const float e = 0.02f + 49.98f * (0.98f / 49.98f - (1.f - 2.f * y) * fabsf(floorf(0.5f - y) + 0.98f / 49.98f));
        t[j] = powf(fmaxf(z, 0.f), e);
      }
      tail += z;
    }
    for (int j = 0; j < k; ++j) red[j] = s_decept(t[j], 0.35f, 0.001f, 0.05f);
    float sum = 0.f, pair = 0.f;
    for (int j = k; j < d; ++j) {
      t[j] = s_multi(t[j], 30.f, 95.f, 0.35f);
      sum += t[j];
    }
    for (int a = k; a < d; ++a)
      for (int b = a + 1; b < d; ++b) pair += fabsf(t[a] - t[b]);
    const int half = (l + 1) / 2;
    red[k] = (sum + 2.f * pair) / (float)half / (float)(1 + 2 * l - 2 * half);
  }
  // position parameters (degeneracy constants 1), shape functions, objective scales 2, 4, ...
  float pos[kMaxObj], head[kMaxObj], tail_f[kMaxObj], scale[kMaxObj], h[kMaxObj];
  const float last = red[k];
  for (int j = 0; j < k; ++j) pos[j] = fmaxf(last, 1.f) * (red[j] - 0.5f) + 0.5f;
  for (int j = 0; j < k; ++j) {
    const float a = pos[j] * kHalfPi;
    if (kind == 12) { head[j] = sinf(a); tail_f[j] = cosf(a); }
    else { head[j] = 1.f - cosf(a); tail_f[j] = 1.f - sinf(a); }
  }
  for (int j = 0; j < m; ++j) scale[j] = 1.f;
  chain_out(head, tail_f, scale, m, false, h);
  if (kind == 10) h[m - 1] = 1.f - pos[0] - cosf(10.f * kPi * pos[0] + kHalfPi) / (10.f * kPi);
  if (kind == 11) { const float c = cosf(5.f * kPi * pos[0]); h[m - 1] = 1.f - pos[0] * c * c; }
  float* o = out + (int64_t)i * m;
  for (int j = 0; j < m; ++j) o[j] = last + (float)(2 * (j + 1)) * h[j];
}

// MaF9: objective j = distance of the point (x_0, x_1) to the line through the polygon vertices j and j+1.
__global__ __launch_bounds__(256) void maf9_kernel(const float* __restrict__ x, int64_t ldx, int n, int m,
                                                   const float* __restrict__ points, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = x[(int64_t)i * ldx], py = x[(int64_t)i * ldx + 1];
  for (int j = 0; j < m; ++j) {
    const int jn = j + 1 < m ? j + 1 : 0;
    const float ax = points[2 * j], ay = points[2 * j + 1], bx = points[2 * jn], by = points[2 * jn + 1];
    const float num = fabsf((ax - px) * (by - py) - (bx - px) * (ay - py));
    out[(int64_t)i * m + j] = num / sqrtf((ax - bx) * (ax - bx) + (ay - by) * (ay - by));
  }
}

}  // namespace

// sublen: m ints (sub-component length of every group); funcs: nfuncs (1 or 2) landscape ids, group i uses funcs[i % nfuncs]
EVOX_API int evox_hip_mo_grouped(const float* x, int64_t ldx, int n, int d, int m, int nk, const int* sublen, const int* funcs,
                                 int nfuncs, int nonlinear, int front, float* out, void* stream) {
  if (n <= 0 || m < 2 || m > kMaxObj || nk < 1 || nfuncs < 1 || nfuncs > 2 || front < 0 || front > 3)
    return (int)hipErrorInvalidValue;
  GroupedSpec sp;
  sp.m = m; sp.nk = nk; sp.nonlinear = nonlinear; sp.front = front; sp.nfuncs = nfuncs;
  sp.funcs[0] = funcs[0];
  sp.funcs[1] = funcs[nfuncs - 1];
  int used = m - 1;
  for (int j = 0; j < kMaxObj; ++j) {
    sp.sublen[j] = j < m ? sublen[j] : 0;
    if (sp.sublen[j] < 0) return (int)hipErrorInvalidValue;
    used += sp.sublen[j] * nk;
  }
  for (int j = 0; j < nfuncs; ++j)
    if (funcs[j] < 0 || funcs[j] > kRastrigin) return (int)hipErrorInvalidValue;
  if (used > d) return (int)hipErrorInvalidValue;           // the groups must lie inside the row
  grouped_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, ldx, n, d, sp, out);
  return evox::last_launch_status();
}

// MaF9: points = m x 2 polygon vertices (device).
EVOX_API int evox_hip_maf9(const float* x, int64_t ldx, int n, int m, const float* points, float* out, void* stream) {
  if (n <= 0 || m < 2 || m > 64) return (int)hipErrorInvalidValue;
  maf9_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, ldx, n, m, points, out);
  return evox::last_launch_status();
}

EVOX_API int evox_hip_maf(const float* x, int64_t ldx, int n, int d, int m, int kind, float* out, void* stream) {
  if (kind >= 10 && kind <= 12) {
    // k = m-1 position parameters and at least one (WFG2: an even number of) distance parameters
    if (n <= 0 || m < 2 || m > kMaxObj || d > kMaxWfgDim || d < m || (kind == 11 && ((d - m + 1) & 1)))
      return (int)hipErrorInvalidValue;
    maf_wfg_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, ldx, n, d, m, kind, out);
    return evox::last_launch_status();
  }
  const bool known = (kind >= 1 && kind <= 7) || kind == 13;
  if (n <= 0 || !known || m < (kind == 13 ? 3 : 2) || m > kMaxObj || d < m || (kind == 13 && d < 2))
    return (int)hipErrorInvalidValue;
  if (kind == 2 && (d - m + 1) / m < 0) return (int)hipErrorInvalidValue;
  maf_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, ldx, n, d, m, kind, out);
  return evox::last_launch_status();
}
