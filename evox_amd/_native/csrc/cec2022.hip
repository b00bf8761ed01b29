// CEC 2022 single-objective suite, functions 3 and 6-12 in one launch each (1, 2, 4, 5 are in moproblems.hip):
// shift, rotation(s), variable shuffle, the basic functions of a hybrid or the components and weights of a composition,
// and the zero floor, for one individual per thread.  The suite is defined for D in {2, 10, 20}; the rotation matrices
// (up to six of D x D), shifts and shuffle sit in LDS, and so do the individual's point and its current image z, stored
// column-wise (entry k of thread t at [k * 256 + t]: conflict-free, and addressable with run-time indices, which the
// shuffled groups of the hybrids need).
#include "hip_common.h"

namespace {

constexpr int NT = 256;
constexpr float kPiF = 3.14159265358979323846f;

struct Vec {                       // n entries of one thread's column vector in LDS
  const float* p;
  int n;
  __device__ __forceinline__ float operator[](int j) const { return p[j * NT]; }
};

__device__ float f_bent_cigar(Vec z) {
  float s = 0.f;
  for (int j = 1; j < z.n; ++j) s = fmaf(z[j], z[j], s);
  return z[0] * z[0] + 1e6f * s;
}
__device__ float f_discus(Vec z) {
  float s = 0.f;
  for (int j = 1; j < z.n; ++j) s = fmaf(z[j], z[j], s);
  return 1e6f * z[0] * z[0] + s;
}
__device__ float f_elliptic(Vec z) {
  float s = 0.f;
  for (int j = 0; j < z.n; ++j) s += exp10f(6.f * (float)j / (float)(z.n > 1 ? z.n - 1 : 1)) * z[j] * z[j];
  return s;
}
__device__ float f_hgbat(Vec z) {
  float sq = 0.f, s = 0.f;
  for (int j = 0; j < z.n; ++j) { const float v = z[j] * 0.05f - 1.f; sq = fmaf(v, v, sq); s += v; }
  return sqrtf(fabsf(sq * sq - s * s)) + (0.5f * sq + s) / (float)z.n + 0.5f;
}
__device__ float f_happycat(Vec z) {
  float sq = 0.f, s = 0.f;
  for (int j = 0; j < z.n; ++j) { const float v = z[j] * 0.05f - 1.f; sq = fmaf(v, v, sq); s += v; }
  return powf(fabsf(sq - (float)z.n), 0.25f) + (0.5f * sq + s) / (float)z.n + 0.5f;
}
__device__ float f_rastrigin(Vec z) {
  float s = 0.f;
  for (int j = 0; j < z.n; ++j) { const float v = z[j] * 0.0512f; s += v * v - 10.f * cospif(2.f * v) + 10.f; }
  return s;
}
__device__ float f_rosenbrock(Vec z) {
  float a = 0.f, b = 0.f;
  for (int j = 0; j + 1 < z.n; ++j) {
    const float u = z[j] + 1.f, v = z[j + 1] + 1.f, d = u * u - v;
    a = fmaf(d, d, a);
    b = fmaf(1.f - u, 1.f - u, b);
  }
  return 100.f * a + b;
}
__device__ float f_ackley(Vec z) {
  float sq = 0.f, c = 0.f;
  for (int j = 0; j < z.n; ++j) { sq = fmaf(z[j], z[j], sq); c += cospif(2.f * z[j]); }
  const float d = (float)z.n;
  return -20.f * expf(-0.2f * sqrtf(sq / d)) - expf(c / d) + 20.f + 2.718281828459045f;
}
__device__ float f_griewank(Vec z) {
  float sq = 0.f, pr = 1.f;
  for (int j = 0; j < z.n; ++j) { sq = fmaf(z[j], z[j], sq); pr *= cosf(z[j] / sqrtf((float)(j + 1))); }
  return sq / 4000.f - pr + 1.f;
}
__device__ float f_schaffer_f7(Vec z) {
  float f = 0.f;
  for (int j = 0; j + 1 < z.n; ++j) {
    const float s = sqrtf(z[j] * z[j] + z[j + 1] * z[j + 1]);
    const float sn = sinf(50.f * powf(s, 0.2f));
    f += sqrtf(s) * (1.f + sn * sn);
  }
  const float m = (float)(z.n - 1);
  return f * f / m / m;
}
__device__ float f_exp_schaffer(Vec z) {
  float f = 0.f;
  for (int j = 0; j < z.n; ++j) {
    const float a = z[j], b = z[j == 0 ? z.n - 1 : j - 1];
    const float sq = a * a + b * b, sn = sinf(sqrtf(sq)), den = 1.f + 0.001f * sq;
    f += 0.5f + (sn * sn - 0.5f) / (den * den);
  }
  return f;
}
__device__ float f_exp_grie_rosen(Vec z) {
  float f = 0.f;
  for (int j = 0; j < z.n; ++j) {
    const float a = z[j] * 0.05f + 1.f, b = z[j + 1 == z.n ? 0 : j + 1] * 0.05f + 1.f;
    const float d = a * a - b, t = 100.f * d * d + (a - 1.f) * (a - 1.f);
    f += t * t / 4000.f - cosf(t) + 1.f;
  }
  return f;
}
__device__ float f_katsuura(Vec z) {
  const float nx = (float)z.n;
  const float e = 10.f / powf(nx, 1.2f);
  float f = 1.f;
  for (int j = 0; j < z.n; ++j) {
    const float v = z[j] * 0.05f;
    float temp = 0.f, p = 2.f;
    for (int q = 0; q < 32; ++q) {
      const float t = p * v;
      temp += fabsf(t - floorf(t + 0.5f)) / p;
      p *= 2.f;
    }
    f *= powf(1.f + (float)(j + 1) * temp, e);
  }
  const float c = 10.f / nx / nx;
  return f * c - c;
}
__device__ float f_schwefel(Vec z) {
  const float nx = (float)z.n;
  float s = 0.f;
  for (int j = 0; j < z.n; ++j) {
    const float v = z[j] * 10.f + 420.9687462275036f;
    float t;
    if (v > 500.f) {
      const float m = fmodf(v, 500.f), q = (v - 500.f) / 100.f;
      t = -(500.f - m) * sinf(sqrtf(fabsf(500.f - m))) + q * q / nx;
    } else if (v < -500.f) {
      const float m = fmodf(fabsf(v), 500.f), q = (v + 500.f) / 100.f;
      t = -(-500.f + m) * sinf(sqrtf(fabsf(500.f - m))) + q * q / nx;
    } else {
      t = -v * sinf(sqrtf(fabsf(v)));
    }
    s += t;
  }
  return s + 418.9828872724338f * nx;
}

enum Basic { kRosen, kEllip, kCigar, kDiscus, kSchwefel, kRastrigin, kHgbat, kExpSchaffer, kGriewank };

__device__ float basic(int which, Vec z) {
  switch (which) {
    case kRosen: return f_rosenbrock(z);
    case kEllip: return f_elliptic(z);
    case kCigar: return f_bent_cigar(z);
    case kDiscus: return f_discus(z);
    case kSchwefel: return f_schwefel(z);
    case kRastrigin: return f_rastrigin(z);
    case kHgbat: return f_hgbat(z);
    case kExpSchaffer: return f_exp_schaffer(z);
    default: return f_griewank(z);
  }
}

// one component of a composition function
struct Comp { int basic; float rate; int rotate; float bias, sigma, lambda; };

struct Recipe {
  int ncomp;          // compositions: number of components; 0 otherwise
  Comp comp[6];
};

template <int D>
__global__ __launch_bounds__(NT) void cec_full_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ o,
                                                      int64_t ldo, const float* __restrict__ M, const int64_t* __restrict__ S,
                                                      int func, Recipe rc, int nmat, float floor_, int n,
                                                      float* __restrict__ out) {
  __shared__ float sm[6 * D * D];
  __shared__ float so[6 * D];
  __shared__ int ss[D];
  __shared__ float xs[D * NT], zs[D * NT];
  for (int t = threadIdx.x; t < nmat * D * D; t += NT) sm[t] = M[t];
  for (int t = threadIdx.x; t < 6 * D; t += NT) so[t] = (t / D) < (rc.ncomp > 0 ? rc.ncomp : 1) ? o[(int64_t)(t / D) * ldo + t % D] : 0.f;
  for (int t = threadIdx.x; t < D; t += NT) ss[t] = S != nullptr ? (int)S[t] : t;
  __syncthreads();
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  float* xc = xs + threadIdx.x;
  float* zc = zs + threadIdx.x;
  const float* r = x + (int64_t)i * ldx;
  for (int k = 0; k < D; ++k) xc[k * NT] = r[k];

  // z = ((x - o_c) * rate) M_c^T, or the plain shift
  auto image = [&](int c, float rate, bool rotate) {
    if (!rotate) {
      for (int j = 0; j < D; ++j) zc[j * NT] = (xc[j * NT] - so[c * D + j]) * rate;
      return;
    }
    for (int j = 0; j < D; ++j) {
      float acc = 0.f;
      for (int k = 0; k < D; ++k) acc = fmaf(sm[(c * D + j) * D + k], (xc[k * NT] - so[c * D + k]) * rate, acc);
      zc[j * NT] = acc;
    }
  };

  float f;
  if (func == 3) {
    image(0, 1.f, false);
    f = f_schaffer_f7(Vec{zc, D});
  } else if (func >= 6 && func <= 8) {
    // hybrids: rotate, then gather the shuffled order into xs (the point itself is no longer needed)
    image(0, 1.f, true);
    for (int j = 0; j < D; ++j) xc[j * NT] = zc[ss[j] * NT];
    const float* g = xc;
    auto part = [&](int lo, int cnt) { return Vec{g + lo * NT, cnt}; };
    if (func == 6) {                     // 0.4, 0.4, 0.2
      const int a = (D * 4) / 10, b = (D * 4) / 10, c = D - a - b;
      f = f_bent_cigar(part(0, a)) + f_hgbat(part(a, b)) + f_rastrigin(part(a + b, c));
    } else if (func == 7) {              // 0.1, 0.2, 0.2, 0.2, 0.1, 0.2
      const int u = D / 10, v = D / 5;
      f = f_hgbat(part(0, u)) + f_katsuura(part(u, v)) + f_ackley(part(u + v, v)) + f_rastrigin(part(u + 2 * v, v)) +
          f_schwefel(part(u + 3 * v, u)) + f_schaffer_f7(part(0, v));
    } else {                             // 0.3, 0.2, 0.2, 0.1, 0.2
      const int a = (D * 3) / 10, v = D / 5, u = D / 10;
      f = f_katsuura(part(0, a)) + f_happycat(part(a, v)) + f_exp_grie_rosen(part(a + v, v)) + f_schwefel(part(a + 2 * v, u)) +
          f_ackley(part(a + 2 * v + u, v));
    }
  } else {
    // compositions: weights from the distances to the component optima
    float wsum = 0.f, acc = 0.f;
    float wv[6], fv[6];
    bool any_opt = false;
    for (int c = 0; c < rc.ncomp; ++c) {
      float d2 = 0.f;
      for (int k = 0; k < D; ++k) { const float d = xc[k * NT] - so[c * D + k]; d2 = fmaf(d, d, d2); }
      const Comp cc = rc.comp[c];
      image(c, cc.rate, cc.rotate != 0);
      fv[c] = cc.lambda * basic(cc.basic, Vec{zc, D}) + cc.bias;
      if (d2 == 0.f) { any_opt = true; wv[c] = -1.f; }
      else wv[c] = expf(-0.5f * d2 / (cc.sigma * cc.sigma * (float)D)) / sqrtf(d2);
    }
    for (int c = 0; c < rc.ncomp; ++c) {
      const float w = any_opt ? (wv[c] < 0.f ? 1.f : 0.f) : wv[c];
      wsum += w;
      acc = fmaf(w, fv[c], acc);
    }
    f = acc / wsum;
  }
  out[i] = f < floor_ ? 0.f : f;
}

Recipe recipe(int func) {
  Recipe r{};
  auto set = [&](int c, int b, float rate, int rot, float bias, float sigma, float lambda) {
    r.comp[c] = Comp{b, rate, rot, bias, sigma, lambda};
  };
  const float rs = 2.048f / 100.f;
  switch (func) {
    case 9:
      r.ncomp = 5;
      set(0, kRosen, rs, 1, 0.f, 10.f, 1.f); set(1, kEllip, 1.f, 1, 200.f, 20.f, 1e-6f); set(2, kCigar, 1.f, 1, 300.f, 30.f, 1e-26f);
      set(3, kDiscus, 1.f, 1, 100.f, 40.f, 1e-6f); set(4, kEllip, 1.f, 0, 400.f, 50.f, 1e-6f);
      break;
    case 10:
      r.ncomp = 3;
      set(0, kSchwefel, 1.f, 0, 0.f, 20.f, 1.f); set(1, kRastrigin, 1.f, 1, 200.f, 10.f, 1.f); set(2, kHgbat, 1.f, 1, 100.f, 10.f, 1.f);
      break;
    case 11:
      r.ncomp = 5;
      set(0, kExpSchaffer, 1.f, 1, 0.f, 20.f, 5e-4f); set(1, kSchwefel, 1.f, 1, 200.f, 20.f, 1.f);
      set(2, kGriewank, 6.f, 1, 300.f, 30.f, 10.f); set(3, kRosen, rs, 1, 400.f, 30.f, 1.f); set(4, kRastrigin, 1.f, 1, 200.f, 20.f, 10.f);
      break;
    case 12:
      r.ncomp = 6;
      set(0, kHgbat, 1.f, 1, 0.f, 10.f, 10.f); set(1, kRastrigin, 1.f, 1, 300.f, 20.f, 10.f); set(2, kSchwefel, 1.f, 1, 500.f, 30.f, 2.5f);
      set(3, kCigar, 1.f, 1, 100.f, 40.f, 1e-26f); set(4, kEllip, 1.f, 1, 400.f, 50.f, 1e-6f);
      set(5, kExpSchaffer, 1.f, 1, 200.f, 60.f, 5e-4f);
      break;
    default: break;
  }
  return r;
}

}  // namespace

// x: n x d rows (row stride ldx); o: shift vectors, one row of stride ldo per component (one row for 3, 6-8);
// M: nmat matrices of d x d (row-major; nmat >= the components that rotate, 1 for the hybrids, may be 0 for function 3);
// S: the 0-based shuffle of the hybrids (int64, d entries) or null.  d in {2, 10, 20}; hybrids need d >= 10.
EVOX_API int evox_hip_cec_full(const float* x, int64_t ldx, const float* o, int64_t ldo, const float* M, int nmat,
                               const int64_t* S, int func, float floor_, int n, int d, float* out, void* stream) {
  if (n <= 0) return 0;
  if (!(func == 3 || (func >= 6 && func <= 12)) || nmat < 0 || nmat > 6) return (int)hipErrorInvalidValue;
  if (func >= 6 && func <= 8 && (d < 10 || S == nullptr || nmat < 1)) return (int)hipErrorInvalidValue;
  const Recipe rc = recipe(func);
  if (rc.ncomp > 0 && nmat < rc.ncomp - (func == 9 ? 1 : 0) && func != 10) return (int)hipErrorInvalidValue;
  if (func == 10 && nmat < 3) return (int)hipErrorInvalidValue;
  const int blocks = (n + NT - 1) / NT;
  hipStream_t st = (hipStream_t)stream;
  switch (d) {
    case 2: cec_full_kernel<2><<<blocks, NT, 0, st>>>(x, ldx, o, ldo, M, S, func, rc, nmat, floor_, n, out); break;
    case 10: cec_full_kernel<10><<<blocks, NT, 0, st>>>(x, ldx, o, ldo, M, S, func, rc, nmat, floor_, n, out); break;
    case 20: cec_full_kernel<20><<<blocks, NT, 0, st>>>(x, ldx, o, ldo, M, S, func, rc, nmat, floor_, n, out); break;
    default: return (int)hipErrorInvalidValue;
  }
  return evox::last_launch_status();
}
