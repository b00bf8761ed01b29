// Fused single-objective benchmark functions: one pass over an (N, D) population, one
// scalar per row.  The population is read exactly once (memory-bound: consecutive lanes read
// consecutive elements, i.e. one fully coalesced 256-byte request per wavefront and step; 16 µs
// for the 40 MB of a 10^4 x 10^3 population), every transcendental is evaluated in registers, and a row is reduced
// inside a group of LPR lanes of one 64-wide wavefront (LPR = 64 for long rows, 16 or 4 for
// short ones so that small-dimension populations still fill the lanes).
#include "hip_common.h"
#include <hip/hip_bf16.h>

using namespace evox;

namespace {

enum Func { kSphere = 0, kAckley = 1, kRastrigin = 2, kRosenbrock = 3, kGriewank = 4, kSchwefel = 5 };

template <typename T> __device__ __forceinline__ float load1(const T* p);
template <> __device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load1<__hip_bfloat16>(const __hip_bfloat16* p) {
  return __bfloat162float(*p);
}

template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPR> __device__ __forceinline__ float group_prod(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v *= __shfl_xor(v, o, 64);
  return v;
}

struct Acc { float s0, s1; };

// cos(2 pi x): x is reduced exactly to r = x - rint(x) in [-1/2, 1/2], folded to s in [0, 1/4] (cos(2 pi r) =
// -cos(2 pi (1/2 - |r|)) beyond a quarter turn), and the even Taylor polynomial through theta^12 is summed on theta in
// [0, pi/2]: truncation below 7e-9, a dozen instructions instead of the general routine's two dozen (this evaluation is
// what bounds the Ackley / Rastrigin passes, not the memory traffic).
__device__ __forceinline__ float cos2pi(float x) {
  const float r = x - rintf(x);
  const float a = fabsf(r);
  const bool flip = a > 0.25f;
  const float sft = flip ? 0.5f - a : a;
  const float u = 39.47841760435743f * sft * sft;          // theta^2
  float p = 2.08767569878681e-9f;                           // 1 / 12!
  p = fmaf(p, u, -2.755731922398589e-7f);                   // -1 / 10!
  p = fmaf(p, u, 2.48015873015873e-5f);                     // 1 / 8!
  p = fmaf(p, u, -1.388888888888889e-3f);                   // -1 / 6!
  p = fmaf(p, u, 4.166666666666666e-2f);                    // 1 / 4!
  p = fmaf(p, u, -0.5f);
  p = fmaf(p, u, 1.0f);
  return flip ? -p : p;
}

template <int FUNC> __device__ __forceinline__ void accumulate(Acc& a, float x, float xnext, int j, bool has_next) {
  if constexpr (FUNC == kSphere) {
    a.s0 = fmaf(x, x, a.s0);
  } else if constexpr (FUNC == kAckley) {
    a.s0 = fmaf(x, x, a.s0);
    a.s1 += cos2pi(x);                 // generic c handled by the caller when c != 2*pi
  } else if constexpr (FUNC == kRastrigin) {
    a.s0 += fmaf(x, x, -10.0f * cos2pi(x));
  } else if constexpr (FUNC == kRosenbrock) {
    if (has_next) {
      const float t = xnext - x * x, u = x - 1.0f;
      a.s0 += fmaf(100.0f * t, t, u * u);
    }
  } else if constexpr (FUNC == kGriewank) {
    a.s0 = fmaf(x, x, a.s0);
    a.s1 *= cosf(x * rsqrtf((float)(j + 1)));
  } else {  // Schwefel
    a.s0 += x * sinf(sqrtf(fabsf(x)));
  }
}

template <int FUNC, int LPR, typename T>
__global__ __launch_bounds__(256) void so_fitness_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t n_rows,
                                                         int dim, int64_t pitch, float p0, float p1, float p2) {
  constexpr int ROWS_PER_WAVE = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t n_groups = (n_rows + ROWS_PER_WAVE - 1) / ROWS_PER_WAVE;
  for (int64_t g = wave_global; g < n_groups; g += n_waves) {
    const int64_t row = g * ROWS_PER_WAVE + lane / LPR;
    const bool live = row < n_rows;
    const T* rp = x + (live ? row : 0) * pitch;
    Acc a;
    a.s0 = 0.0f;
    a.s1 = (FUNC == kGriewank) ? 1.0f : 0.0f;
    if (live) {
      // rows on 16-byte boundaries: four elements per lane and load (1 KiB per wavefront instruction)
      const bool vec4 = sizeof(T) == 4 && FUNC != kRosenbrock && LPR == 64 && (dim & 3) == 0 && (pitch & 3) == 0 &&
                        ((uintptr_t)x & 15) == 0 && !(FUNC == kAckley && p2 != 6.283185307179586f);
      if (vec4) {
        if constexpr (sizeof(T) == 4 && FUNC != kRosenbrock) {
          const float4* rp4 = reinterpret_cast<const float4*>(rp);
          const int nq = dim >> 2;
          // four loads of the row in flight before the first transcendental (4 KiB per wavefront)
          for (int j0 = 0; j0 < nq; j0 += 4 * LPR) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u * LPR + sub;
              v[u] = j < nq ? rp4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + u * LPR + sub;
              if (j < nq) {
                accumulate<FUNC>(a, v[u].x, 0.f, 4 * j, false);
                accumulate<FUNC>(a, v[u].y, 0.f, 4 * j + 1, false);
                accumulate<FUNC>(a, v[u].z, 0.f, 4 * j + 2, false);
                accumulate<FUNC>(a, v[u].w, 0.f, 4 * j + 3, false);
              }
            }
          }
        }
      } else if constexpr (FUNC == kAckley) {
        // general c: cos(c*x); the default c = 2*pi takes the exact cospi path
        if (p2 == 6.283185307179586f) {
          for (int j = sub; j < dim; j += LPR) accumulate<FUNC>(a, load1(rp + j), 0.f, j, false);
        } else {
          for (int j = sub; j < dim; j += LPR) {
            const float v = load1(rp + j);
            a.s0 = fmaf(v, v, a.s0);
            a.s1 += cosf(p2 * v);
          }
        }
      } else if constexpr (FUNC == kRosenbrock) {
        for (int j = sub; j < dim; j += LPR) {
          const bool hn = j + 1 < dim;
          accumulate<FUNC>(a, load1(rp + j), hn ? load1(rp + j + 1) : 0.f, j, hn);
        }
      } else {
        for (int j = sub; j < dim; j += LPR) accumulate<FUNC>(a, load1(rp + j), 0.f, j, false);
      }
    }
    float s0 = group_sum<LPR>(a.s0);
    float s1 = (FUNC == kGriewank) ? group_prod<LPR>(a.s1) : group_sum<LPR>(a.s1);
    if (live && sub == 0) {
      float r;
      const float inv_d = 1.0f / (float)dim;
      if constexpr (FUNC == kSphere) r = s0;
      else if constexpr (FUNC == kAckley)
        r = -p0 * expf(-p1 * sqrtf(s0 * inv_d)) - expf(s1 * inv_d) + p0 + 2.718281828459045f;
      else if constexpr (FUNC == kRastrigin) r = 10.0f * (float)dim + s0;
      else if constexpr (FUNC == kRosenbrock) r = s0;
      else if constexpr (FUNC == kGriewank) r = s0 * (1.0f / 4000.0f) - s1 + 1.0f;
      else r = 418.9828872724338f * (float)dim - s0;
      out[row] = r;
    }
  }
}

template <int FUNC, typename T>
int launch(const T* x, float* out, int64_t n, int dim, int64_t pitch, float p0, float p1, float p2, hipStream_t st) {
  if (n <= 0) return 0;
  const int block = 256;
  if (dim >= 128) {
    const int64_t waves = n;
    int grid = (int)((waves + 3) / 4);
    if (grid > kNumCU * 8) grid = kNumCU * 8;           // every wavefront resident at once (eight per SIMD)
    so_fitness_kernel<FUNC, 64, T><<<grid, block, 0, st>>>(x, out, n, dim, pitch, p0, p1, p2);
  } else if (dim >= 12) {
    const int64_t waves = (n + 3) / 4;
    int grid = (int)((waves + 3) / 4);
    if (grid > kNumCU * 16) grid = kNumCU * 16;
    so_fitness_kernel<FUNC, 16, T><<<grid, block, 0, st>>>(x, out, n, dim, pitch, p0, p1, p2);
  } else {
    const int64_t waves = (n + 15) / 16;
    int grid = (int)((waves + 3) / 4);
    if (grid > kNumCU * 16) grid = kNumCU * 16;
    so_fitness_kernel<FUNC, 4, T><<<grid, block, 0, st>>>(x, out, n, dim, pitch, p0, p1, p2);
  }
  return last_launch_status();
}

template <typename T>
int dispatch(int func, const T* x, float* out, int64_t n, int dim, int64_t pitch, float p0, float p1, float p2,
             hipStream_t st) {
  switch (func) {
    case kSphere: return launch<kSphere>(x, out, n, dim, pitch, p0, p1, p2, st);
    case kAckley: return launch<kAckley>(x, out, n, dim, pitch, p0, p1, p2, st);
    case kRastrigin: return launch<kRastrigin>(x, out, n, dim, pitch, p0, p1, p2, st);
    case kRosenbrock: return launch<kRosenbrock>(x, out, n, dim, pitch, p0, p1, p2, st);
    case kGriewank: return launch<kGriewank>(x, out, n, dim, pitch, p0, p1, p2, st);
    case kSchwefel: return launch<kSchwefel>(x, out, n, dim, pitch, p0, p1, p2, st);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace

// dtype: 0 = fp32, 1 = bf16.  pitch = elements between consecutive rows.
EVOX_API int evox_hip_so_fitness(int func, int dtype, const void* x, float* out, int64_t n, int dim, int64_t pitch,
                                 float p0, float p1, float p2, void* stream) {
  if (dtype == 0) return dispatch<float>(func, (const float*)x, out, n, dim, pitch, p0, p1, p2, (hipStream_t)stream);
  return dispatch<__hip_bfloat16>(func, (const __hip_bfloat16*)x, out, n, dim, pitch, p0, p1, p2, (hipStream_t)stream);
}
