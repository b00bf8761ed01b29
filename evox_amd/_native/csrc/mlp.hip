// Per-individual MLP forward: every individual p has its own parameter vector flat[p, :] (layer after
// layer, each [W (in x out, row-major), b (out)]) and its own E observations.
//
// The op is a pure stream over the parameters (each weight is used E times), so it is written as a
// bandwidth kernel: one workgroup per individual walks its ~MBs of weights exactly once with wide
// coalesced loads, keeps activations in LDS (ping-pong buffers), fuses bias + activation, and never
// writes an intermediate to HBM.  fp32 or bf16 parameters, fp32 accumulation.
//
//   wide layers  (out > 64): a thread owns 4 consecutive output columns (one 16-byte / 8-byte load per
//                            input row), the LDS read of the input activation is a broadcast;
//   narrow layers (out <= 64): the weight block is walked as a flat array (coalesced), products are
//                            accumulated with LDS float atomics, then bias + activation.
#include "hip_common.h"
#include <hip/hip_bf16.h>

using namespace evox;

namespace {

constexpr int NT = 256;
constexpr int kMaxLayers = 8;

struct MlpDesc {
  int n_layers;
  int in[kMaxLayers], out[kMaxLayers], act[kMaxLayers];     // act: 0 identity, 1 tanh, 2 relu, 3 sigmoid
  int64_t w_off[kMaxLayers], b_off[kMaxLayers];              // b_off < 0: no bias
  int vec4[kMaxLayers];                                      // 16-byte (fp32) / 8-byte (bf16) loads allowed
  int max_width;
};

__device__ __forceinline__ float activate(float x, int act) {
  switch (act) {
    case 1: return tanhf(x);
    case 2: return fmaxf(x, 0.f);
    case 3: return 1.f / (1.f + __expf(-x));
    default: return x;
  }
}

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const uint16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }
__device__ __forceinline__ void ld4(const float* p, float (&w)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
__device__ __forceinline__ void ld4(const uint16_t* p, float (&w)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xffff0000u);
  w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xffff0000u);
}

template <typename T, int E>
__global__ __launch_bounds__(NT) void mlp_forward_kernel(const T* __restrict__ flat, int64_t pitch,
                                                         const float* __restrict__ obs, float* __restrict__ out,
                                                         MlpDesc d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int W = d.max_width;
  float* buf0 = lds;                 // [E][W]
  float* buf1 = lds + E * W;
  const int p = blockIdx.x;
  const T* theta = flat + (int64_t)p * pitch;
  const int t = threadIdx.x;

  const int in0 = d.in[0];
  for (int e = t; e < E * in0; e += NT) buf0[(e / in0) * W + (e % in0)] = obs[(int64_t)p * E * in0 + e];
  __syncthreads();

  float* hin = buf0;
  float* hout = buf1;
  for (int l = 0; l < d.n_layers; ++l) {
    const int nin = d.in[l], nout = d.out[l], act = d.act[l];
    const T* Wl = theta + d.w_off[l];
    const bool has_b = d.b_off[l] >= 0;
    const T* bl = theta + (has_b ? d.b_off[l] : 0);
    if (nout > 64) {
      if (d.vec4[l]) {
        for (int c0 = t * 4; c0 < nout; c0 += NT * 4) {
          float acc[E][4];
#pragma unroll
          for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[e][k] = has_b ? ld1(bl + c0 + k) : 0.f;
          const T* col = Wl + c0;
#pragma unroll 8
          for (int i = 0; i < nin; ++i) {
            float w[4];
            ld4(col + (int64_t)i * nout, w);
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float h = hin[e * W + i];
#pragma unroll
              for (int k = 0; k < 4; ++k) acc[e][k] = fmaf(h, w[k], acc[e][k]);
            }
          }
#pragma unroll
          for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) hout[e * W + c0 + k] = activate(acc[e][k], act);
        }
      } else {
        for (int c = t; c < nout; c += NT) {
          float acc[E];
#pragma unroll
          for (int e = 0; e < E; ++e) acc[e] = has_b ? ld1(bl + c) : 0.f;
          const T* col = Wl + c;
#pragma unroll 8
          for (int i = 0; i < nin; ++i) {
            const float w = ld1(col + (int64_t)i * nout);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = fmaf(hin[e * W + i], w, acc[e]);
          }
#pragma unroll
          for (int e = 0; e < E; ++e) hout[e * W + c] = activate(acc[e], act);
        }
      }
    } else {
      for (int e = t; e < E * nout; e += NT) hout[(e / nout) * W + (e % nout)] = has_b ? ld1(bl + (e % nout)) : 0.f;
      __syncthreads();
      const int total = nin * nout;
      for (int idx = t; idx < total; idx += NT) {
        const float w = ld1(Wl + idx);
        const int i = idx / nout, c = idx - i * nout;
#pragma unroll
        for (int e = 0; e < E; ++e) atomicAdd(&hout[e * W + c], hin[e * W + i] * w);
      }
      __syncthreads();
      for (int e = t; e < E * nout; e += NT) {
        float* q = &hout[(e / nout) * W + (e % nout)];
        *q = activate(*q, act);
      }
    }
    __syncthreads();
    float* tmp = hin; hin = hout; hout = tmp;
  }
  const int nlast = d.out[d.n_layers - 1];
  for (int e = t; e < E * nlast; e += NT) out[(int64_t)p * E * nlast + e] = hin[(e / nlast) * W + (e % nlast)];
}

template <typename T>
int launch(const T* flat, int64_t pitch, const float* obs, float* out, int P, int E, const MlpDesc& d, hipStream_t s) {
  const size_t lds = (size_t)2 * E * d.max_width * sizeof(float);
  switch (E) {
    case 1: mlp_forward_kernel<T, 1><<<P, NT, lds, s>>>(flat, pitch, obs, out, d); break;
    case 2: mlp_forward_kernel<T, 2><<<P, NT, lds, s>>>(flat, pitch, obs, out, d); break;
    case 4: mlp_forward_kernel<T, 4><<<P, NT, lds, s>>>(flat, pitch, obs, out, d); break;
    default: return (int)hipErrorInvalidValue;
  }
  return last_launch_status();
}

}  // namespace

// sizes: n_layers + 1 widths; acts: per layer; dtype: 0 fp32, 1 bf16.  obs (P, E, sizes[0]) fp32,
// out (P, E, sizes[n_layers]) fp32.  E in {1, 2, 4}; every width <= 4096 / E.
EVOX_API int evox_hip_mlp_forward(const void* flat, int dtype, int64_t pitch, const float* obs, float* out, int P, int E,
                                  const int* sizes, const int* acts, int n_layers, int bias, void* stream) {
  if (n_layers < 1 || n_layers > kMaxLayers || P <= 0) return (int)hipErrorInvalidValue;
  MlpDesc d;
  d.n_layers = n_layers;
  d.max_width = 0;
  int64_t off = 0;
  const int64_t esz = dtype == 1 ? 2 : 4;
  for (int l = 0; l < n_layers; ++l) {
    d.in[l] = sizes[l]; d.out[l] = sizes[l + 1]; d.act[l] = acts[l];
    d.w_off[l] = off; off += (int64_t)sizes[l] * sizes[l + 1];
    d.b_off[l] = bias ? off : -1; if (bias) off += sizes[l + 1];
    const bool aligned = ((uintptr_t)flat % 16 == 0) && (pitch % 4 == 0) && (d.w_off[l] % 4 == 0) && (sizes[l + 1] % 4 == 0);
    d.vec4[l] = aligned ? 1 : 0;
    if (sizes[l] > d.max_width) d.max_width = sizes[l];
    if (sizes[l + 1] > d.max_width) d.max_width = sizes[l + 1];
  }
  (void)esz;
  if (off > pitch) return (int)hipErrorInvalidValue;
  d.max_width = (d.max_width + 3) / 4 * 4;
  if ((size_t)2 * E * d.max_width * sizeof(float) > 64 * 1024) return (int)hipErrorInvalidValue;
  if (dtype == 1) return launch<uint16_t>((const uint16_t*)flat, pitch, obs, out, P, E, d, (hipStream_t)stream);
  return launch<float>((const float*)flat, pitch, obs, out, P, E, d, (hipStream_t)stream);
}
