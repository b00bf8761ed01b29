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
//   narrow layers (out <= 64): the weight block is walked as a flat array (coalesced), every thread sums
//                            its rows of one column, the partial sums are added in a fixed order (no
//                            atomics: the same inputs give the same bits), then bias + activation.
#include "hip_common.h"
#include <hip/hip_bf16.h>
#include <cstdlib>

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

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// Parameter rows are read exactly once: non-temporal loads keep them from displacing the data that
// IS reused across workgroups (the centre of a mirrored population) from the L2.
__device__ __forceinline__ float ld1(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld1(const uint16_t* p) {
  return __uint_as_float((uint32_t)__builtin_nontemporal_load(p) << 16);
}
__device__ __forceinline__ void ld4(const float* p, float (&w)[4]) {
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
__device__ __forceinline__ void ld4(const uint16_t* p, float (&w)[4]) {
  const v2u v = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
  w[0] = __uint_as_float(v.x << 16); w[1] = __uint_as_float(v.x & 0xffff0000u);
  w[2] = __uint_as_float(v.y << 16); w[3] = __uint_as_float(v.y & 0xffff0000u);
}
// centre of a mirrored population (fp32, or bf16 next to bf16 rows): ordinary (cached) loads
__device__ __forceinline__ float ld1c(const float* p) { return *p; }
__device__ __forceinline__ float ld1c(const uint16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }
__device__ __forceinline__ void ld4c(const float* p, float (&w)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
__device__ __forceinline__ void ld4c(const uint16_t* p, float (&w)[4]) {
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
  float* red = lds + 2 * E * W;      // [E][parts][nout] partial sums of a narrow layer, at most E * NT floats
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
      // narrow layer: the weight block is walked as a flat array (coalesced) — thread t owns column t % nout and the
      // rows t / nout, t / nout + parts, ...; the partial sums meet in LDS and are added in a FIXED order (bias first,
      // then part 0, 1, ...), so that the same inputs give the same bits on every call
      const int parts = NT / nout, c = t % nout, part = t / nout;
      float acc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) acc[e] = 0.f;
      if (part < parts) {
        for (int i = part; i < nin; i += parts) {
          const float w = ld1(Wl + (int64_t)i * nout + c);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[e] = fmaf(hin[e * W + i], w, acc[e]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) red[(e * parts + part) * nout + c] = acc[e];
      }
      __syncthreads();
      for (int e = t; e < E * nout; e += NT) {
        const int ee = e / nout, cc = e - ee * nout;
        float sum = has_b ? ld1(bl + cc) : 0.f;
        for (int pt = 0; pt < parts; ++pt) sum += red[(ee * parts + pt) * nout + cc];
        hout[ee * W + cc] = activate(sum, act);
      }
    }
    __syncthreads();
    float* tmp = hin; hin = hout; hout = tmp;
  }
  const int nlast = d.out[d.n_layers - 1];
  for (int e = t; e < E * nlast; e += NT) out[(int64_t)p * E * nlast + e] = hin[(e / nlast) * W + (e % nlast)];
}

// Mirrored populations, NP pairs per workgroup.  Row half+p equals 2·center − row p, so a workgroup
// reads the "+" rows of its NP pairs once (HBM, non-temporal) and ONE pass of the centre (cached; its
// cost is shared by the 2·NP individuals), rebuilds the "−" weights in registers and advances the
// 2·NP·E activation vectors together.  Activation set index: (q·2 + sign)·E + e.
template <typename T, typename CT, int E, int NP>
__global__ __launch_bounds__(NT) void mlp_pair_kernel(const T* __restrict__ flat, int64_t pitch,
                                                      const CT* __restrict__ center, int half,
                                                      const float* __restrict__ obs, float* __restrict__ out,
                                                      MlpDesc d) {
  constexpr int EA = 2 * NP * E;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int W = d.max_width;
  float* buf0 = lds;
  float* buf1 = lds + EA * W;
  float* red = lds + 2 * EA * W;                        // partial sums of a narrow layer, at most EA * NT floats
  const int p0 = blockIdx.x * NP;                       // first pair of this workgroup
  const int t = threadIdx.x;
  auto member = [&](int ea) -> int64_t {                 // population row evaluated by activation set ea
    const int q = ea / (2 * E), sign = (ea / E) & 1;
    return (int64_t)(sign ? half : 0) + p0 + q;
  };
  const int in0 = d.in[0];
  for (int e = t; e < EA * in0; e += NT) {
    const int ea = e / in0, c = e - ea * in0;
    buf0[ea * W + c] = obs[(member(ea) * E + (ea % E)) * in0 + c];
  }
  __syncthreads();

  float* hin = buf0;
  float* hout = buf1;
  for (int l = 0; l < d.n_layers; ++l) {
    const int nin = d.in[l], nout = d.out[l], act = d.act[l];
    const bool has_b = d.b_off[l] >= 0;
    const CT* Cl = center + d.w_off[l];
    const CT* cbl = center + (has_b ? d.b_off[l] : 0);
    const T* Wq[NP];
    const T* bq[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const T* theta = flat + (int64_t)(p0 + q) * pitch;
      Wq[q] = theta + d.w_off[l];
      bq[q] = theta + (has_b ? d.b_off[l] : 0);
    }
    if (nout > 64 && d.vec4[l]) {
      for (int c0 = t * 4; c0 < nout; c0 += NT * 4) {
        float acc[EA][4];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float b = has_b ? ld1(bq[q] + c0 + k) : 0.f;
            const float bm = has_b ? 2.f * ld1c(cbl + c0 + k) - b : 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              acc[(q * 2) * E + e][k] = b;
              acc[(q * 2 + 1) * E + e][k] = bm;
            }
          }
#pragma unroll 4
        for (int i = 0; i < nin; ++i) {
          float cw[4];
          ld4c(Cl + (int64_t)i * nout + c0, cw);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            float w[4], wm[4];
            ld4(Wq[q] + (int64_t)i * nout + c0, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) wm[k] = 2.f * cw[k] - w[k];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const float hp = hin[((q * 2) * E + e) * W + i], hm = hin[((q * 2 + 1) * E + e) * W + i];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                acc[(q * 2) * E + e][k] = fmaf(hp, w[k], acc[(q * 2) * E + e][k]);
                acc[(q * 2 + 1) * E + e][k] = fmaf(hm, wm[k], acc[(q * 2 + 1) * E + e][k]);
              }
            }
          }
        }
#pragma unroll
        for (int ea = 0; ea < EA; ++ea)
#pragma unroll
          for (int k = 0; k < 4; ++k) hout[ea * W + c0 + k] = activate(acc[ea][k], act);
      }
    } else if (nout > 64) {
      // wide layer whose rows are not 16-byte aligned: one output column per thread
      for (int c = t; c < nout; c += NT) {
        float acc[EA];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const float b = has_b ? ld1(bq[q] + c) : 0.f;
          const float bm = has_b ? 2.f * ld1c(cbl + c) - b : 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) { acc[(q * 2) * E + e] = b; acc[(q * 2 + 1) * E + e] = bm; }
        }
#pragma unroll 8
        for (int i = 0; i < nin; ++i) {
          const float cw = ld1c(Cl + (int64_t)i * nout + c);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const float w = ld1(Wq[q] + (int64_t)i * nout + c);
            const float wm = 2.f * cw - w;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              acc[(q * 2) * E + e] = fmaf(hin[((q * 2) * E + e) * W + i], w, acc[(q * 2) * E + e]);
              acc[(q * 2 + 1) * E + e] = fmaf(hin[((q * 2 + 1) * E + e) * W + i], wm, acc[(q * 2 + 1) * E + e]);
            }
          }
        }
#pragma unroll
        for (int ea = 0; ea < EA; ++ea) hout[ea * W + c] = activate(acc[ea], act);
      }
    } else {
      // narrow layers (a small share of the parameters): flat coalesced walk, thread t owns column t % nout and every
      // parts-th row; partial sums meet in LDS and are added in a fixed order (see mlp_forward_kernel)
      const int parts = NT / nout, c = t % nout, part = t / nout;
      float acc[EA];
#pragma unroll
      for (int ea = 0; ea < EA; ++ea) acc[ea] = 0.f;
      if (part < parts) {
        for (int i = part; i < nin; i += parts) {
          const int64_t idx = (int64_t)i * nout + c;
          const float cw = ld1c(Cl + idx);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const float w = ld1(Wq[q] + idx);
            const float wm = 2.f * cw - w;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              acc[(q * 2) * E + e] = fmaf(hin[((q * 2) * E + e) * W + i], w, acc[(q * 2) * E + e]);
              acc[(q * 2 + 1) * E + e] = fmaf(hin[((q * 2 + 1) * E + e) * W + i], wm, acc[(q * 2 + 1) * E + e]);
            }
          }
        }
#pragma unroll
        for (int ea = 0; ea < EA; ++ea) red[(ea * parts + part) * nout + c] = acc[ea];
      }
      __syncthreads();
      for (int e = t; e < EA * nout; e += NT) {
        const int ea = e / nout, cc = e - ea * nout;
        const int q = ea / (2 * E), sign = (ea / E) & 1;
        float sum = has_b ? ld1(bq[q] + cc) : 0.f;
        if (has_b && sign) sum = 2.f * ld1c(cbl + cc) - sum;
        for (int pt = 0; pt < parts; ++pt) sum += red[(ea * parts + pt) * nout + cc];
        hout[ea * W + cc] = activate(sum, act);
      }
    }
    __syncthreads();
    float* tmp = hin; hin = hout; hout = tmp;
  }
  const int nlast = d.out[d.n_layers - 1];
  for (int e = t; e < EA * nlast; e += NT) {
    const int ea = e / nlast, c = e - ea * nlast;
    out[(member(ea) * E + (ea % E)) * nlast + c] = hin[ea * W + c];
  }
}

template <typename T, typename CT, int E>
int launch_pairs(const T* flat, int64_t pitch, const CT* center, int half, const float* obs, float* out,
                 const MlpDesc& d, hipStream_t s) {
  // Pairs per workgroup: sharing one centre pass between several pairs was measured slower than one
  // pair per workgroup (98 / 123 / 165 ms per OpenES generation for 1 / 2 / 4 pairs: fewer, fatter
  // workgroups hide less latency and the centre stays cached anyway), so 1 is the default.
  auto lds_of = [&](int np) { return (size_t)(2 * np * E) * (2 * d.max_width + NT) * sizeof(float); };
  const char* cap_env = getenv("EVOX_AMD_MLP_PAIRS");
  const int cap = cap_env ? atoi(cap_env) : 1;
  if (cap >= 4 && half % 4 == 0 && lds_of(4) <= 64 * 1024 && E == 1)
    mlp_pair_kernel<T, CT, E, 4><<<half / 4, NT, lds_of(4), s>>>(flat, pitch, center, half, obs, out, d);
  else if (cap >= 2 && half % 2 == 0 && lds_of(2) <= 64 * 1024)
    mlp_pair_kernel<T, CT, E, 2><<<half / 2, NT, lds_of(2), s>>>(flat, pitch, center, half, obs, out, d);
  else if (lds_of(1) <= 64 * 1024)
    mlp_pair_kernel<T, CT, E, 1><<<half, NT, lds_of(1), s>>>(flat, pitch, center, half, obs, out, d);
  else
    return (int)hipErrorInvalidValue;
  return last_launch_status();
}

template <typename T>
int launch(const T* flat, int64_t pitch, const void* center, int center_dtype, int half, const float* obs, float* out,
           int P, int E, const MlpDesc& d, hipStream_t s) {
  if (center != nullptr) {
    if (center_dtype == 1) {
      const uint16_t* c16 = (const uint16_t*)center;
      if (E == 1) return launch_pairs<T, uint16_t, 1>(flat, pitch, c16, half, obs, out, d, s);
      if (E == 2) return launch_pairs<T, uint16_t, 2>(flat, pitch, c16, half, obs, out, d, s);
    } else {
      const float* c32 = (const float*)center;
      if (E == 1) return launch_pairs<T, float, 1>(flat, pitch, c32, half, obs, out, d, s);
      if (E == 2) return launch_pairs<T, float, 2>(flat, pitch, c32, half, obs, out, d, s);
    }
    return (int)hipErrorInvalidValue;
  }
  const size_t lds = (size_t)E * (2 * d.max_width + NT) * sizeof(float);
#define EVOX_MLP(EE) mlp_forward_kernel<T, EE><<<P, NT, lds, s>>>(flat, pitch, obs, out, d)
  switch (E) {
    case 1: EVOX_MLP(1); break;
    case 2: EVOX_MLP(2); break;
    case 4: EVOX_MLP(4); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef EVOX_MLP
  return last_launch_status();
}

// Small policies (every width <= 64: the 32- or 64-unit networks of locomotion tasks, a few thousand parameters): the
// kernels above give an individual a whole workgroup, which for 1.5 k parameters is all latency (12.8 us per call for
// 4096 individuals).  Here the population is read TRANSPOSED, wt[k][p] (made once per generation by the caller), so the
// 64 lanes of a wave are 64 individuals and every weight load is one coalesced 256-byte row segment whose address is a
// scalar base plus the lane's individual.  A workgroup is 16 waves on the same 64 individuals: wave w accumulates the
// outputs [w·OC, w·OC + OC) of the layer over all inputs (activations are shared through LDS, one barrier per layer), so
// sixteen waves' loads are in flight on the compute unit instead of one wave's.
// MIRROR: the population is mirrored (row half + p = 2·centre − row p): wt holds the first half only, a lane evaluates
// individual p AND its mirror image from one load per weight (the centre value of a weight is one scalar for the wave),
// which halves the bytes read per individual.
template <int MAXW, bool MIRROR>
__global__ __launch_bounds__(1024) void mlp_small_kernel(const float* __restrict__ wt, int pitch,
                                                         const float* __restrict__ center, const float* __restrict__ obs,
                                                         float* __restrict__ out, MlpDesc d, int P, int E, int n_params) {
  constexpr int NW = 16, OC = MAXW / NW, SG = MIRROR ? 2 : 1;
  __shared__ float buf[2][SG][MAXW * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rows = MIRROR ? P / 2 : P;                       // individuals whose parameters are stored
  const int g = blockIdx.x * 64 + lane;                      // (stored individual, observation) pair of this lane
  const bool live = g < rows * E;
  const unsigned p = live ? (unsigned)(g / E) : 0u;
  const int nin0 = d.in[0];
  const int64_t g2 = (int64_t)g + (int64_t)rows * E;         // the same observation slot of the mirror image
  for (int i = wave; i < nin0; i += NW) {
    buf[0][0][i * 64 + lane] = live ? obs[(int64_t)g * nin0 + i] : 0.f;
    if constexpr (MIRROR) buf[0][1][i * 64 + lane] = live ? obs[g2 * nin0 + i] : 0.f;
  }
  __syncthreads();
  // Row k of the transposed population starts at wt + k * pitch: a wave-uniform address (32-bit index arithmetic, so
  // that it stays in scalar registers; the host checks n_params * pitch < 2^31), to which the lane adds its individual.
  auto clampk = [&](int k) { return min(k, n_params - 1); };
  auto weight = [&](int k) { return (wt + (unsigned)clampk(k) * (unsigned)pitch)[p]; };   // cached: a population that fits the L2 stays there between time steps
  int cur = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int nin = d.in[l], nout = d.out[l], act = d.act[l];
    const int w_off = (int)d.w_off[l], b_off = (int)d.b_off[l];
    const int o0 = wave * OC;
    if (o0 < nout) {                                           // wave-uniform
      float acc[SG][OC];
#pragma unroll
      for (int k = 0; k < OC; ++k) {                           // past the layer: never used
        const float w = b_off >= 0 ? weight(b_off + o0 + k) : 0.f;
        acc[0][k] = w;
        if constexpr (MIRROR) acc[1][k] = b_off >= 0 ? fmaf(2.f, center[clampk(b_off + o0 + k)], -w) : 0.f;
      }
      const float* a = buf[cur][0];
      const float* a2 = buf[cur][SG - 1];
      // TRIP inputs per trip, all of their loads issued together (a trip's loads are independent); a trip that runs
      // past the layer's inputs multiplies (clamped, in-matrix) values of the same individual by zero instead of
      // leaving the pipelined form.
      constexpr int TRIP = 16 / OC;                            // 16 loads in flight per wave (24 and 48 were measured slower)
      for (int i0 = 0; i0 < nin; i0 += TRIP) {
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
          const int i = i0 + u;
          const float x = i < nin ? a[i * 64 + lane] : 0.f;
          const float x2 = MIRROR ? (i < nin ? a2[i * 64 + lane] : 0.f) : 0.f;
          const int base = w_off + i * nout + o0;
#pragma unroll
          for (int k = 0; k < OC; ++k) {
            const float w = weight(base + k);
            acc[0][k] = fmaf(x, w, acc[0][k]);
            if constexpr (MIRROR) acc[1][k] = fmaf(x2, fmaf(2.f, center[clampk(base + k)], -w), acc[1][k]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < OC; ++k)
        if (o0 + k < nout) {
          buf[cur ^ 1][0][(o0 + k) * 64 + lane] = activate(acc[0][k], act);
          if constexpr (MIRROR) buf[cur ^ 1][1][(o0 + k) * 64 + lane] = activate(acc[1][k], act);
        }
    }
    __syncthreads();
    cur ^= 1;
  }
  const int nl = d.out[d.n_layers - 1];
  if (live)
    for (int o = wave; o < nl; o += NW) {
      out[(int64_t)g * nl + o] = buf[cur][0][o * 64 + lane];
      if constexpr (MIRROR) out[g2 * nl + o] = buf[cur][1][o * 64 + lane];
    }
}

}  // namespace

// The forward pass for small networks on a TRANSPOSED fp32 population wt (n_params rows of `pitch` values): every width
// <= 64, n_layers <= 8.  obs (P, E, sizes[0]), out (P, E, sizes[n_layers]), both fp32.  center != nullptr: the
// population is mirrored (row P/2 + p = 2·center − row p), wt holds the first P/2 individuals only.
EVOX_API int evox_hip_mlp_small(const float* wt, int64_t pitch, const float* center, const float* obs, float* out, int P, int E,
                                const int* sizes, const int* acts, int n_layers, int bias, void* stream) {
  const int rows = center ? P / 2 : P;
  if (n_layers < 1 || n_layers > kMaxLayers || P <= 0 || E <= 0 || pitch < rows || (center && (P & 1)))
    return (int)hipErrorInvalidValue;
  MlpDesc d;
  d.n_layers = n_layers;
  d.max_width = 0;
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    d.in[l] = sizes[l]; d.out[l] = sizes[l + 1]; d.act[l] = acts[l];
    d.w_off[l] = off; off += (int64_t)sizes[l] * sizes[l + 1];
    d.b_off[l] = bias ? off : -1; if (bias) off += sizes[l + 1];
    d.vec4[l] = 0;
    if (sizes[l] > d.max_width) d.max_width = sizes[l];
    if (sizes[l + 1] > d.max_width) d.max_width = sizes[l + 1];
  }
  if (d.max_width > 64 || d.max_width < 1) return (int)hipErrorInvalidValue;
  const int64_t pairs = (int64_t)rows * E;
  if ((int64_t)P * E >= (1ll << 31) || off * pitch >= (1ll << 31)) return (int)hipErrorInvalidValue;   // 32-bit index arithmetic
  hipStream_t st = (hipStream_t)stream;
  const int groups = (int)((pairs + 63) / 64);
#define EVOX_MLP_SMALL(W, M) mlp_small_kernel<W, M><<<groups, 1024, 0, st>>>(wt, (int)pitch, center, obs, out, d, P, E, (int)off)
  if (center) { if (d.max_width <= 32) EVOX_MLP_SMALL(32, true); else EVOX_MLP_SMALL(64, true); }
  else { if (d.max_width <= 32) EVOX_MLP_SMALL(32, false); else EVOX_MLP_SMALL(64, false); }
#undef EVOX_MLP_SMALL
  return last_launch_status();
}

// sizes: n_layers + 1 widths; acts: per layer; dtype: 0 fp32, 1 bf16.  obs (P, E, sizes[0]) fp32,
// out (P, E, sizes[n_layers]) fp32.  E in {1, 2, 4}; every width <= 4096 / E.
// center != nullptr: the population is mirrored, row half + p = 2·center − row p for p < half = P/2
// (center fp32, or bf16 when center_dtype == 1, with the layout of one row); only rows [0, half) are
// read.  E in {1, 2} then.
static int mlp_dispatch(const void* flat, int dtype, int64_t pitch, const void* center, int center_dtype,
                        const float* obs, float* out, int P, int E, const int* sizes, const int* acts, int n_layers, int bias, void* stream) {
  if (n_layers < 1 || n_layers > kMaxLayers || P <= 0) return (int)hipErrorInvalidValue;
  if (center && (P % 2)) return (int)hipErrorInvalidValue;
  MlpDesc d;
  d.n_layers = n_layers;
  d.max_width = 0;
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    d.in[l] = sizes[l]; d.out[l] = sizes[l + 1]; d.act[l] = acts[l];
    d.w_off[l] = off; off += (int64_t)sizes[l] * sizes[l + 1];
    d.b_off[l] = bias ? off : -1; if (bias) off += sizes[l + 1];
    const bool aligned = ((uintptr_t)flat % 16 == 0) && (pitch % 4 == 0) && (d.w_off[l] % 4 == 0) && (sizes[l + 1] % 4 == 0) &&
                         (!center || (uintptr_t)center % 16 == 0);
    d.vec4[l] = aligned ? 1 : 0;
    if (sizes[l] > d.max_width) d.max_width = sizes[l];
    if (sizes[l + 1] > d.max_width) d.max_width = sizes[l + 1];
  }
  if (off > pitch) return (int)hipErrorInvalidValue;
  d.max_width = (d.max_width + 3) / 4 * 4;
  const int ea = center ? 2 * E : E;
  if ((size_t)ea * (2 * d.max_width + NT) * sizeof(float) > 64 * 1024) return (int)hipErrorInvalidValue;
  const int half = P / 2;
  if (dtype == 1)
    return launch<uint16_t>((const uint16_t*)flat, pitch, center, center_dtype, half, obs, out, P, E, d, (hipStream_t)stream);
  return launch<float>((const float*)flat, pitch, center, center_dtype, half, obs, out, P, E, d, (hipStream_t)stream);
}

EVOX_API int evox_hip_mlp_forward(const void* flat, int dtype, int64_t pitch, const float* obs, float* out, int P, int E,
                                  const int* sizes, const int* acts, int n_layers, int bias, void* stream) {
  return mlp_dispatch(flat, dtype, pitch, nullptr, 0, obs, out, P, E, sizes, acts, n_layers, bias, stream);
}

EVOX_API int evox_hip_mlp_forward_mirrored(const void* flat, int dtype, int64_t pitch, const void* center,
                                           int center_dtype, const float* obs, float* out, int P, int E, const int* sizes, const int* acts,
                                           int n_layers, int bias, void* stream) {
  if (!center) return (int)hipErrorInvalidValue;
  return mlp_dispatch(flat, dtype, pitch, center, center_dtype, obs, out, P, E, sizes, acts, n_layers, bias, stream);
}
