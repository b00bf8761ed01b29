// Adaptive-DE bookkeeping (JADE / SHADE) in three launches instead of ~40 tensor ops:
//
//   shade_params   per individual: memory slot r = mulhi(bits(k_slot, i), H),
//                  F = clip(N(0,1)(kF, i)·0.1 + M_F[r]), CR = clip(Cauchy(kCR, i)·0.1 + M_CR[r])
//                  (JADE: H = 1 with the two running means as the "memory")
//   de_select      greedy replacement trial-vs-parent for the whole population, plus per-block
//                  partial sums of the success statistics  Σw, Σw·CR, Σw·F², Σw·F  (w = the fitness
//                  improvement for SHADE, 1 for JADE) and the block's best (fitness, index)
//   de_finalize    one workgroup folds the partials: best index, and either SHADE's history
//                  memory (shift right, new pair in slot 0) or JADE's running means
//
// The random elements are the ones the tensor formulation draws (same keys, same element index).
#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ uint32_t bits_at(const int32_t* key, uint64_t e) {
  const u32x4 b = philox_block((uint32_t)key[0], (uint32_t)key[1], e >> 2, kStreamDraw);
  switch ((int)(e & 3)) { case 0: return b.x; case 1: return b.y; case 2: return b.z; default: return b.w; }
}

__device__ __forceinline__ float normal_at(const int32_t* key, uint64_t e) {
  const f32x4 z = normal4(philox_block((uint32_t)key[0], (uint32_t)key[1], e >> 2, kStreamDraw));
  return lane_of(z, (int)(e & 3));
}

__global__ __launch_bounds__(256) void shade_params_kernel(const int32_t* __restrict__ k_slot,
                                                           const int32_t* __restrict__ k_f,
                                                           const int32_t* __restrict__ k_cr,
                                                           const float* __restrict__ mem_f,
                                                           const float* __restrict__ mem_cr, int H, int n,
                                                           float* __restrict__ F, float* __restrict__ CR) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int slot = H > 1 ? (int)bits_to_range(bits_at(k_slot, (uint64_t)i), (uint32_t)H) : 0;
  const float f = normal_at(k_f, (uint64_t)i) * 0.1f + mem_f[slot];
  const float u = bits_to_unit(bits_at(k_cr, (uint64_t)i));
  const float cr = tanf(kPi * (u - 0.5f)) * 0.1f + mem_cr[slot];
  F[i] = fminf(fmaxf(f, 0.f), 1.f);
  CR[i] = fminf(fmaxf(cr, 0.f), 1.f);
}

// partial layout per block: {Σw, Σw·CR, Σw·F², Σw·F, best fitness, best index (as int bits)}
constexpr int kPart = 6;

// LPR lanes cooperate on one row (coalesced row copies); 256 / LPR rows per block
template <int LPR>
__global__ __launch_bounds__(256) void de_select_kernel(const float* __restrict__ pop, const float* __restrict__ trial,
                                                        const float* __restrict__ fit,
                                                        const float* __restrict__ trial_fit,
                                                        const float* __restrict__ F, const float* __restrict__ CR,
                                                        int strict, int weight_by_gain, int n, int d, int vec,
                                                        float* __restrict__ new_pop, float* __restrict__ new_fit,
                                                        float* __restrict__ partials) {
  constexpr int RPB = 256 / LPR;
  __shared__ float red[4][kPart];
  const int sub = threadIdx.x % LPR;
  float w = 0.f, wcr = 0.f, wf2 = 0.f, wf = 0.f, best = INFINITY;
  int best_i = 0x7FFFFFFF;
  // a workgroup takes the row groups blockIdx.x, blockIdx.x + gridDim.x, …: one group each up to kSelectMaxBlocks groups,
  // several beyond (a million rows as a quarter of a million workgroups were bound by workgroup launches)
  const int groups = (n + RPB - 1) / RPB;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int r = g * RPB + threadIdx.x / LPR;
    if (r >= n) continue;
    const float pf = fit[r], tf = trial_fit[r];
    const bool ok = strict ? (tf < pf) : (tf <= pf);
    const float* src = (ok ? trial : pop) + (int64_t)r * d;
    float* dst = new_pop + (int64_t)r * d;
    if (vec) {
      for (int c = sub; c < (d >> 2); c += LPR)
        reinterpret_cast<float4*>(dst)[c] = reinterpret_cast<const float4*>(src)[c];
    } else {
      for (int c = sub; c < d; c += LPR) dst[c] = src[c];
    }
    if (sub == 0) {
      const float nf = ok ? tf : pf;
      new_fit[r] = nf;
      const float key = nf == nf ? nf : INFINITY;
      if (key < best || (key == best && r < best_i)) { best = key; best_i = r; }
      if (ok) {
        float gain = weight_by_gain ? pf - tf : 1.f;
        if (!(fabsf(gain) < INFINITY)) gain = 0.f;           // inf − x or NaN: no weight
        const float f = F[r], cr = CR[r];
        w += gain; wcr += gain * cr; wf2 += gain * f * f; wf += gain * f;
      }
    }
  }
  // wave reduction (sum ×4, lexicographic min of (fitness, index))
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    w += __shfl_xor(w, o, 64); wcr += __shfl_xor(wcr, o, 64);
    wf2 += __shfl_xor(wf2, o, 64); wf += __shfl_xor(wf, o, 64);
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = w; red[wave][1] = wcr; red[wave][2] = wf2; red[wave][3] = wf;
    red[wave][4] = best; red[wave][5] = __int_as_float(best_i);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* p = partials + (int64_t)blockIdx.x * kPart;
    float b = red[0][4];
    int bi = __float_as_int(red[0][5]);
    for (int v = 1; v < 4; ++v) {
      const float ob = red[v][4];
      const int oi = __float_as_int(red[v][5]);
      if (ob < b || (ob == b && oi < bi)) { b = ob; bi = oi; }
    }
    p[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    p[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    p[2] = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    p[3] = (red[0][3] + red[1][3]) + (red[2][3] + red[3][3]);
    p[4] = b;
    p[5] = __int_as_float(bi);
  }
}

// Sum (four weights) and lexicographic minimum (fitness, index) of the records first, first + stride, … < last, by
// one 256-thread workgroup; the result is valid in thread 0.
struct DeTotals { float s[4]; float best; int best_i; };

__device__ inline DeTotals de_reduce_records(const float* partials, int64_t first, int64_t last, int64_t stride,
                                             float (*red)[kPart]) {
  float w = 0.f, wcr = 0.f, wf2 = 0.f, wf = 0.f, best = INFINITY;
  int best_i = 0x7FFFFFFF;
  for (int64_t b = first + (int64_t)threadIdx.x * stride; b < last; b += (int64_t)blockDim.x * stride) {
    const float* p = partials + b * kPart;
    w += p[0]; wcr += p[1]; wf2 += p[2]; wf += p[3];
    const float ob = p[4];
    const int oi = __float_as_int(p[5]);
    if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    w += __shfl_xor(w, o, 64); wcr += __shfl_xor(wcr, o, 64);
    wf2 += __shfl_xor(wf2, o, 64); wf += __shfl_xor(wf, o, 64);
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = w; red[wave][1] = wcr; red[wave][2] = wf2; red[wave][3] = wf;
    red[wave][4] = best; red[wave][5] = __int_as_float(best_i);
  }
  __syncthreads();
  DeTotals t{};
  if (threadIdx.x == 0) {
    float b = red[0][4];
    int bi = __float_as_int(red[0][5]);
    for (int v = 1; v < 4; ++v) {
      const float ob = red[v][4];
      const int oi = __float_as_int(red[v][5]);
      if (ob < b || (ob == b && oi < bi)) { b = ob; bi = oi; }
    }
    for (int k = 0; k < 4; ++k) t.s[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    t.best = b;
    t.best_i = bi;
  }
  return t;
}

// Long populations leave hundreds of thousands of records: workgroup g folds records [g·chunk, (g+1)·chunk) into the
// first record of its own chunk (nobody else reads that chunk), and the finalising workgroup then reads every chunk-th
// record — one workgroup walking 250 000 records alone took as long as the selection pass itself.
__global__ __launch_bounds__(256) void de_fold_kernel(float* partials, int nblocks, int chunk) {
  __shared__ float red[4][kPart];
  const int64_t first = (int64_t)blockIdx.x * chunk;
  const int64_t last = min((int64_t)nblocks, first + chunk);
  const DeTotals t = de_reduce_records(partials, first, last, 1, red);
  if (threadIdx.x == 0) {
    float* p = partials + first * kPart;
    p[0] = t.s[0]; p[1] = t.s[1]; p[2] = t.s[2]; p[3] = t.s[3];
    p[4] = t.best; p[5] = __int_as_float(t.best_i);
  }
}

// mode 0 (SHADE): mem (2×H) → mem_out with the new (M_F, M_CR) in slot 0 and the rest shifted right,
//                 unchanged when nothing improved or a mean is not finite.
// mode 1 (JADE):  mem = (F_u, CR_u), H = 1: mean ← (1−c)·mean + c·(Lehmer mean of F | mean of CR).
// The records read are 0, stride, 2·stride, … < nblocks.
__global__ __launch_bounds__(256) void de_finalize_kernel(const float* __restrict__ partials, int nblocks, int stride,
                                                          int mode, float c, const float* __restrict__ mem_f,
                                                          const float* __restrict__ mem_cr, int H,
                                                          float* __restrict__ out_f, float* __restrict__ out_cr,
                                                          int64_t* __restrict__ best_index) {
  __shared__ float red[4][kPart];
  __shared__ float tot[kPart];
  const DeTotals t = de_reduce_records(partials, 0, nblocks, stride, red);
  if (threadIdx.x == 0) {
    for (int k = 0; k < 4; ++k) tot[k] = t.s[k];
    best_index[0] = t.best_i == 0x7FFFFFFF ? 0 : t.best_i;
  }
  __syncthreads();
  const float sw = tot[0];
  const float m_cr = tot[1] / sw, m_f = tot[2] / tot[3];
  const bool finite = fabsf(m_cr) < INFINITY && fabsf(m_f) < INFINITY;     // false for NaN too
  if (mode == 0) {
    const bool valid = sw > 0.f && finite;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
      out_f[j] = valid ? (j == 0 ? m_f : mem_f[j - 1]) : mem_f[j];
      out_cr[j] = valid ? (j == 0 ? m_cr : mem_cr[j - 1]) : mem_cr[j];
    }
  } else if (threadIdx.x == 0) {
    const bool any = sw > 0.f;
    out_f[0] = any ? (1.f - c) * mem_f[0] + c * (m_f == m_f ? m_f : 0.f) : mem_f[0];
    out_cr[0] = any ? (1.f - c) * mem_cr[0] + c * (m_cr == m_cr ? m_cr : 0.f) : mem_cr[0];
  }
}

}  // namespace

EVOX_API int evox_hip_shade_params(const int32_t* k_slot, const int32_t* k_f, const int32_t* k_cr, const float* mem_f,
                                   const float* mem_cr, int H, int n, float* F, float* CR, void* stream) {
  if (n <= 0 || H < 1) return (int)hipErrorInvalidValue;
  shade_params_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(k_slot, k_f, k_cr, mem_f, mem_cr, H, n, F, CR);
  return last_launch_status();
}

constexpr int kSelectMaxBlocks = 16384;

EVOX_API int evox_hip_de_select_blocks(int n, int d) {
  const int lpr = d >= 48 ? 64 : (d >= 24 ? 32 : (d >= 12 ? 16 : 8));
  const int rpb = 256 / lpr;
  const int groups = (n + rpb - 1) / rpb;
  return groups < kSelectMaxBlocks ? groups : kSelectMaxBlocks;
}

// partials: evox_hip_de_select_blocks(n, d) × 6 floats
EVOX_API int evox_hip_de_select(const float* pop, const float* trial, const float* fit, const float* trial_fit,
                                const float* F, const float* CR, int strict, int weight_by_gain, int n, int d,
                                float* new_pop, float* new_fit, float* partials, void* stream) {
  if (n <= 0 || d <= 0) return (int)hipErrorInvalidValue;
  const int blocks = evox_hip_de_select_blocks(n, d);
  hipStream_t s = (hipStream_t)stream;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec = d % 4 == 0 && aligned(pop) && aligned(trial) && aligned(new_pop);
#define EVOX_SEL(L) de_select_kernel<L><<<blocks, 256, 0, s>>>(pop, trial, fit, trial_fit, F, CR, strict, weight_by_gain, \
                                                                 n, d, vec, new_pop, new_fit, partials)
  if (d >= 48) EVOX_SEL(64);
  else if (d >= 24) EVOX_SEL(32);
  else if (d >= 12) EVOX_SEL(16);
  else EVOX_SEL(8);
#undef EVOX_SEL
  return last_launch_status();
}

constexpr int kFoldMin = 4096;       // records beyond which a fold pass precedes the finalising workgroup
constexpr int kFoldGroups = 256;

// The records beyond kFoldMin are folded in place: partials is written as well as read.
EVOX_API int evox_hip_de_finalize(const float* partials, int nblocks, int mode, float c, const float* mem_f,
                                  const float* mem_cr, int H, float* out_f, float* out_cr, int64_t* best_index,
                                  void* stream) {
  if (nblocks <= 0 || H < 1 || (mode != 0 && mode != 1)) return (int)hipErrorInvalidValue;
  int stride = 1;
  if (nblocks > kFoldMin) {
    stride = (nblocks + kFoldGroups - 1) / kFoldGroups;
    const int groups = (nblocks + stride - 1) / stride;
    de_fold_kernel<<<groups, 256, 0, (hipStream_t)stream>>>(const_cast<float*>(partials), nblocks, stride);
  }
  de_finalize_kernel<<<1, 256, 0, (hipStream_t)stream>>>(partials, nblocks, stride, mode, c, mem_f, mem_cr, H, out_f,
                                                         out_cr, best_index);
  return last_launch_status();
}
