// Weighted reductions over regenerated Gaussian noise:
//     g[j]  = sum_i w[i] * z(i, j)                 (ES gradient estimate)
//     h[j]  = sum_i v[i] * (z(i, j)^2 - 1)         (variance gradient of PGPE / SNES; optional)
// z(i, j) is element (row_offset + i) * dim + j of the Philox normal stream of `key`, i.e. the
// value the sampling step produced; nothing is read from memory except the two weight vectors.
// One thread owns four consecutive columns (one counter block when dim % 4 == 0) and walks a
// slice of the rows; blockIdx.y selects the slice, partial sums go to slab[slice][dim].
#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

template <bool WITH_V>
__global__ __launch_bounds__(256) void noise_wsum_kernel(const int32_t* __restrict__ key, const float* __restrict__ w,
                                                         const float* __restrict__ v, float* __restrict__ gslab,
                                                         float* __restrict__ hslab, int64_t n, int64_t dim,
                                                         int64_t row_offset, int splits) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // column quad
  const int64_t c0 = q * 4;
  if (c0 >= dim) return;
  const int64_t rows_per = (n + splits - 1) / splits;
  const int64_t rbeg = (int64_t)blockIdx.y * rows_per;
  const int64_t rend = min(n, rbeg + rows_per);
  float g[4] = {0.f, 0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
  const bool aligned = (dim & 3) == 0;
  for (int64_t i = rbeg; i < rend; ++i) {
    const float wi = w[i];
    const float vi = WITH_V ? v[i] : 0.f;
    const uint64_t e = (uint64_t)(row_offset + i) * (uint64_t)dim + (uint64_t)c0;
    float z[4];
    const f32x4 a = normal4(philox_block(k0, k1, e >> 2, kStreamDraw));
    if (aligned) {
      z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w;
    } else {
      const int sh = (int)(e & 3);
      const f32x4 b = normal4(philox_block(k0, k1, (e >> 2) + 1, kStreamDraw));
#pragma unroll
      for (int l = 0; l < 4; ++l) z[l] = (l + sh < 4) ? lane_of(a, l + sh) : lane_of(b, l + sh - 4);
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      g[l] = fmaf(wi, z[l], g[l]);
      if (WITH_V) h[l] = fmaf(vi, fmaf(z[l], z[l], -1.0f), h[l]);
    }
  }
  float* go = gslab + (int64_t)blockIdx.y * dim + c0;
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (c0 + l < dim) go[l] = g[l];
  if (WITH_V) {
    float* ho = hslab + (int64_t)blockIdx.y * dim + c0;
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (c0 + l < dim) ho[l] = h[l];
  }
}

// out[r, c] = center[c] + sign(r) * scale[c] * z(noise_row(r), c): the population of an evolution
// strategy written in one pass (no noise tensor, no temporaries), optionally as bf16 and with a
// padded row pitch.  With half > 0 the logical population is [+z; -z] stacked (mirrored sampling):
// logical row g >= half reuses noise row g - half with the opposite sign.
template <typename T>
__global__ __launch_bounds__(256) void es_perturb_kernel(const int32_t* __restrict__ key, const float* __restrict__ center,
                                                         const float* __restrict__ scale, int scale_is_vec,
                                                         T* __restrict__ out, int64_t pitch, int64_t n_rows, int64_t dim,
                                                         int64_t row_offset, int64_t half) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const int64_t c0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c0 >= dim) return;
  float ctr[4], sc[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int64_t c = c0 + l < dim ? c0 + l : dim - 1;
    ctr[l] = center[c];
    sc[l] = scale_is_vec ? scale[c] : scale[0];
  }
  const bool aligned = (dim & 3) == 0;
  for (int64_t r = blockIdx.y; r < n_rows; r += gridDim.y) {
    const int64_t g = row_offset + r;
    const bool neg = half > 0 && g >= half;
    const int64_t nr = neg ? g - half : g;
    const uint64_t e = (uint64_t)nr * (uint64_t)dim + (uint64_t)c0;
    float z[4];
    const f32x4 a = normal4(philox_block(k0, k1, e >> 2, kStreamDraw));
    if (aligned) {
      z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w;
    } else {
      const int sh = (int)(e & 3);
      const f32x4 b = normal4(philox_block(k0, k1, (e >> 2) + 1, kStreamDraw));
#pragma unroll
      for (int l = 0; l < 4; ++l) z[l] = (l + sh < 4) ? lane_of(a, l + sh) : lane_of(b, l + sh - 4);
    }
    T* o = out + r * pitch + c0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (c0 + l < dim) {
        // same rounding as center + scale * z / center - scale * z in two fp32 steps
        const float v = neg ? ctr[l] - sc[l] * z[l] : ctr[l] + sc[l] * z[l];
        if constexpr (sizeof(T) == 2) o[l] = (T)((__float_as_uint(v) + 0x7fffu + ((__float_as_uint(v) >> 16) & 1u)) >> 16);
        else o[l] = v;
      }
    }
  }
}

// The same population from the generator's point of view: a thread owns ONE Philox block (four
// consecutive elements of the flat noise stream, which may straddle a row boundary when dim is not a
// multiple of four) and writes them wherever they belong — and, for a mirrored population held in one
// tensor, their negated twins half rows further down.  Every normal is computed exactly once; the
// column-quad kernel above computes each up to four times (two blocks per unaligned quad, and again
// for the mirrored row), which made it generator-bound at 1.4 TB/s of output.
// Noise rows [nr0, nr0 + R); out row of noise row nr: nr − nr0 (and nr − nr0 + R when MIRROR).
// VEC: dim and pitch are multiples of four and all operands aligned — a block of four normals lies inside one row and
// is written (and its centre read) with one 16-byte access (8 bytes as bf16) instead of four.
__device__ __forceinline__ uint32_t bf16_bits(float v) {
  return (__float_as_uint(v) + 0x7fffu + ((__float_as_uint(v) >> 16) & 1u)) >> 16;
}
template <typename T>
__device__ __forceinline__ void store_quad(T* o, float a, float b, float c, float d) {
  if constexpr (sizeof(T) == 2) {
    uint2 w;
    w.x = bf16_bits(a) | (bf16_bits(b) << 16);
    w.y = bf16_bits(c) | (bf16_bits(d) << 16);
    *reinterpret_cast<uint2*>(o) = w;
  } else {
    *reinterpret_cast<float4*>(o) = make_float4(a, b, c, d);
  }
}

template <typename T, bool MIRROR, bool VEC>
__global__ __launch_bounds__(256) void es_perturb_flat_kernel(const int32_t* __restrict__ key,
                                                              const float* __restrict__ center,
                                                              const float* __restrict__ scale, int scale_is_vec,
                                                              T* __restrict__ out, int64_t pitch, int64_t nr0, int64_t R,
                                                              int64_t dim) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const uint64_t first = (uint64_t)nr0 * (uint64_t)dim, last = (uint64_t)(nr0 + R) * (uint64_t)dim;   // [first, last)
  const uint64_t b_lo = first >> 2, b_hi = (last + 3) >> 2;
  const float s0 = scale[0];
  if constexpr (VEC) {
    const bool narrow = last < ((uint64_t)1 << 32);
    for (uint64_t b = b_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < b_hi; b += (uint64_t)gridDim.x * blockDim.x) {
      const f32x4 z = normal4(philox_block(k0, k1, b, kStreamDraw));
      const uint64_t e = b << 2;
      const int64_t nr = narrow ? (int64_t)((uint32_t)e / (uint32_t)dim) : (int64_t)(e / (uint64_t)dim);
      const int64_t c = (int64_t)(e - (uint64_t)nr * (uint64_t)dim);
      const float4 ctr = *reinterpret_cast<const float4*>(center + c);
      float4 sc = make_float4(s0, s0, s0, s0);
      if (scale_is_vec) sc = *reinterpret_cast<const float4*>(scale + c);
      T* o = out + (nr - nr0) * pitch + c;
      store_quad<T>(o, ctr.x + sc.x * z.x, ctr.y + sc.y * z.y, ctr.z + sc.z * z.z, ctr.w + sc.w * z.w);
      if (MIRROR)
        store_quad<T>(o + R * pitch, ctr.x - sc.x * z.x, ctr.y - sc.y * z.y, ctr.z - sc.z * z.z, ctr.w - sc.w * z.w);
    }
    return;
  }
  // Any dimension: a lane still generates one block of four normals, but the workgroup's 1024 values are turned round in
  // LDS so that consecutive lanes write consecutive elements (four stores of 256 contiguous bytes per wavefront instead
  // of four stores that each touch every fourth word of a kilobyte).  Every thread of the workgroup takes part in the
  // barriers: the loop's trip count depends on the workgroup alone.
  __shared__ float tile[1024];
  const bool narrow = last < ((uint64_t)1 << 32);
  for (uint64_t base = b_lo + (uint64_t)blockIdx.x * 256; base < b_hi; base += (uint64_t)gridDim.x * 256) {
    const uint64_t b = base + threadIdx.x;
    f32x4 z4{0.f, 0.f, 0.f, 0.f};
    if (b < b_hi) z4 = normal4(philox_block(k0, k1, b, kStreamDraw));
    *reinterpret_cast<float4*>(&tile[4 * threadIdx.x]) = make_float4(z4.x, z4.y, z4.z, z4.w);
    __syncthreads();
    uint64_t e = (base << 2) + threadIdx.x;
    int64_t nr = narrow ? (int64_t)((uint32_t)e / (uint32_t)dim) : (int64_t)(e / (uint64_t)dim);
    int64_t c = (int64_t)(e - (uint64_t)nr * (uint64_t)dim);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (e >= first && e < last) {
        const float z = tile[j * 256 + threadIdx.x];
        const float ctr = center[c], sc = scale_is_vec ? scale[c] : s0;
        const float up = ctr + sc * z;
        T* o = out + (nr - nr0) * pitch + c;
        if constexpr (sizeof(T) == 2) o[0] = (T)bf16_bits(up);
        else o[0] = up;
        if (MIRROR) {
          const float dn = ctr - sc * z;
          T* o2 = o + R * pitch;
          if constexpr (sizeof(T) == 2) o2[0] = (T)bf16_bits(dn);
          else o2[0] = dn;
        }
      }
      e += 256;
      c += 256;
      while (c >= dim) { c -= dim; ++nr; }
    }
    __syncthreads();
  }
}

}  // namespace

EVOX_API int evox_hip_es_perturb(const int32_t* key, const float* center, const float* scale, int scale_is_vec, void* out,
                                 int out_dtype, int64_t pitch, int64_t n_rows, int64_t dim, int64_t row_offset,
                                 int64_t half, void* stream) {
  if (n_rows <= 0 || dim <= 0) return 0;
  if (pitch < dim) return (int)hipErrorInvalidValue;
  {
    // windows that map to one contiguous range of noise rows with one sign: all-plus windows, and the
    // whole mirrored population [+z; −z] in one tensor
    const bool whole_mirror = half > 0 && row_offset == 0 && n_rows == 2 * half;
    const bool plain = half == 0 || row_offset + n_rows <= half;
    if (whole_mirror || plain) {
      const int64_t R = whole_mirror ? half : n_rows;
      const int64_t blocks4 = (R * dim + 3) / 4 + 1;
      const int grid1 = (int)((blocks4 + 255) / 256 < 16384 ? (blocks4 + 255) / 256 : 16384);
      hipStream_t s = (hipStream_t)stream;
      auto aligned = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
      const bool vec = dim % 4 == 0 && pitch % 4 == 0 && aligned(center, 16) && (!scale_is_vec || aligned(scale, 16)) &&
                       aligned(out, out_dtype == 1 ? 8 : 16);
#define EVOX_FLAT(T, M, V) es_perturb_flat_kernel<T, M, V><<<grid1, 256, 0, s>>>(key, center, scale, scale_is_vec, (T*)out, \
                                                                                pitch, row_offset, R, dim)
#define EVOX_FLAT_T(T)                          \
  do {                                          \
    if (whole_mirror) {                         \
      if (vec) EVOX_FLAT(T, true, true);        \
      else EVOX_FLAT(T, true, false);           \
    } else {                                    \
      if (vec) EVOX_FLAT(T, false, true);       \
      else EVOX_FLAT(T, false, false);          \
    }                                           \
  } while (0)
      if (out_dtype == 1) EVOX_FLAT_T(uint16_t);
      else EVOX_FLAT_T(float);
#undef EVOX_FLAT_T
#undef EVOX_FLAT
      return last_launch_status();
    }
  }
  const int64_t quads = (dim + 3) / 4;
  dim3 grid((unsigned)((quads + 255) / 256), (unsigned)(n_rows < 4096 ? n_rows : 4096));
  if (out_dtype == 1)
    es_perturb_kernel<uint16_t><<<grid, 256, 0, (hipStream_t)stream>>>(key, center, scale, scale_is_vec, (uint16_t*)out,
                                                                        pitch, n_rows, dim, row_offset, half);
  else
    es_perturb_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(key, center, scale, scale_is_vec, (float*)out, pitch,
                                                                     n_rows, dim, row_offset, half);
  return last_launch_status();
}

// [center + scale·z; center − scale·z] for the noise rows [noise_start, noise_start + count): a rank's
// share of a mirrored population with both members of every pair on the same rank.  out: 2·count rows.
EVOX_API int evox_hip_es_perturb_pairs(const int32_t* key, const float* center, const float* scale, int scale_is_vec,
                                       void* out, int out_dtype, int64_t pitch, int64_t noise_start, int64_t count,
                                       int64_t dim, void* stream) {
  if (count <= 0 || dim <= 0) return 0;
  if (pitch < dim || noise_start < 0) return (int)hipErrorInvalidValue;
  const int64_t blocks4 = (count * dim + 3) / 4 + 1;
  const int grid1 = (int)((blocks4 + 255) / 256 < 16384 ? (blocks4 + 255) / 256 : 16384);
  hipStream_t s = (hipStream_t)stream;
  auto aligned = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  const bool vec = dim % 4 == 0 && pitch % 4 == 0 && aligned(center, 16) && (!scale_is_vec || aligned(scale, 16)) &&
                   aligned(out, out_dtype == 1 ? 8 : 16);
  if (out_dtype == 1) {
    if (vec) es_perturb_flat_kernel<uint16_t, true, true><<<grid1, 256, 0, s>>>(key, center, scale, scale_is_vec, (uint16_t*)out, pitch, noise_start, count, dim);
    else es_perturb_flat_kernel<uint16_t, true, false><<<grid1, 256, 0, s>>>(key, center, scale, scale_is_vec, (uint16_t*)out, pitch, noise_start, count, dim);
  } else {
    if (vec) es_perturb_flat_kernel<float, true, true><<<grid1, 256, 0, s>>>(key, center, scale, scale_is_vec, (float*)out, pitch, noise_start, count, dim);
    else es_perturb_flat_kernel<float, true, false><<<grid1, 256, 0, s>>>(key, center, scale, scale_is_vec, (float*)out, pitch, noise_start, count, dim);
  }
  return last_launch_status();
}

EVOX_API int evox_hip_noise_wsum(const int32_t* key, const float* w, const float* v, float* gslab, float* hslab,
                                 void* reserved, int64_t n, int64_t dim, int64_t row_offset, int64_t splits, int flags,
                                 void* stream) {
  (void)reserved; (void)flags;
  if (n <= 0 || dim <= 0) return 0;
  const int64_t quads = (dim + 3) / 4;
  dim3 grid((unsigned)((quads + 255) / 256), (unsigned)splits);
  if (v != nullptr)
    noise_wsum_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(key, w, v, gslab, hslab, n, dim, row_offset, (int)splits);
  else
    noise_wsum_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(key, w, v, gslab, hslab, n, dim, row_offset, (int)splits);
  return last_launch_status();
}
