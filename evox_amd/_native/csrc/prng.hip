// Philox draw kernels for gfx950.  One thread = one counter block (4 outputs); the block
// index is taken in the GLOBAL element space (offset + i), so a shard of a logical tensor
// gets exactly the values the unsharded draw would have produced.
#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

enum DrawKind { kBits = 0, kUniform = 1, kNormal = 2, kRandint = 3 };

template <int KIND, typename OutT>
__global__ __launch_bounds__(256) void draw_kernel(const int32_t* __restrict__ key, OutT* __restrict__ out,
                                                    int64_t n, int64_t offset, float a, float b,
                                                    int64_t ilo, uint32_t irange) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const uint64_t first = (uint64_t)offset >> 2;
  const int64_t nblk = (int64_t)((((uint64_t)(offset + n - 1)) >> 2) - first + 1);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk;
       t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = first + (uint64_t)t;
    const u32x4 bits = philox_block(k0, k1, blk, kStreamDraw);
    OutT v[4];
    if constexpr (KIND == kBits) {
      v[0] = (OutT)bits.x; v[1] = (OutT)bits.y; v[2] = (OutT)bits.z; v[3] = (OutT)bits.w;
    } else if constexpr (KIND == kUniform) {
      const f32x4 u = unit4(bits);
      v[0] = fmaf(u.x, b - a, a); v[1] = fmaf(u.y, b - a, a);
      v[2] = fmaf(u.z, b - a, a); v[3] = fmaf(u.w, b - a, a);
    } else if constexpr (KIND == kNormal) {
      const f32x4 g = normal4(bits);
      v[0] = fmaf(g.x, b, a); v[1] = fmaf(g.y, b, a); v[2] = fmaf(g.z, b, a); v[3] = fmaf(g.w, b, a);
    } else {
      v[0] = (OutT)(ilo + (int64_t)bits_to_range(bits.x, irange));
      v[1] = (OutT)(ilo + (int64_t)bits_to_range(bits.y, irange));
      v[2] = (OutT)(ilo + (int64_t)bits_to_range(bits.z, irange));
      v[3] = (OutT)(ilo + (int64_t)bits_to_range(bits.w, irange));
    }
    const int64_t e0 = (int64_t)(blk * 4) - offset;
    if (e0 >= 0 && e0 + 3 < n && (offset & 3) == 0 && sizeof(OutT) == 4) {
      // aligned interior block: one 16-byte store
      struct alignas(16) Pack { OutT q[4]; };
      Pack p; p.q[0] = v[0]; p.q[1] = v[1]; p.q[2] = v[2]; p.q[3] = v[3];
      *reinterpret_cast<Pack*>(out + e0) = p;
    } else {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int64_t e = e0 + l;
        if (e >= 0 && e < n) out[e] = v[l];
      }
    }
  }
}

// Gaussian draws rounded (to nearest even) to bf16: the same values as the fp32 draw followed by a cast,
// written once at half the traffic.
__global__ __launch_bounds__(256) void normal_bf16_kernel(const int32_t* __restrict__ key, uint16_t* __restrict__ out,
                                                          int64_t n, int64_t offset, float a, float b) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const uint64_t first = (uint64_t)offset >> 2;
  const int64_t nblk = (int64_t)((((uint64_t)(offset + n - 1)) >> 2) - first + 1);
  auto cvt = [](float v) -> uint32_t {
    const uint32_t u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk;
       t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = first + (uint64_t)t;
    const f32x4 g = normal4(philox_block(k0, k1, blk, kStreamDraw));
    const uint32_t v[4] = {cvt(fmaf(g.x, b, a)), cvt(fmaf(g.y, b, a)), cvt(fmaf(g.z, b, a)), cvt(fmaf(g.w, b, a))};
    const int64_t e0 = (int64_t)(blk * 4) - offset;
    if (e0 >= 0 && e0 + 3 < n && (offset & 3) == 0) {
      *reinterpret_cast<uint2*>(out + e0) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    } else {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int64_t e = e0 + l;
        if (e >= 0 && e < n) out[e] = (uint16_t)v[l];
      }
    }
  }
}

__global__ void split_kernel(const int32_t* __restrict__ keys, int32_t* __restrict__ out, int64_t nkeys,
                             int64_t n) {
  const int64_t total = nkeys * n;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = t / n, i = t % n;
    const u32x4 b = philox_block((uint32_t)keys[2 * q], (uint32_t)keys[2 * q + 1], (uint64_t)i, kStreamSplit);
    out[2 * t] = (int32_t)b.x;
    out[2 * t + 1] = (int32_t)b.y;
  }
}

// split of ONE key into n keys, and a float counter advanced by one in the same launch (the generation counter of an
// algorithm whose ask() splits its key: one launch instead of two)
__global__ void split_count_kernel(const int32_t* __restrict__ key, int32_t* __restrict__ out, int64_t n,
                                   const float* __restrict__ count_in, float* __restrict__ count_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const u32x4 b = philox_block((uint32_t)key[0], (uint32_t)key[1], (uint64_t)t, kStreamSplit);
    out[2 * t] = (int32_t)b.x;
    out[2 * t + 1] = (int32_t)b.y;
  }
  if (t == 0 && count_out != nullptr) *count_out = *count_in + 1.0f;
}

__global__ void fold_kernel(const int32_t* __restrict__ keys, int32_t* __restrict__ out, int64_t nkeys,
                            int64_t data) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nkeys) {
    const u32x4 b = philox_block((uint32_t)keys[2 * q], (uint32_t)keys[2 * q + 1], (uint64_t)data, kStreamFold);
    out[2 * q] = (int32_t)b.x;
    out[2 * q + 1] = (int32_t)b.y;
  }
}

inline int64_t nblocks_of(int64_t n, int64_t offset) {
  return (int64_t)((((uint64_t)(offset + n - 1)) >> 2) - ((uint64_t)offset >> 2) + 1);
}

}  // namespace

EVOX_API int evox_hip_random_bits(const int32_t* key, uint32_t* out, int64_t n, int64_t offset, void* stream) {
  if (n <= 0) return 0;
  const int grid = stream_grid(nblocks_of(n, offset), 256);
  draw_kernel<kBits, uint32_t><<<grid, 256, 0, (hipStream_t)stream>>>(key, out, n, offset, 0.f, 0.f, 0, 0u);
  return last_launch_status();
}

EVOX_API int evox_hip_uniform(const int32_t* key, float* out, int64_t n, int64_t offset, float lo, float hi,
                              void* stream) {
  if (n <= 0) return 0;
  const int grid = stream_grid(nblocks_of(n, offset), 256);
  draw_kernel<kUniform, float><<<grid, 256, 0, (hipStream_t)stream>>>(key, out, n, offset, lo, hi, 0, 0u);
  return last_launch_status();
}

EVOX_API int evox_hip_normal(const int32_t* key, float* out, int64_t n, int64_t offset, float mean, float stddev,
                             void* stream) {
  if (n <= 0) return 0;
  const int grid = stream_grid(nblocks_of(n, offset), 256);
  draw_kernel<kNormal, float><<<grid, 256, 0, (hipStream_t)stream>>>(key, out, n, offset, mean, stddev, 0, 0u);
  return last_launch_status();
}

EVOX_API int evox_hip_normal_bf16(const int32_t* key, uint16_t* out, int64_t n, int64_t offset, float mean, float stddev,
                                  void* stream) {
  if (n <= 0) return 0;
  const int grid = stream_grid(nblocks_of(n, offset), 256);
  normal_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(key, out, n, offset, mean, stddev);
  return last_launch_status();
}

EVOX_API int evox_hip_randint(const int32_t* key, int64_t* out, int64_t n, int64_t offset, int64_t lo, int64_t hi,
                              void* stream) {
  if (n <= 0) return 0;
  const int grid = stream_grid(nblocks_of(n, offset), 256);
  draw_kernel<kRandint, int64_t><<<grid, 256, 0, (hipStream_t)stream>>>(key, out, n, offset, 0.f, 0.f, lo,
                                                                        (uint32_t)(hi - lo));
  return last_launch_status();
}

EVOX_API int evox_hip_split(const int32_t* keys, int32_t* out, int64_t nkeys, int64_t n, void* stream) {
  const int64_t total = nkeys * n;
  if (total <= 0) return 0;
  split_kernel<<<stream_grid(total, 64), 64, 0, (hipStream_t)stream>>>(keys, out, nkeys, n);
  return last_launch_status();
}

// Two levels of splitting in one launch: out[0, n1) = the n1 children of `key`, out[n1, n1 + n2) = the n2 children of
// child `which` — what split(key, n1) followed by split(children[which], n2) gives.
namespace {
__global__ void split_nested_kernel(const int32_t* __restrict__ key, int32_t* __restrict__ out, int n1, int which, int n2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n1 + n2) return;
  uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  uint64_t i = (uint64_t)t;
  if (t >= n1) {
    const u32x4 parent = philox_block(k0, k1, (uint64_t)which, kStreamSplit);
    k0 = parent.x; k1 = parent.y;
    i = (uint64_t)(t - n1);
  }
  const u32x4 b = philox_block(k0, k1, i, kStreamSplit);
  out[2 * t] = (int32_t)b.x;
  out[2 * t + 1] = (int32_t)b.y;
}
}  // namespace

EVOX_API int evox_hip_split_nested(const int32_t* key, int32_t* out, int n1, int which, int n2, void* stream) {
  if (n1 < 1 || n2 < 1 || which < 0 || which >= n1) return (int)hipErrorInvalidValue;
  split_nested_kernel<<<(n1 + n2 + 63) / 64, 64, 0, (hipStream_t)stream>>>(key, out, n1, which, n2);
  return last_launch_status();
}

EVOX_API int evox_hip_split_count(const int32_t* key, int32_t* out, int64_t n, const float* count_in, float* count_out,
                                  void* stream) {
  if (n < 0 || (n == 0 && count_out == nullptr)) return (int)hipErrorInvalidValue;     // n == 0: the counter alone
  split_count_kernel<<<n > 0 ? (int)((n + 63) / 64) : 1, 64, 0, (hipStream_t)stream>>>(key, out, n, count_in, count_out);
  return last_launch_status();
}

EVOX_API int evox_hip_fold_in(const int32_t* keys, int32_t* out, int64_t nkeys, int64_t data, void* stream) {
  if (nkeys <= 0) return 0;
  fold_kernel<<<(int)((nkeys + 63) / 64), 64, 0, (hipStream_t)stream>>>(keys, out, nkeys, data);
  return last_launch_status();
}
