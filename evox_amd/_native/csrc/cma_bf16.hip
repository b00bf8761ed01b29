// CMA-ES hot path on bf16 matrix cores: entry points built on the hand-written NT GEMM of bf16_gemm.h.
//
//   evox_hip_normal_bf16_pitched  Z[r][c] = bf16(N(0,1) draw number (row_offset + r) * n + c) into a row-padded buffer
//   evox_hip_bf16_sample          X = mean + sigma * Z * S^T            (reference: es_variants/cma_es.py:128-135)
//   evox_hip_cma_gather_t         Yt[c][i] = w_i^(1/2) (x_{order[i]} - mean)[c] / sigma, plus the weighted mean shift
//   evox_hip_bf16_cov             C' = alpha C + c1 pc pc^T + cmu Yt Yt^T  (reference: cma_es.py:178-184), upper
//                                 triangle only on the matrix cores, K split over workgroups, mirrored on write
//
// Row pitches of every bf16 operand are multiples of 64 elements; pad columns must hold zeros (the callers
// allocate the buffers zero-filled once and the kernels never write the pad).
#include "bf16_gemm.h"
#include "philox.h"

using namespace evox;
using namespace evox::bg;

namespace {

__global__ __launch_bounds__(256) void normal_bf16_pitched_kernel(const int32_t* __restrict__ key, uint16_t* __restrict__ out,
                                                                  int64_t rows, int64_t n, int64_t ld, int64_t row_offset) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const uint64_t first_e = (uint64_t)row_offset * (uint64_t)n, last_e = first_e + (uint64_t)rows * (uint64_t)n;   // [first, last)
  const uint64_t first_b = first_e >> 2;
  const int64_t nblk = (int64_t)(((last_e - 1) >> 2) - first_b + 1);
  const bool fast = (n & 3) == 0 && (ld & 3) == 0;     // a counter block never straddles rows, stores are 8-byte aligned
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblk; t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = first_b + (uint64_t)t;
    const f32x4 g = normal4(philox_block(k0, k1, blk, kStreamDraw));
    const uint32_t v[4] = {f32_to_bf16_bits(g.x), f32_to_bf16_bits(g.y), f32_to_bf16_bits(g.z), f32_to_bf16_bits(g.w)};
    const uint64_t e0 = blk * 4;
    if (fast) {
      const uint64_t r = e0 / (uint64_t)n, c = e0 - r * (uint64_t)n;
      *reinterpret_cast<uint2*>(out + (r - (uint64_t)row_offset) * ld + c) = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
    } else {
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const uint64_t e = e0 + l;
        if (e >= first_e && e < last_e) {
          const uint64_t r = e / (uint64_t)n, c = e - r * (uint64_t)n;
          out[(r - (uint64_t)row_offset) * ld + c] = (uint16_t)v[l];
        }
      }
    }
  }
}

// The same draws for rows whose length is a multiple of 4 (a counter block never straddles rows): one wavefront per
// row, lanes over the row's counter blocks — no 64-bit division per block, one 8-byte store per lane and block.
__global__ __launch_bounds__(256) void normal_bf16_rows_kernel(const int32_t* __restrict__ key, uint16_t* __restrict__ out,
                                                               int64_t rows, int n, int64_t ld, int64_t row_offset) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const int lane = threadIdx.x & 63;
  const int bpr = n >> 2;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const uint64_t b0 = (uint64_t)(row_offset + r) * (uint64_t)bpr;
    uint16_t* orow = out + r * ld;
    for (int cb = lane; cb < bpr; cb += 64) {
      const f32x4 g = normal4(philox_block(k0, k1, b0 + (uint64_t)cb, kStreamDraw));
      const uint32_t lo = f32_to_bf16_bits(g.x) | (f32_to_bf16_bits(g.y) << 16);
      const uint32_t hi = f32_to_bf16_bits(g.z) | (f32_to_bf16_bits(g.w) << 16);
      *reinterpret_cast<uint2*>(orow + 4 * cb) = make_uint2(lo, hi);
    }
  }
}

// Gather + centre + scale + transpose.  Block (ct, rb): columns [64 ct, 64 ct + 64), selected individuals in tiles of
// 64 (tile rb, rb + gridDim.y, ...).  A tile is read row-wise with 16-byte loads (every lane has its four loads of a
// tile in flight together) and written column-wise (128 contiguous bytes per column) through LDS.
__global__ __launch_bounds__(256) void gather_t_kernel(const float* __restrict__ pop, int64_t ldpop,
                                                       const int64_t* __restrict__ order, const float* __restrict__ mean,
                                                       const float* __restrict__ rowscale, const float* __restrict__ sigma,
                                                       uint16_t* __restrict__ yt, int64_t ldy, float* __restrict__ colsum,
                                                       int mu, int n, int lam) {
  __shared__ float tile[64][65];
  __shared__ float part[16][64];
  const float inv_sigma = 1.0f / sigma[0];
  const int c0 = blockIdx.x * 64;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;      // column quad, row group (16 rows per pass)
  const int c = c0 + 4 * q;
  const bool vec = (ldpop & 3) == 0 && (((uintptr_t)pop | (uintptr_t)mean) & 15) == 0 && c + 3 < n;
  float m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = c + k < n ? mean[c + k] : 0.f;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int tiles = (mu + 63) / 64;
  for (int rt = blockIdx.y; rt < tiles; rt += gridDim.y) {
    const int i0 = rt * 64;
    int64_t src[4];
    float w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int i = i0 + rg + 16 * p;
      int64_t sidx = i < mu ? order[i] : 0;
      src[p] = sidx < 0 ? 0 : (sidx >= lam ? lam - 1 : sidx);      // a corrupt index must not leave the population
      w[p] = i < mu ? rowscale[i] : 0.f;
    }
    float4 x[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float* rp = pop + src[p] * ldpop + c;
      if (vec) {
        x[p] = *reinterpret_cast<const float4*>(rp);
      } else {
        x[p] = make_float4(c < n ? rp[0] : 0.f, c + 1 < n ? rp[1] : 0.f, c + 2 < n ? rp[2] : 0.f, c + 3 < n ? rp[3] : 0.f);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int il = rg + 16 * p;
      const float xs[4] = {x[p].x, x[p].y, x[p].z, x[p].w};
      const float sc = w[p] * inv_sigma, w2 = w[p] * w[p];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = (c + k < n) ? xs[k] - m[k] : 0.f;
        tile[il][4 * q + k] = d * sc;            // zero beyond mu: w = 0
        acc[k] = fmaf(w2, d, acc[k]);
      }
    }
    __syncthreads();
    {
      // thread -> column cl = t / 4, individuals 16 (t % 4) ... + 16 of the tile: 32 contiguous bytes
      const int cl = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;
      if (c0 + cl < n) {
        uint32_t wv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          wv[k] = f32_to_bf16_bits(tile[seg + 2 * k][cl]) | (f32_to_bf16_bits(tile[seg + 2 * k + 1][cl]) << 16);
        uint16_t* dst = yt + (int64_t)(c0 + cl) * ldy + i0 + seg;
        *reinterpret_cast<uint4*>(dst) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
        *reinterpret_cast<uint4*>(dst + 8) = make_uint4(wv[4], wv[5], wv[6], wv[7]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[rg][4 * q + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + (int)threadIdx.x < n) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += part[g][threadIdx.x];
    colsum[(int64_t)blockIdx.y * n + c0 + threadIdx.x] = s;
  }
}

// step[c] = sum_b colsum[b][c] in a fixed order: 64 columns per block, four row groups, then LDS
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ colsum, int rb, int n, float* __restrict__ step) {
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  float s = 0.f;
  if (c < n) {
#pragma unroll 8
    for (int b = grp; b < rb; b += 4) s += colsum[(int64_t)b * n + c];
  }
  part[grp][col] = s;
  __syncthreads();
  if (grp == 0 && c < n) step[c] = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}

// out = alpha C + c1 pc pc^T + cmu * sum_s slab_s on 64 x 64 tiles (i <= j), written to (m, n) and (n, m).
// 16-byte loads, every slab of an element group in flight together (the pass is bound by memory latency otherwise).
template <int MAXS>
__global__ __launch_bounds__(256) void cov_reduce_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ C,
                                                         const float* __restrict__ pc, const float* __restrict__ alpha, float c1,
                                                         float cmu, float* __restrict__ out, int n, int tiles) {
  __shared__ float tr[64][65];
  int i = 0, rem = blockIdx.x;
  while (rem >= tiles - i) {
    rem -= tiles - i;
    ++i;
  }
  const int j = i + rem;
  const int r0 = i * 64, c0 = j * 64;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const float a = alpha[0];
  const int64_t slab = (int64_t)n * n;
  const int cc = c0 + 4 * q;
  float pcc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pcc[k] = cc + k < n ? pc[cc + k] : 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int rl = rg + 16 * p, r = r0 + rl;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n && cc < n) {                          // n is a multiple of 4: the quad is inside or outside as a whole
      const int64_t e = (int64_t)r * n + cc;
      float4 g[MAXS];
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
        g[s] = s < splits ? *reinterpret_cast<const float4*>(ws + s * slab + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = MAXS; s < splits; ++s) {          // more slabs than the unrolled batch
        const float4 t = *reinterpret_cast<const float4*>(ws + s * slab + e);
        g[0].x += t.x; g[0].y += t.y; g[0].z += t.z; g[0].w += t.w;
      }
      // alpha = 0 (a plain Gram product): C is not read at all — it may be the uninitialised output buffer itself
      const float4 cv = a != 0.f ? *reinterpret_cast<const float4*>(C + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 sum = g[0];
#pragma unroll
      for (int s = 1; s < MAXS; ++s) { sum.x += g[s].x; sum.y += g[s].y; sum.z += g[s].z; sum.w += g[s].w; }
      const float pr = pc[r];
      v[0] = fmaf(a, cv.x, fmaf(c1, pr * pcc[0], cmu * sum.x));
      v[1] = fmaf(a, cv.y, fmaf(c1, pr * pcc[1], cmu * sum.y));
      v[2] = fmaf(a, cv.z, fmaf(c1, pr * pcc[2], cmu * sum.z));
      v[3] = fmaf(a, cv.w, fmaf(c1, pr * pcc[3], cmu * sum.w));
      if (i != j) *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tr[rl][4 * q + k] = v[k];
  }
  __syncthreads();
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (i == j) {
    // a diagonal tile takes both halves from its upper triangle: the result is symmetric to the last bit
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int rl = rr * 4 + grp, r = r0 + rl;
      if (r < n && c0 + col < n) out[(int64_t)r * n + c0 + col] = rl <= col ? tr[rl][col] : tr[col][rl];
    }
    return;
  }
#pragma unroll 4
  for (int rr = 0; rr < 16; ++rr) {
    const int rl = rr * 4 + grp;                   // row of the mirrored tile = column of the computed one
    const int r = c0 + rl, c = r0 + col;
    if (r < n && c < n) out[(int64_t)r * n + c] = tr[col][rl];
  }
}

// Tile-packed upper triangle of a symmetric n x n matrix: the 64 x 64 tiles (i <= j) in the order of upper_tiles(), each
// stored contiguously (row-major, zero beyond n).  It is what a multi-process job sums across ranks instead of the full
// square: T(T+1)/2 tiles of 4096 floats — 0.56 n² at n = 1000.
// gram_pack_kernel: tile = sum of the K-split slabs of the Gram product (same element order of summation as
// cov_reduce_kernel).
template <int MAXS>
__global__ __launch_bounds__(256) void gram_pack_kernel(const float* __restrict__ ws, int splits, float* __restrict__ packed,
                                                        int n, int tiles) {
  int i = 0, rem = blockIdx.x;
  while (rem >= tiles - i) {
    rem -= tiles - i;
    ++i;
  }
  const int j = i + rem;
  const int r0 = i * 64, c0 = j * 64;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int64_t slab = (int64_t)n * n;
  const int cc = c0 + 4 * q;
  float* tile = packed + (int64_t)blockIdx.x * 4096;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int rl = rg + 16 * p, r = r0 + rl;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n && cc < n) {
      const int64_t e = (int64_t)r * n + cc;
      float4 g[MAXS];
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
        g[s] = s < splits ? *reinterpret_cast<const float4*>(ws + s * slab + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = MAXS; s < splits; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(ws + s * slab + e);
        g[0].x += t.x; g[0].y += t.y; g[0].z += t.z; g[0].w += t.w;
      }
      sum = g[0];
#pragma unroll
      for (int s = 1; s < MAXS; ++s) { sum.x += g[s].x; sum.y += g[s].y; sum.z += g[s].z; sum.w += g[s].w; }
    }
    *reinterpret_cast<float4*>(tile + rl * 64 + 4 * q) = sum;
  }
}

// out (n x n, both triangles, symmetric to the last bit) = alpha C + c1 pc pc^T + cmu G, G given tile-packed.  `poison`
// (optional, one float): a positive value turns the result into NaNs.
__global__ __launch_bounds__(256) void cov_from_packed_kernel(const float* __restrict__ packed, const float* __restrict__ C,
                                                              const float* __restrict__ pc, const float* __restrict__ alpha,
                                                              float c1, float cmu, const float* __restrict__ poison,
                                                              float* __restrict__ out, int n, int tiles) {
  __shared__ float tr[64][65];
  int i = 0, rem = blockIdx.x;
  while (rem >= tiles - i) {
    rem -= tiles - i;
    ++i;
  }
  const int j = i + rem;
  const int r0 = i * 64, c0 = j * 64;
  const int q = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const float a = alpha[0];
  const bool bad = poison != nullptr && *poison > 0.f;
  const float* tile = packed + (int64_t)blockIdx.x * 4096;
  const int cc = c0 + 4 * q;
  float pcc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pcc[k] = cc + k < n ? pc[cc + k] : 0.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int rl = rg + 16 * p, r = r0 + rl;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < n && cc < n) {
      const int64_t e = (int64_t)r * n + cc;
      const float4 gsum = *reinterpret_cast<const float4*>(tile + rl * 64 + 4 * q);
      const float4 cv = *reinterpret_cast<const float4*>(C + e);
      const float pr = pc[r];
      v[0] = fmaf(a, cv.x, fmaf(c1, pr * pcc[0], cmu * gsum.x));
      v[1] = fmaf(a, cv.y, fmaf(c1, pr * pcc[1], cmu * gsum.y));
      v[2] = fmaf(a, cv.z, fmaf(c1, pr * pcc[2], cmu * gsum.z));
      v[3] = fmaf(a, cv.w, fmaf(c1, pr * pcc[3], cmu * gsum.w));
      if (bad) v[0] = v[1] = v[2] = v[3] = __builtin_nanf("");
      if (i != j) *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tr[rl][4 * q + k] = v[k];
  }
  __syncthreads();
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  if (i == j) {
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
      const int rl = rr * 4 + grp, r = r0 + rl;
      if (r < n && c0 + col < n) out[(int64_t)r * n + c0 + col] = rl <= col ? tr[rl][col] : tr[col][rl];
    }
    return;
  }
#pragma unroll 4
  for (int rr = 0; rr < 16; ++rr) {
    const int rl = rr * 4 + grp;
    const int r = c0 + rl, c = r0 + col;
    if (r < n && c < n) out[(int64_t)r * n + c] = tr[col][rl];
  }
}

inline int ksteps_of(int64_t k) { return (int)((k + 63) / 64); }

}  // namespace

EVOX_API int evox_hip_normal_bf16_pitched(const int32_t* key, void* out, int64_t rows, int64_t n, int64_t ld,
                                          int64_t row_offset, void* stream) {
  if (rows <= 0 || n <= 0) return 0;
  if (ld < n) return (int)hipErrorInvalidValue;
  if ((n & 3) == 0 && (ld & 3) == 0 && n < (1ll << 31) && ((uintptr_t)out & 7) == 0) {
    int64_t blocks = (rows + 3) / 4;                       // four rows (wavefronts) per workgroup
    if (blocks > kNumCU * 16) blocks = kNumCU * 16;
    normal_bf16_rows_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(key, (uint16_t*)out, rows, (int)n, ld, row_offset);
    return last_launch_status();
  }
  const uint64_t first_e = (uint64_t)row_offset * (uint64_t)n, last_e = first_e + (uint64_t)rows * (uint64_t)n;
  const int64_t nblk = (int64_t)(((last_e - 1) >> 2) - (first_e >> 2) + 1);
  normal_bf16_pitched_kernel<<<stream_grid(nblk, 256), 256, 0, (hipStream_t)stream>>>(key, (uint16_t*)out, rows, n, ld, row_offset);
  return last_launch_status();
}

// X[M][N] (fp32, pitch ldx) = mean + sigma * Z[M][K] S[N][K]^T.  ldz, lds: multiples of 64 elements >= K rounded up
// to 64; pad columns zero.  X and mean 16-byte aligned, ldx a multiple of 4.
EVOX_API int evox_hip_bf16_sample(const void* Z, int64_t ldz, const void* S, int64_t lds, const float* mean,
                                  const float* sigma, float* X, int64_t ldx, int M, int N, int K, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  const int ks = ksteps_of(K);
  if (ldz < (int64_t)ks * 64 || lds < (int64_t)ks * 64 || (ldz & 7) || (lds & 7) || (ldx & 3) || ldx < N ||
      ((uintptr_t)X & 15) || ((uintptr_t)mean & 15) || ((uintptr_t)Z & 15) || ((uintptr_t)S & 15))
    return (int)hipErrorInvalidValue;
  Args<1> g{};
  g.a[0][0] = (const uint16_t*)Z;
  g.b[0][0] = (const uint16_t*)S;
  g.lda = ldz;
  g.ldb = lds;
  g.M = M;
  g.N = N;
  g.ksteps = ks;
  g.ksteps_per_split = ks;
  EpiSample epi{X, mean, sigma, (long long)ldx, M, N};
  hipStream_t st = (hipStream_t)stream;
  // 160 x 256 tiles: at lambda = 10^4, n = 10^3 that is 63 x 4 = 252 workgroups, one round on the 256 CUs
  const int64_t big = (int64_t)((M + 159) / 160) * ((N + 255) / 256);
  if (big >= 128) {
    g.tiles_m = (M + 159) / 160;
    g.tiles_n = (N + 255) / 256;
    return launch<160, 256, 2, 4, 1, false, 3, EpiSample>(g, epi, g.tiles_m * g.tiles_n, 1, 1, st);
  }
  g.tiles_m = (M + 63) / 64;
  g.tiles_n = (N + 63) / 64;
  return launch<64, 64, 2, 2, 1, false, 2, EpiSample>(g, epi, g.tiles_m * g.tiles_n, 1, 1, st);
}

// rb: number of row blocks of the partial mean shift (colsum is rb x n); step = their sum in a fixed order.
EVOX_API int evox_hip_cma_gather_t(const float* pop, int64_t ldpop, const int64_t* order, const float* mean,
                                   const float* rowscale, const float* sigma, void* yt, int64_t ldy, float* colsum, int rb,
                                   float* step, int mu, int n, int lam, void* stream) {
  if (mu <= 0 || n <= 0 || rb <= 0 || lam <= 0) return (int)hipErrorInvalidValue;
  if (ldy < (int64_t)ksteps_of(mu) * 64 || (ldy & 7) || ((uintptr_t)yt & 15)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  gather_t_kernel<<<dim3((n + 63) / 64, rb), 256, 0, st>>>(pop, ldpop, order, mean, rowscale, sigma, (uint16_t*)yt, ldy,
                                                          colsum, mu, n, lam);
  int e = last_launch_status();
  if (e || step == nullptr) return e;              // step == nullptr: the caller adds the partial sums up itself
  colsum_reduce_kernel<<<(n + 63) / 64, 256, 0, st>>>(colsum, rb, n, step);
  return last_launch_status();
}

EVOX_API int evox_hip_bf16_cov_splits(int n, int mu) {
  // enough K splits that (upper 128 x 128 tiles) x splits covers the 256 CUs about once
  const int T = (n + 127) / 128, tiles = T * (T + 1) / 2, ks = ksteps_of(mu);
  int want = (kNumCU + tiles - 1) / tiles;
  if (want > ks) want = ks;
  if (want < 1) want = 1;
  const int per = (ks + want - 1) / want;
  return (ks + per - 1) / per;
}

static int gram_slabs(const void* Yt, int64_t ldy, int n, int mu, float* ws, int64_t ws_floats, int* splits_out,
                      hipStream_t st) {
  if (n <= 0 || mu <= 0) return (int)hipErrorInvalidValue;
  const int ks = ksteps_of(mu);
  const int splits = evox_hip_bf16_cov_splits(n, mu);
  const int per = (ks + splits - 1) / splits;
  if (ldy < (int64_t)ks * 64 || (ldy & 7) || ((uintptr_t)Yt & 15) || (n & 3) || ((uintptr_t)ws & 15) ||
      ws_floats < (int64_t)splits * n * n)
    return (int)hipErrorInvalidValue;
  const int T = (n + 127) / 128;
  Args<1> g{};
  g.a[0][0] = g.b[0][0] = (const uint16_t*)Yt;
  g.lda = g.ldb = ldy;
  g.M = g.N = n;
  g.ksteps = ks;
  g.ksteps_per_split = per;
  g.tiles_m = g.tiles_n = T;
  EpiSlab epi{ws, (long long)n, (long long)n * n, n, n, splits};
  *splits_out = splits;
  return launch<128, 128, 2, 4, 1, true, 3, EpiSlab>(g, epi, upper_tiles(T), splits, 1, st);
}

// Floats of the tile-packed upper triangle of an n x n matrix (see gram_pack_kernel).
EVOX_API int64_t evox_hip_packed_upper_floats(int n) {
  const int T64 = (n + 63) / 64;
  return (int64_t)upper_tiles(T64) * 4096;
}

// packed (evox_hip_packed_upper_floats(n) floats, 16-byte aligned) = upper triangle of Yt Yt^T, tile-packed.
EVOX_API int evox_hip_bf16_gram_packed(const void* Yt, int64_t ldy, int n, int mu, float* packed, float* ws, int64_t ws_floats,
                                       void* stream) {
  if ((uintptr_t)packed & 15) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  int splits = 0;
  int e = gram_slabs(Yt, ldy, n, mu, ws, ws_floats, &splits, st);
  if (e) return e;
  const int T64 = (n + 63) / 64;
  gram_pack_kernel<8><<<upper_tiles(T64), 256, 0, st>>>(ws, splits, packed, n, T64);
  return last_launch_status();
}

// out (n x n, both triangles) = alpha[0] C + c1 pc pc^T + cmu G from the tile-packed upper triangle of G; a positive
// *poison (optional) turns the result into NaNs.
EVOX_API int evox_hip_cov_from_packed(const float* packed, int n, const float* C, const float* pc, const float* alpha, float c1,
                                      float cmu, const float* poison, float* out, void* stream) {
  if (n <= 0 || (n & 3) || ((uintptr_t)packed & 15) || ((uintptr_t)C & 15) || ((uintptr_t)out & 15))
    return (int)hipErrorInvalidValue;
  const int T64 = (n + 63) / 64;
  cov_from_packed_kernel<<<upper_tiles(T64), 256, 0, (hipStream_t)stream>>>(packed, C, pc, alpha, c1, cmu, poison, out, n, T64);
  return last_launch_status();
}

// out (n x n fp32, both triangles) = alpha[0] * C + c1 * pc pc^T + cmu * Yt Yt^T, Yt bf16 [n][ldy] with K = mu.
// ws: evox_hip_bf16_cov_splits(n, mu) * n * n floats.
EVOX_API int evox_hip_bf16_cov(const void* Yt, int64_t ldy, int n, int mu, const float* C, const float* pc,
                               const float* alpha, float c1, float cmu, float* out, float* ws, int64_t ws_floats,
                               void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int splits = 0;
  int e = gram_slabs(Yt, ldy, n, mu, ws, ws_floats, &splits, st);
  if (e) return e;
  const int T64 = (n + 63) / 64;
  cov_reduce_kernel<8><<<upper_tiles(T64), 256, 0, st>>>(ws, splits, C, pc, alpha, c1, cmu, out, n, T64);
  return last_launch_status();
}
