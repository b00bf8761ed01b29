// fp32 GEMM family on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: fp32 in, fp32
// accumulate, bit-exact fmaf chain) with fused operand producers and epilogues:
//
//   * plain      C = alpha * op(A) op(B) + beta * C0
//   * cma_sample X = mean + sigma * (Z o d) B^T,  Z ~ N(0, I) generated from Philox inside the
//                operand producer (Z never exists in memory)
//   * cma_cov    C' = a C + c1 pc pc^T + cmu * Y^T diag(w) Y,  Y gathered from the population
//                through the fitness order, centred and scaled on the fly
//
// Structure: BM x BN output tile per 256-thread workgroup (4 wavefronts of 64 lanes as 2 x 2,
// each owning (BM/2) x (BN/2) as 32 x 32 MFMA tiles), BK = 16 deep K-steps, operands staged
// k-major in LDS ([k][m] / [k][n], so that the one-float-per-lane MFMA fragments are read with
// conflict-free ds_read_b32), register-staged double buffering with one barrier per K-step.
// blockIdx is remapped so that each XCD's private L2 sees a contiguous band of tiles.
#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 16;
constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------
// operand producers.  Each fills a (BK x R) k-major LDS tile for rows [r0, r0+R) and k in
// [k0, k0+BK).  fetch() issues the global loads / computes into registers, store() writes LDS.
// ---------------------------------------------------------------------------------------------

// X[r][k], k contiguous
template <int R>
struct RowMajorK {
  static constexpr int NV = R * BK / 4 / NT;  // float4 per thread
  const float* __restrict__ x;
  int64_t ld;
  int rows, kdim;
  bool vec;
  struct Regs { float4 v[NV]; };
  __device__ __forceinline__ void fetch(int r0, int k0, Regs& g) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int r = r0 + (f >> 2), k = k0 + 4 * (f & 3);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows) {
        const float* p = x + (int64_t)r * ld + k;
        if (vec && k + 3 < kdim) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (k < kdim) v.x = p[0];
          if (k + 1 < kdim) v.y = p[1];
          if (k + 2 < kdim) v.z = p[2];
          if (k + 3 < kdim) v.w = p[3];
        }
      }
      g.v[j] = v;
    }
  }
  template <int LD>
  __device__ __forceinline__ void store(const Regs& g, float* lds) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int r = f >> 2, k = 4 * (f & 3);
      lds[(k + 0) * LD + r] = g.v[j].x;
      lds[(k + 1) * LD + r] = g.v[j].y;
      lds[(k + 2) * LD + r] = g.v[j].z;
      lds[(k + 3) * LD + r] = g.v[j].w;
    }
  }
};

// X[k][r], r contiguous
template <int R>
struct KMajor {
  static constexpr int NV = R * BK / 4 / NT;
  const float* __restrict__ x;
  int64_t ld;
  int rows, kdim;
  bool vec;
  struct Regs { float4 v[NV]; };
  __device__ __forceinline__ void fetch(int r0, int k0, Regs& g) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int k = k0 + f / (R / 4), r = r0 + 4 * (f % (R / 4));
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kdim) {
        const float* p = x + (int64_t)k * ld + r;
        if (vec && r + 3 < rows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (r < rows) v.x = p[0];
          if (r + 1 < rows) v.y = p[1];
          if (r + 2 < rows) v.z = p[2];
          if (r + 3 < rows) v.w = p[3];
        }
      }
      g.v[j] = v;
    }
  }
  template <int LD>
  __device__ __forceinline__ void store(const Regs& g, float* lds) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int k = f / (R / 4), r = 4 * (f % (R / 4));
      *reinterpret_cast<float4*>(lds + k * LD + r) = g.v[j];
    }
  }
};

// (Z o d)[r][k] with Z[r][k] = normal(key, element (row_offset + r) * kdim + k)
template <int R>
struct PhiloxNormalTimesD {
  static constexpr int NV = R * BK / 4 / NT;
  uint32_t k0key, k1key;
  const float* __restrict__ d;
  int64_t row_offset;
  int rows, kdim;
  int pitch;   // columns of the logical noise tensor (kdim may be cut by split-K)
  struct Regs { float4 v[NV]; };
  __device__ __forceinline__ void fetch(int r0, int k0, Regs& g) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int r = r0 + (f >> 2), k = k0 + 4 * (f & 3);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && k < kdim) {
        const uint64_t e = (uint64_t)(row_offset + r) * (uint64_t)pitch + (uint64_t)k;
        const int sh = (int)(e & 3);
        const f32x4 a = normal4(philox_block(k0key, k1key, e >> 2, kStreamDraw));
        float z[4];
        if (sh == 0) {
          z[0] = a.x; z[1] = a.y; z[2] = a.z; z[3] = a.w;
        } else {
          const f32x4 b = normal4(philox_block(k0key, k1key, (e >> 2) + 1, kStreamDraw));
#pragma unroll
          for (int l = 0; l < 4; ++l) z[l] = (l + sh < 4) ? lane_of(a, l + sh) : lane_of(b, l + sh - 4);
        }
        v.x = z[0] * d[k];
        if (k + 1 < kdim) v.y = z[1] * d[k + 1];
        if (k + 2 < kdim) v.z = z[2] * d[k + 2];
        if (k + 3 < kdim) v.w = z[3] * d[k + 3];
      }
      g.v[j] = v;
    }
  }
  template <int LD>
  __device__ __forceinline__ void store(const Regs& g, float* lds) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int r = f >> 2, k = 4 * (f & 3);
      lds[(k + 0) * LD + r] = g.v[j].x;
      lds[(k + 1) * LD + r] = g.v[j].y;
      lds[(k + 2) * LD + r] = g.v[j].z;
      lds[(k + 3) * LD + r] = g.v[j].w;
    }
  }
};

// Y^T operand of the rank-mu update: element (k, r) = (pop[order[k]][r] - mean[r]) * s_k with
// s_k = sqrt(w_k) / sigma, so that sum_k s_k^2 y_k y_k^T = Y^T diag(w) Y / sigma^2.
template <int R>
struct GatherCentered {
  static constexpr int NV = R * BK / 4 / NT;
  const float* __restrict__ pop;
  const int64_t* __restrict__ order;
  const float* __restrict__ mean;
  const float* __restrict__ w;
  const float* __restrict__ sigma;
  int64_t ld;
  int rows, kdim;
  bool vec;
  struct Regs { float4 v[NV]; };
  __device__ __forceinline__ void fetch(int r0, int k0, Regs& g) const {
    const float inv_sigma = 1.0f / sigma[0];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int k = k0 + f / (R / 4), r = r0 + 4 * (f % (R / 4));
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kdim) {
        const float s = sqrtf(w[k]) * inv_sigma;
        const float* p = pop + order[k] * ld + r;
        if (vec && r + 3 < rows) {
          const float4 x = *reinterpret_cast<const float4*>(p);
          const float4 m = *reinterpret_cast<const float4*>(mean + r);
          v = make_float4((x.x - m.x) * s, (x.y - m.y) * s, (x.z - m.z) * s, (x.w - m.w) * s);
        } else {
          if (r < rows) v.x = (p[0] - mean[r]) * s;
          if (r + 1 < rows) v.y = (p[1] - mean[r + 1]) * s;
          if (r + 2 < rows) v.z = (p[2] - mean[r + 2]) * s;
          if (r + 3 < rows) v.w = (p[3] - mean[r + 3]) * s;
        }
      }
      g.v[j] = v;
    }
  }
  template <int LD>
  __device__ __forceinline__ void store(const Regs& g, float* lds) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = threadIdx.x + j * NT;
      const int k = f / (R / 4), r = 4 * (f % (R / 4));
      *reinterpret_cast<float4*>(lds + k * LD + r) = g.v[j];
    }
  }
};

// ---------------------------------------------------------------------------------------------
// epilogues: called once per accumulator element with its global (row, col)
// ---------------------------------------------------------------------------------------------
struct EpiAxpby {
  float* __restrict__ c;
  const float* __restrict__ c0;
  int64_t ldc, ldc0;
  float alpha, beta;
  __device__ __forceinline__ void operator()(int row, int col, float acc, int) const {
    float v = alpha * acc;
    if (c0 != nullptr) v = fmaf(beta, c0[(int64_t)row * ldc0 + col], v);
    c[(int64_t)row * ldc + col] = v;
  }
};

struct EpiSample {
  float* __restrict__ out;
  const float* __restrict__ mean;
  const float* __restrict__ sigma;
  int64_t ldo;
  __device__ __forceinline__ void operator()(int row, int col, float acc, int) const {
    out[(int64_t)row * ldo + col] = fmaf(sigma[0], acc, mean[col]);
  }
};

struct EpiCov {
  float* __restrict__ out;
  const float* __restrict__ c;
  const float* __restrict__ pc;
  const float* __restrict__ alpha;   // device scalar
  float c1, cmu;
  int64_t ld;
  __device__ __forceinline__ void operator()(int row, int col, float acc, int) const {
    const float v = fmaf(alpha[0], c[(int64_t)row * ld + col], fmaf(c1, pc[row] * pc[col], cmu * acc));
    out[(int64_t)row * ld + col] = v;
  }
};

// split-K partial result: slab `split` of a [splits][n][ld] workspace
struct EpiSlab {
  float* __restrict__ ws;
  int64_t ld, slab;
  __device__ __forceinline__ void operator()(int row, int col, float acc, int split) const {
    ws[split * slab + (int64_t)row * ld + col] = acc;
  }
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, class PA, class PB, class Epi, bool UPPER_ONLY>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(PA pa, PB pb, Epi epi, int M, int N, int K, int tiles_m,
                                                      int tiles_n) {
  constexpr int LDA = BM + 4, LDB = BN + 4;     // multiples of 4 keep float4 LDS stores aligned
  constexpr int WM = BM / 2, WN = BN / 2;       // wavefront tile
  constexpr int TM = WM / 32, TN = WN / 32;     // MFMA tiles per wavefront
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int nwg = tiles_m * tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  // column-major walk over tiles inside a band keeps the B panel of a band L2-resident
  const int tm = wg / tiles_n, tn = wg % tiles_n;
  if (UPPER_ONLY && tm > tn) return;
  const int m0 = tm * BM, n0 = tn * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typename PA::Regs ra;
  typename PB::Regs rb;
  // split-K: gridDim.y slices of the K range, each a multiple of BK long
  const int kchunk = ((K + (int)gridDim.y - 1) / (int)gridDim.y + BK - 1) / BK * BK;
  const int kbeg = blockIdx.y * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  pa.kdim = kend;
  pb.kdim = kend;
  pa.fetch(m0, kbeg, ra);
  pb.fetch(n0, kbeg, rb);
  pa.template store<LDA>(ra, As);
  pb.template store<LDB>(rb, Bs);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      pa.fetch(m0, kbeg + (kt + 1) * BK, ra);
      pb.fetch(n0, kbeg + (kt + 1) * BK, rb);
    }
    const float* a = As + cur * BK * LDA + wm + l31;
    const float* b = Bs + cur * BK * LDB + wn + l31;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a[(ks + lhi) * LDA + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b[(ks + lhi) * LDB + 32 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      pa.template store<LDA>(ra, As + (cur ^ 1) * BK * LDA);
      pb.template store<LDB>(rb, Bs + (cur ^ 1) * BK * LDB);
    }
    __syncthreads();
  }

  // accumulator element r of lane l: col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int col = n0 + wn + 32 * j + l31;
        if (row < M && col < N) {
          epi(row, col, acc[i][j][r], (int)blockIdx.y);
          if (UPPER_ONLY && tm != tn) epi(col, row, acc[i][j][r], (int)blockIdx.y);
        }
      }
}

template <int BM, int BN, class PA, class PB, class Epi, bool UPPER_ONLY = false>
int launch(PA pa, PB pb, Epi epi, int M, int N, int K, hipStream_t st, int splits = 1) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  gemm_f32_kernel<BM, BN, PA, PB, Epi, UPPER_ONLY><<<dim3(tiles_m * tiles_n, splits), NT, 0, st>>>(
      pa, pb, epi, M, N, K, tiles_m, tiles_n);
  return last_launch_status();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int BM, int BN>
int plain(const float* A, const float* B, float* C, const float* C0, int M, int N, int K, int64_t lda, int64_t ldb,
          int64_t ldc, int64_t ldc0, int a_kmajor, int b_kmajor, float alpha, float beta, hipStream_t st) {
  EpiAxpby epi{C, C0, ldc, ldc0, alpha, beta};
  const bool va = aligned16(A) && (lda % 4 == 0), vb = aligned16(B) && (ldb % 4 == 0);
  if (!a_kmajor && !b_kmajor)
    return launch<BM, BN>(RowMajorK<BM>{A, lda, M, K, va}, RowMajorK<BN>{B, ldb, N, K, vb}, epi, M, N, K, st);
  if (a_kmajor && !b_kmajor)
    return launch<BM, BN>(KMajor<BM>{A, lda, M, K, va}, RowMajorK<BN>{B, ldb, N, K, vb}, epi, M, N, K, st);
  if (!a_kmajor && b_kmajor)
    return launch<BM, BN>(RowMajorK<BM>{A, lda, M, K, va}, KMajor<BN>{B, ldb, N, K, vb}, epi, M, N, K, st);
  return launch<BM, BN>(KMajor<BM>{A, lda, M, K, va}, KMajor<BN>{B, ldb, N, K, vb}, epi, M, N, K, st);
}

}  // namespace

// C[M][N] = alpha * sum_k a(m,k) b(n,k) + beta * C0.
// a_kmajor = 0: A stored [M][K] (lda between rows); 1: A stored [K][M].  Same for B with N.
EVOX_API int evox_hip_gemm_f32(const float* A, const float* B, float* C, const float* C0, int M, int N, int K,
                               int64_t lda, int64_t ldb, int64_t ldc, int64_t ldc0, int a_kmajor, int b_kmajor,
                               float alpha, float beta, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  // small outputs use 64 x 64 tiles so that the launch still covers the 256 CUs
  const int64_t big_tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  if (big_tiles >= 192)
    return plain<128, 128>(A, B, C, C0, M, N, K, lda, ldb, ldc, ldc0, a_kmajor, b_kmajor, alpha, beta, st);
  return plain<64, 64>(A, B, C, C0, M, N, K, lda, ldb, ldc, ldc0, a_kmajor, b_kmajor, alpha, beta, st);
}

// The key is read on the device by the producer, so no host round trip is needed: the
// producer struct carries a pointer instead of the two words.
namespace {
template <int R>
struct PhiloxNormalTimesDPtr {
  static constexpr int NV = R * BK / 4 / NT;
  const int32_t* __restrict__ key;
  const float* __restrict__ d;
  int64_t row_offset;
  int rows, kdim, pitch;
  using Regs = typename PhiloxNormalTimesD<R>::Regs;
  __device__ __forceinline__ void fetch(int r0, int k0, Regs& g) const {
    PhiloxNormalTimesD<R> p{(uint32_t)key[0], (uint32_t)key[1], d, row_offset, rows, kdim, pitch};
    p.fetch(r0, k0, g);
  }
  template <int LD>
  __device__ __forceinline__ void store(const Regs& g, float* lds) const {
    PhiloxNormalTimesD<R> p{0u, 0u, d, row_offset, rows, kdim, pitch};
    p.template store<LD>(g, lds);
  }
};
}  // namespace

// out[lam][n] = mean + sigma * (Z o d) Bm^T ; Bm stored [n][n] row-major (eigenvectors in columns)
EVOX_API int evox_hip_cma_sample(const int32_t* key, const float* mean, const float* sigma, const float* Bm,
                                 const float* d, float* out, int lam, int n, int64_t row_offset, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool vb = aligned16(Bm) && (n % 4 == 0);
  EpiSample epi{out, mean, sigma, (int64_t)n};
  const int64_t big_tiles = (int64_t)((lam + 127) / 128) * ((n + 127) / 128);
  if (big_tiles >= 192)
    return launch<128, 128>(PhiloxNormalTimesDPtr<128>{key, d, row_offset, lam, n, n}, RowMajorK<128>{Bm, n, n, n, vb},
                            epi, lam, n, n, st);
  return launch<64, 64>(PhiloxNormalTimesDPtr<64>{key, d, row_offset, lam, n, n}, RowMajorK<64>{Bm, n, n, n, vb}, epi,
                        lam, n, n, st);
}

namespace {
// out = alpha C + c1 pc pc^T + cmu * sum_s slab_s  (the second half of the split-K rank-mu update)
__global__ __launch_bounds__(256) void cov_reduce_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ C,
                                                         const float* __restrict__ pc, const float* __restrict__ alpha,
                                                         float c1, float cmu, float* __restrict__ out, int n) {
  const int64_t total = (int64_t)n * n;
  const float a = alpha[0];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / n), col = (int)(e - (int64_t)row * n);
    float g = 0.f;
    for (int s = 0; s < splits; ++s) g += ws[(int64_t)s * total + e];
    out[e] = fmaf(a, C[e], fmaf(c1, pc[row] * pc[col], cmu * g));
  }
}
}  // namespace

// out = alpha * C + c1 * pc pc^T + cmu * sum_k w_k y_k y_k^T, y_k = (pop[order[k]] - mean) / sigma.
// ws: workspace of splits * n * n floats (splits > 1 spreads the K = mu reduction over more
// workgroups than the 64 x 64 upper-triangular tiles alone provide; the partial slabs are summed
// in a fixed order, so the result does not depend on scheduling).
EVOX_API int evox_hip_cma_cov(const float* C, const float* pop, const int64_t* order, const float* mean,
                              const float* sigma, const float* w, const float* pc, const float* alpha, float c1,
                              float cmu, float* out, int n, int mu, int64_t ldpop, float* ws, int splits,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const bool v = aligned16(pop) && aligned16(mean) && (ldpop % 4 == 0);
  GatherCentered<64> pa{pop, order, mean, w, sigma, ldpop, n, mu, v};
  if (splits <= 1 || ws == nullptr) {
    EpiCov epi{out, C, pc, alpha, c1, cmu, (int64_t)n};
    return launch<64, 64, GatherCentered<64>, GatherCentered<64>, EpiCov, true>(pa, pa, epi, n, n, mu, st);
  }
  EpiSlab epi{ws, (int64_t)n, (int64_t)n * n};
  int rc = launch<64, 64, GatherCentered<64>, GatherCentered<64>, EpiSlab, true>(pa, pa, epi, n, n, mu, st, splits);
  if (rc) return rc;
  cov_reduce_kernel<<<stream_grid((int64_t)n * n, 256), 256, 0, st>>>(ws, splits, C, pc, alpha, c1, cmu, out, n);
  return last_launch_status();
}
