// Direct hipBLASLt dispatch with cached, timed plans, and the native driver of the coupled
// Newton–Schulz square-root iteration that is built on it.
//
// Why not the tensor library's matmul: its host path costs 13–26 µs per product (descriptor
// construction, heuristic cache lookup, allocator traffic).  The CMA-ES generation at
// λ=10⁴, n=10³ issues ~30 GEMMs of 20–200 µs, so that dispatch cost — not the matrix cores —
// bounded the step.  Here a plan (descriptors + algorithm) is built once per shape, the
// candidate algorithms the heuristic returns are timed on the device once, and a product
// afterwards costs one hipblasLtMatmul call.
//
// The library is not linked: the process already holds one hipBLASLt (the tensor library's),
// and a second copy from another ROCm release must not be mixed in.  evox_hip_blas_init()
// receives the path of the copy to use and resolves the entry points with dlsym.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "bf16_gemm.h"
#include "hip_common.h"
#include "ns_scal.h"

using namespace evox::nsk;

namespace {

struct LtApi {
  void* lib = nullptr;
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
};

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  int candidates = 0;
  float best_us = -1.f;
};

struct Key {
  int64_t v[11];
  bool operator==(const Key& o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 1469598103934665603ull;
    for (int i = 0; i < 11; ++i) { h ^= (uint64_t)k.v[i]; h *= 1099511628211ull; }
    return (size_t)h;
  }
};

constexpr int kMaxDev = 16;
constexpr size_t kWorkspace = 64ull << 20;

LtApi g_lt;
std::mutex g_mu;
hipblasLtHandle_t g_handle[kMaxDev] = {};
void* g_ws[kMaxDev] = {};
std::unordered_map<Key, Plan, KeyHash> g_plans[kMaxDev];

template <class F>
bool resolve(F& slot, const char* name) {
  slot = (F)dlsym(g_lt.lib, name);
  return slot != nullptr;
}

int ensure_device(int dev) {
  if (dev < 0 || dev >= kMaxDev) return (int)hipErrorInvalidDevice;
  if (!g_handle[dev]) {
    if (g_lt.create(&g_handle[dev]) != HIPBLAS_STATUS_SUCCESS) return (int)hipErrorUnknown;
    hipError_t e = hipMalloc(&g_ws[dev], kWorkspace);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// Row-major D(M×N) = alpha·op(A)·op(B) + beta·C is issued as the column-major product
// Dᵀ = op(B)ᵀ·op(A)ᵀ: a row-major matrix IS its transpose in column-major storage, so the
// operands only swap places and keep their transposition flags.
int make_plan(int dev, const Key& key, int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
              int64_t lda, int64_t ldb, int64_t ldc, int64_t ldd, bool has_bias, bool dev_scalars, Plan& p) {
  hipDataType in_t = in_dtype == 1 ? HIP_R_16BF : HIP_R_32F;
  if (g_lt.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return (int)hipErrorUnknown;
  int32_t opA = transB ? HIPBLAS_OP_T : HIPBLAS_OP_N;   // library's A is our B
  int32_t opB = transA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(opA));
  g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(opB));
  // stored shapes (column-major): our B row-major (K×N | N×K if transB) is (N×K | K×N)
  bool ok = true;
  ok &= g_lt.layout_create(&p.a, in_t, transB ? K : N, transB ? N : K, ldb) == HIPBLAS_STATUS_SUCCESS;
  ok &= g_lt.layout_create(&p.b, in_t, transA ? M : K, transA ? K : M, lda) == HIPBLAS_STATUS_SUCCESS;
  ok &= g_lt.layout_create(&p.c, HIP_R_32F, N, M, ldc) == HIPBLAS_STATUS_SUCCESS;
  ok &= g_lt.layout_create(&p.d, HIP_R_32F, N, M, ldd) == HIPBLAS_STATUS_SUCCESS;
  if (!ok) return (int)hipErrorUnknown;
  if (has_bias) {
    // bias of length N (our columns = the library's rows of D), added to every row of our D
    uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;
    int32_t bias_t = HIP_R_32F;
    ok &= g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS;
    ok &= g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_t, sizeof(bias_t)) == HIPBLAS_STATUS_SUCCESS;
  }
  if (dev_scalars) {
    // alpha: device vector with one scale per row of the library's D (= per column of ours)
    int32_t mode = HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST;
    ok &= g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &mode, sizeof(mode)) == HIPBLAS_STATUS_SUCCESS;
  }
  if (!ok) return (int)hipErrorNotSupported;
  (void)key;
  (void)dev;
  return 0;
}

int pick_algorithm(int dev, Plan& p, int tune, int64_t M, int64_t N, const void* alpha, const void* A, const void* B,
                   const void* beta, const void* C, hipStream_t stream) {
  hipblasLtMatmulPreference_t pref;
  if (g_lt.pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) return (int)hipErrorUnknown;
  uint64_t ws = kWorkspace;
  g_lt.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
  constexpr int kReq = 32;
  hipblasLtMatmulHeuristicResult_t res[kReq];
  int got = 0;
  hipblasStatus_t st = g_lt.heuristic(g_handle[dev], p.desc, p.a, p.b, p.c, p.d, pref, kReq, res, &got);
  g_lt.pref_destroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || got <= 0) return (int)hipErrorNotSupported;
  p.candidates = got;
  p.algo = res[0].algo;
  p.ws = res[0].workspaceSize;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipStreamIsCapturing(stream, &cap);
  if (!tune || got == 1 || cap != hipStreamCaptureStatusNone) return 0;
  // time every candidate into a scratch output (the caller's D may alias C)
  float* scratch = nullptr;
  if (hipMalloc((void**)&scratch, sizeof(float) * (size_t)M * (size_t)N) != hipSuccess) return 0;
  hipblasLtMatrixLayout_t dl = nullptr;
  if (g_lt.layout_create(&dl, HIP_R_32F, N, M, N) != HIPBLAS_STATUS_SUCCESS) { hipFree(scratch); return 0; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  const bool verbose = getenv("EVOX_AMD_BLAS_VERBOSE") != nullptr;
  for (int i = 0; i < got; ++i) {
    if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > kWorkspace) continue;
    bool fine = true;
    for (int w = 0; w < 2 && fine; ++w)
      fine = g_lt.matmul(g_handle[dev], p.desc, alpha, B, p.a, A, p.b, beta, C, p.c, scratch, dl, &res[i].algo,
                         g_ws[dev], kWorkspace, stream) == HIPBLAS_STATUS_SUCCESS;
    if (!fine) continue;
    constexpr int kRuns = 10;
    hipEventRecord(e0, stream);
    for (int r = 0; r < kRuns; ++r)
      g_lt.matmul(g_handle[dev], p.desc, alpha, B, p.a, A, p.b, beta, C, p.c, scratch, dl, &res[i].algo, g_ws[dev],
                  kWorkspace, stream);
    hipEventRecord(e1, stream);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (verbose) fprintf(stderr, "[evox blas] M=%ld N=%ld candidate %d: %.1f us (workspace %zu)\n", (long)M, (long)N, i, ms * 1000.f / kRuns, res[i].workspaceSize);
    if (ms < best) { best = ms; p.algo = res[i].algo; p.ws = res[i].workspaceSize; }
  }
  p.best_us = best < 1e29f ? best * 1000.f / 10.f : -1.f;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipStreamSynchronize(stream);
  hipFree(scratch);
  return 0;
}

// bias (optional): fp32 vector of length N added to every row of D.  alpha_dev (optional): a
// device vector of N per-column scales that replaces `alpha` (read when the kernel runs).
int gemm(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void* A,
         int64_t lda, const void* B, int64_t ldb, float beta, const float* C, int64_t ldc, float* D, int64_t ldd,
         int tune, hipStream_t stream, const float* bias = nullptr, const float* alpha_dev = nullptr) {
  if (!g_lt.matmul) return (int)hipErrorNotInitialized;
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  int e = ensure_device(dev);
  if (e) return e;
  const void* pa = alpha_dev ? (const void*)alpha_dev : (const void*)&alpha;
  const void* pb = (const void*)&beta;
  const int64_t flags = (int64_t)(beta != 0.f) | ((int64_t)(C != D) << 1) | ((int64_t)(bias != nullptr) << 2) |
                        ((int64_t)(alpha_dev != nullptr) << 3);
  Key key{{in_dtype, transA, transB, M, N, K, lda, ldb, ldc, ldd, flags}};
  auto it = g_plans[dev].find(key);
  if (it == g_plans[dev].end()) {
    Plan p;
    e = make_plan(dev, key, in_dtype, transA, transB, M, N, K, lda, ldb, ldc, ldd, bias != nullptr,
                  alpha_dev != nullptr, p);
    if (e) return e;
    if (bias) g_lt.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    e = pick_algorithm(dev, p, tune, M, N, pa, A, B, pb, C, stream);
    if (e) return e;
    it = g_plans[dev].emplace(key, p).first;
  }
  Plan* plan = &it->second;
  if (bias) g_lt.desc_set(plan->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  hipblasStatus_t st = g_lt.matmul(g_handle[dev], plan->desc, pa, B, plan->a, A, plan->b, pb, C, plan->c, D, plan->d,
                                   &plan->algo, g_ws[dev], kWorkspace, stream);
  return st == HIPBLAS_STATUS_SUCCESS ? 0 : 1000 + (int)st;
}

// ---------------------------------------------------------------------------------------
// Newton–Schulz helper kernels
// ---------------------------------------------------------------------------------------
// one wave per row: w = C·(v/‖v‖); the first step also collects ‖C‖∞ and the trace
__global__ __launch_bounds__(256) void ns_matvec_kernel(const float* __restrict__ C, int n,
                                                        const float* __restrict__ vin, float* __restrict__ vout,
                                                        float* scal, int step) {
  __shared__ float red[4];
  __shared__ float wsq[4], wmx[4], wdg[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float inv = 1.f;
  if (step > 0) inv = rsqrtf(fmaxf(block_reduce(part_row(scal, step - 1), gridDim.x, false, red), 1e-38f));
  float sq = 0.f, mx = 0.f, dg = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    const float* r = C + (size_t)row * n;
    float acc = 0.f, asum = 0.f;
    if ((n & 3) == 0) {                              // rows are 16-byte aligned: four columns per load
      const float4* r4 = reinterpret_cast<const float4*>(r);
      const float4* v4 = reinterpret_cast<const float4*>(vin);
      for (int j = lane; j < (n >> 2); j += 64) {
        const float4 c = r4[j], v = v4[j];
        acc += (c.x * v.x + c.y * v.y) + (c.z * v.z + c.w * v.w);
        asum += (fabsf(c.x) + fabsf(c.y)) + (fabsf(c.z) + fabsf(c.w));
      }
    } else {
      for (int j = lane; j < n; j += 64) {
        const float c = r[j];
        acc += c * vin[j];
        asum += fabsf(c);
      }
    }
    acc = evox::wave_sum(acc) * inv;
    if (lane == 0) vout[row] = acc;
    sq += acc * acc;
    if (step == 0) {
      mx = fmaxf(mx, evox::wave_sum(asum));
      dg += r[row];
    }
  }
  if (lane == 0) { wsq[wave] = sq; wmx[wave] = mx; wdg[wave] = dg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part_row(scal, step)[blockIdx.x] = (wsq[0] + wsq[1]) + (wsq[2] + wsq[3]);
    if (step == 0) {
      part_row(scal, kRowInf)[blockIdx.x] = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
      part_row(scal, kRowDiag)[blockIdx.x] = (wdg[0] + wdg[1]) + (wdg[2] + wdg[3]);
    }
  }
}


// Y0 = C/s, T = 1.5 I − 0.5 Y0; block 0 also stores s and the normalised power vector.
// With Q != nullptr (degree-4 first step) T = c0 I + c1 Y0 and Q = c2 I + c3 Y0 instead: the two linear parts
// of  p(M) = (c0 I + c1 M) + M² ((c2 I + c3 M) + c4 M²),  the degree-4 Taylor polynomial of M^{-1/2} about I.
// safe: s = max(tr C/n, ‖C‖∞/1.9) (spectrum of Y0 provably inside (0, 1.9]); otherwise s is the
// power-iteration estimate ‖C v‖ (‖v‖ = 1), which never exceeds λmax — iterates with eigenvalues
// up to 3 still converge, so an under-estimate by less than that factor is harmless.
__global__ __launch_bounds__(256) void ns_scale_kernel(const float* __restrict__ C, int n, float* __restrict__ Y,
                                                       float* __restrict__ T, float* scal, float* vec, int steps,
                                                       int safe, int mv_blocks, float* __restrict__ Q) {
  __shared__ float red[4];
  const float inf_norm = block_reduce(part_row(scal, kRowInf), mv_blocks, true, red);
  const float trace = block_reduce(part_row(scal, kRowDiag), mv_blocks, false, red);
  const float last = block_reduce(part_row(scal, steps - 1), mv_blocks, false, red);
  const float safe_s = fmaxf(trace / (float)n, inf_norm / 1.9f);
  const float lam = sqrtf(last) * 1.02f;
  const float s = (!safe && lam > 0.f && lam < 3.0e38f) ? fminf(lam, safe_s * 1.9f) : safe_s;
  const float is = 1.f / s;
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float y = C[i] * is;
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    Y[i] = y;
    if (Q != nullptr) {
      T[i] = (r == c ? kP4c0 : 0.f) + kP4c1 * y;
      Q[i] = (r == c ? kP4c2 : 0.f) + kP4c3 * y;
    } else {
      T[i] = (r == c ? 1.5f : 0.f) - 0.5f * y;
    }
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) scal[13] = s;
    const float* w = vec + (steps & 1) * n;        // step j writes buffer (j+1)&1
    const float inv = rsqrtf(fmaxf(last, 1e-38f));
    for (int j = threadIdx.x; j < n; j += blockDim.x) vec[j] = w[j] * inv;
  }
}

// q += c · a  (n² elements)
__global__ __launch_bounds__(256) void ns_axpy_kernel(const float* __restrict__ a, float c, float* __restrict__ q, int64_t total) {
  const int64_t n4 = total >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  float4* q4 = reinterpret_cast<float4*>(q);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 x = a4[i];
    float4 y = q4[i];
    y.x += c * x.x; y.y += c * x.y; y.z += c * x.z; y.w += c * x.w;
    q4[i] = y;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    q[i] += c * a[i];
}

// Mean-square deviation of T from the identity, estimated from every kResStride-th row (the deviation is
// spread over the whole matrix — it is a polynomial in the scaled covariance — and a non-finite entry reaches
// every row within one product, so a row sample serves both the iteration control and the output guard).
constexpr int kResStride = 8;

__global__ __launch_bounds__(256) void ns_residual_kernel(const float* __restrict__ T, int n, float* scal) {
  const int rows = (n + kResStride - 1) / kResStride;
  const int64_t total = (int64_t)rows * n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / n) * kResStride, c = (int)(i % n);
    const float d = T[(int64_t)r * n + c] - (r == c ? 1.f : 0.f);
    acc += d * d;
  }
  acc = evox::wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    part_row(scal, kRowRes)[blockIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) * ((float)n / (float)rows);
}

// sqrtC = ½(Y+Yᵀ)√s, invsqrtC = ½(Z+Zᵀ)/√s through 32×32 LDS tiles; when the residual shows the
// iteration has not contracted (or is not finite) the previous factors are passed on instead
__global__ __launch_bounds__(256) void ns_finish_kernel(const float* __restrict__ Y, const float* __restrict__ Z, int n,
                                                        float* scal, const float* __restrict__ prev_sqrt,
                                                        const float* __restrict__ prev_inv, float* __restrict__ out_sqrt,
                                                        float* __restrict__ out_inv, int res_blocks,
                                                        uint16_t* __restrict__ out_sqrt_bf16) {
  __shared__ float ty[32][33], tz[32][33];
  __shared__ float red[4];
  const float res = 4.f * block_reduce(part_row(scal, kRowRes), res_blocks, false, red) / (float)n;  // (2‖T−I‖_F/√n)²
  const bool good = res < 0.25f;                       // also false for NaN
  const bool keep = !good && prev_sqrt != nullptr;
  const float s = scal[13];
  const float rs = sqrtf(s), irs = 1.f / rs;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, tyi = threadIdx.x >> 5;   // 32 × 8
  for (int k = tyi; k < 32; k += 8) {
    const int r = bx + k, c = by + tx;                 // the mirrored tile
    if (r < n && c < n) {
      ty[k][tx] = Y[(size_t)r * n + c];
      tz[k][tx] = Z[(size_t)r * n + c];
    }
  }
  __syncthreads();
  for (int k = tyi; k < 32; k += 8) {
    const int r = by + k, c = bx + tx;
    if (r < n && c < n) {
      const size_t i = (size_t)r * n + c;
      const float vs = keep ? prev_sqrt[i] : 0.5f * (Y[i] + ty[tx][k]) * rs;
      out_sqrt[i] = vs;
      out_inv[i] = keep ? prev_inv[i] : 0.5f * (Z[i] + tz[tx][k]) * irs;
      if (out_sqrt_bf16 != nullptr) {                   // rounded copy for bf16-operand consumers (to nearest even)
        const uint32_t u = __float_as_uint(vs);
        out_sqrt_bf16[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      }
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    scal[14] = res;
    scal[15] = keep ? 1.f : 0.f;
  }
}

}  // namespace

EVOX_API int evox_hip_blas_init(const char* path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_lt.matmul) return 0;
  g_lt.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!g_lt.lib) return 1;
  bool ok = true;
  ok &= resolve(g_lt.create, "hipblasLtCreate");
  ok &= resolve(g_lt.desc_create, "hipblasLtMatmulDescCreate");
  ok &= resolve(g_lt.desc_set, "hipblasLtMatmulDescSetAttribute");
  ok &= resolve(g_lt.layout_create, "hipblasLtMatrixLayoutCreate");
  ok &= resolve(g_lt.pref_create, "hipblasLtMatmulPreferenceCreate");
  ok &= resolve(g_lt.pref_set, "hipblasLtMatmulPreferenceSetAttribute");
  ok &= resolve(g_lt.pref_destroy, "hipblasLtMatmulPreferenceDestroy");
  ok &= resolve(g_lt.heuristic, "hipblasLtMatmulAlgoGetHeuristic");
  decltype(&hipblasLtMatmul) mm = nullptr;
  ok &= resolve(mm, "hipblasLtMatmul");
  if (!ok) return 2;
  g_lt.matmul = mm;
  return 0;
}

// in_dtype: 0 = fp32 operands, 1 = bf16 operands (fp32 accumulate and output either way)
EVOX_API int evox_hip_gemm(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                           const void* A, int64_t lda, const void* B, int64_t ldb, float beta, const float* C,
                           int64_t ldc, float* D, int64_t ldd, const float* bias, const float* alpha_dev, int tune,
                           void* stream) {
  return gemm(in_dtype, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C ? C : D, C ? ldc : ldd, D, ldd, tune,
              (hipStream_t)stream, bias, alpha_dev);
}

// info[0] = number of candidate algorithms, info[1] = best timed µs (−1 when not timed)
EVOX_API int evox_hip_gemm_plan_info(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                                     int64_t lda, int64_t ldb, int64_t ldc, int64_t ldd, int flags, float* info) {
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_mu);
  Key key{{in_dtype, transA, transB, M, N, K, lda, ldb, ldc, ldd, (int64_t)flags}};
  auto it = g_plans[dev].find(key);
  if (it == g_plans[dev].end()) return 1;
  info[0] = (float)it->second.candidates;
  info[1] = it->second.best_us;
  return 0;
}

// Coupled Newton–Schulz iteration for C^{±1/2} (C symmetric positive definite, n×n, dense):
//   Y₀ = C/s, Z₀ = I;  T = (3I − Z Y)/2;  Y ← Y T;  Z ← T Z
// work: 6·n² floats (Y, Z, T ping-pong); eye: n×n identity; vec: 2n floats, vec[0:n] holds the
// unit power-iteration vector carried between calls; scal: evox_hip_ns_scal_floats() floats.
// power_steps matvecs refine the λmax estimate that sets s; safe != 0 uses the norm bound
// max(tr C/n, ‖C‖∞/1.9) instead (always convergent, more iterations).
EVOX_API int evox_hip_ns_scal_floats() { return kScalFloats; }

//
// first_step = 1 replaces iteration 1 (Z₀ = I, so everything is a polynomial in Y₀) by the fifth-order step
// Z₁ = p(Y₀), Y₁ = Y₀ p(Y₀) with the degree-4 polynomial above: 3 GEMMs and one elementwise pass instead of 1
// GEMM, after which one standard iteration reaches what took two (error e → ≈ e⁵/2 instead of ¾e²).
EVOX_API int evox_hip_ns_sqrt(const float* C, int n, int iters, int power_steps, int safe, float* work,
                              const float* eye, float* vec, float* scal, const float* prev_sqrt,
                              const float* prev_inv, float* out_sqrt, float* out_inv, int tune, int first_step,
                              void* out_sqrt_bf16, void* stream_) {
  if (n <= 0 || iters < 1 || power_steps < 1 || power_steps > kMaxPow) return (int)hipErrorInvalidValue;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nn = (size_t)n * n;
  float* Yb[2] = {work, work + nn};
  float* Zb[2] = {work + 2 * nn, work + 3 * nn};
  float* Tb[2] = {work + 4 * nn, work + 5 * nn};
  int mv_blocks = (n + 3) / 4;
  if (mv_blocks > kPartRow) mv_blocks = kPartRow;
  for (int j = 0; j < power_steps; ++j)
    ns_matvec_kernel<<<mv_blocks, 256, 0, stream>>>(C, n, vec + (j & 1) * n, vec + ((j + 1) & 1) * n, scal, j);
  const bool poly4 = first_step == 1 && (nn & 3) == 0;
  ns_scale_kernel<<<evox::stream_grid((int64_t)nn, 256), 256, 0, stream>>>(C, n, Yb[0], Tb[0], scal, vec, power_steps,
                                                                            safe, mv_blocks, poly4 ? Zb[0] : nullptr);
  int e = evox::last_launch_status();
  if (e) return e;
  const float* Y;
  const float* Z = Tb[0];
  const float* Tlast = Tb[0];
  int yi = 0, zi = 0;
  if (poly4) {
    // A2 = Y₀², q = (c2 I + c3 Y₀) + c4 A2, p = (c0 I + c1 Y₀) + A2 q (in place over its linear part), Y₁ = Y₀ p
    e = gemm(0, 0, 0, n, n, n, 1.f, Yb[0], n, Yb[0], n, 0.f, Yb[1], n, Yb[1], n, tune, stream);
    if (e) return e;
    ns_axpy_kernel<<<evox::stream_grid((int64_t)(nn >> 2), 256), 256, 0, stream>>>(Yb[1], kP4c4, Zb[0], (int64_t)nn);
    e = evox::last_launch_status();
    if (e) return e;
    e = gemm(0, 0, 0, n, n, n, 1.f, Yb[1], n, Zb[0], n, 1.f, Tb[0], n, Tb[0], n, tune, stream);
    if (e) return e;
    e = gemm(0, 0, 0, n, n, n, 1.f, Yb[0], n, Tb[0], n, 0.f, Zb[1], n, Zb[1], n, tune, stream);
    if (e) return e;
    Y = Zb[1];
  } else {
    // iteration 1 with Z₀ = I: Y₁ = Y₀ T, Z₁ = T
    e = gemm(0, 0, 0, n, n, n, 1.f, Yb[0], n, Tb[0], n, 0.f, Yb[1], n, Yb[1], n, tune, stream);
    if (e) return e;
    Y = Yb[1];
  }
  for (int it = 1; it < iters; ++it) {
    float* Tn = Tb[it & 1];
    e = gemm(0, 0, 0, n, n, n, -0.5f, Z, n, Y, n, 1.5f, eye, n, Tn, n, tune, stream);
    if (e) return e;
    float* Yn = Yb[yi];
    float* Zn = Zb[zi];
    e = gemm(0, 0, 0, n, n, n, 1.f, Y, n, Tn, n, 0.f, Yn, n, Yn, n, tune, stream);
    if (e) return e;
    e = gemm(0, 0, 0, n, n, n, 1.f, Tn, n, Z, n, 0.f, Zn, n, Zn, n, tune, stream);
    if (e) return e;
    Y = Yn; Z = Zn; Tlast = Tn;
    yi ^= 1; zi ^= 1;
  }
  int res_blocks = evox::stream_grid((int64_t)((n + kResStride - 1) / kResStride) * n, 256);
  if (res_blocks > kPartRow) res_blocks = kPartRow;
  ns_residual_kernel<<<res_blocks, 256, 0, stream>>>(Tlast, n, scal);
  dim3 grid((n + 31) / 32, (n + 31) / 32);
  ns_finish_kernel<<<grid, 256, 0, stream>>>(Y, Z, n, scal, prev_sqrt, prev_inv, out_sqrt, out_inv, res_blocks,
                                             (uint16_t*)out_sqrt_bf16);
  return evox::last_launch_status();
}

// ---------------------------------------------------------------------------------------
// The same iteration on the bf16 matrix cores (hand-written products, bf16_gemm.h).
//
// Every iterate is a symmetric polynomial in C, so a product of two of them is symmetric: only the tiles on and
// above the diagonal are multiplied and each result is mirrored on write.  The ORDER of the factors still matters:
// Y <- Y T and Z <- T Z is the coupled form whose first-order error propagation is idempotent and maps symmetric
// errors to symmetric errors (so mirroring the upper triangle does not disturb it); Z <- Z T amplifies the error
// component between eigenvalues l_i << l_j by (sqrt(l_j / l_i) - 1) / 2 per step and diverges at cond(C) ~ 10^3.  An fp32 matrix is carried as NSP bf16
// planes (x = hi + lo [+ lo2], each plane the rounding of what the previous ones left), and a product is the sum
// of the plane products whose orders add up to less than NSP: 16 (NSP = 2) or 24 (NSP = 3) significant bits per
// operand at 3 or 6 matrix instructions per tile step, against 16 times fewer flops per clock for fp32 inputs.
// ---------------------------------------------------------------------------------------
namespace {

using namespace evox::bg;

// Y0 = C / s as planes; T0 = c0 I + c1 Y0 and Q0 = c2 I + c3 Y0 in fp32 (addends of the first products)
template <int NSP>
__global__ __launch_bounds__(256) void ns_scale_planes_kernel(const float* __restrict__ C, int n, int64_t ldp,
                                                              uint16_t* __restrict__ y0, int64_t plane, float* __restrict__ T0,
                                                              float* __restrict__ Q0, float* scal, float* vec, int steps,
                                                              int safe, int mv_blocks) {
  __shared__ float red[4];
  const float inf_norm = block_reduce(part_row(scal, kRowInf), mv_blocks, true, red);
  const float trace = block_reduce(part_row(scal, kRowDiag), mv_blocks, false, red);
  const float last = block_reduce(part_row(scal, steps - 1), mv_blocks, false, red);
  const float safe_s = fmaxf(trace / (float)n, inf_norm / 1.9f);
  const float lam = sqrtf(last) * 1.02f;
  const float s = (!safe && lam > 0.f && lam < 3.0e38f) ? fminf(lam, safe_s * 1.9f) : safe_s;
  const float is = 1.f / s;
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float y = C[i] * is;
    const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
    float rem = y;
#pragma unroll
    for (int p = 0; p < NSP; ++p) {
      const uint32_t b = f32_to_bf16_bits(rem);
      y0[p * plane + (int64_t)r * ldp + c] = (uint16_t)b;
      rem -= bf16_bits_to_f32(b);
    }
    T0[i] = (r == c ? kP4c0 : 0.f) + kP4c1 * y;
    Q0[i] = (r == c ? kP4c2 : 0.f) + kP4c3 * y;
  }
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      scal[13] = s;
      scal[12] = sqrtf(s);
      scal[11] = 1.f / sqrtf(s);
    }
    const float* w = vec + (steps & 1) * n;
    const float inv = rsqrtf(fmaxf(last, 1e-38f));
    for (int j = threadIdx.x; j < n; j += blockDim.x) vec[j] = w[j] * inv;
  }
}

// Residual of a finished factorisation without forming Z Y: |(Z (S v) - v)| / |v| for a fixed pseudo-random probe v.
// For E = Z S - I and a random direction, E|E v|^2 = |E|_F^2 / n, the quantity the T-based bound measures.

// one wave per row: y = A x, or with x == nullptr y = A probe
__global__ __launch_bounds__(256) void ns_probe_mv_kernel(const float* __restrict__ A, int n, const float* __restrict__ x,
                                                          float* __restrict__ y, float* scal, int residual) {
  __shared__ float sd[4], sv[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float d2 = 0.f, v2 = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
    const float* r = A + (size_t)row * n;
    float acc = 0.f;
    if ((n & 3) == 0) {
      const float4* r4 = reinterpret_cast<const float4*>(r);
      for (int j = lane; j < (n >> 2); j += 64) {
        const float4 a = r4[j];
        float4 v;
        if (x != nullptr) v = reinterpret_cast<const float4*>(x)[j];
        else v = make_float4(probe_v(4 * j), probe_v(4 * j + 1), probe_v(4 * j + 2), probe_v(4 * j + 3));
        acc += (a.x * v.x + a.y * v.y) + (a.z * v.z + a.w * v.w);
      }
    } else {
      for (int j = lane; j < n; j += 64) acc = fmaf(r[j], x != nullptr ? x[j] : probe_v(j), acc);
    }
    acc = evox::wave_sum(acc);
    if (residual) {
      const float v = probe_v(row), d = acc - v;
      d2 = fmaf(d, d, d2);
      v2 = fmaf(v, v, v2);
    } else if (lane == 0) {
      y[row] = acc;
    }
  }
  if (residual) {
    if (lane == 0) { sd[wave] = d2; sv[wave] = v2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      part_row(scal, kRowRes)[blockIdx.x] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
      part_row(scal, kMaxPow - 1)[blockIdx.x] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
    }
  }
}

// The probe residual from its partials and the guard in one launch: every block derives the verdict itself (250
// partials), block 0 records it, and only a failed factorisation copies the previous factors over the outputs.
__global__ __launch_bounds__(256) void ns_probe_guard_kernel(float* scal, int blocks, const float* __restrict__ prev_sqrt,
                                                             const float* __restrict__ prev_inv, float* __restrict__ out_sqrt,
                                                             float* __restrict__ out_inv, uint16_t* __restrict__ out_b16, int n,
                                                             int64_t ldp) {
  __shared__ float red[4];
  const float d2 = block_reduce(part_row(scal, kRowRes), blocks, false, red);
  const float v2 = block_reduce(part_row(scal, kMaxPow - 1), blocks, false, red);
  const float res = sqrtf(d2 / fmaxf(v2, 1e-38f));
  const bool failed = !(res < 0.25f) && prev_sqrt != nullptr && prev_inv != nullptr;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[14] = res;
    scal[15] = failed ? 1.f : 0.f;
  }
  if (!failed) return;
  const int64_t total = (int64_t)n * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float sv = prev_sqrt[i];
    out_sqrt[i] = sv;
    out_inv[i] = prev_inv[i];
    if (out_b16 != nullptr) out_b16[(i / n) * ldp + (i % n)] = (uint16_t)f32_to_bf16_bits(sv);
  }
}

// residual bound from the partials of ns_residual_kernel, the keep flag and the output scales
__global__ __launch_bounds__(256) void ns_flags_kernel(float* scal, int n, int res_blocks, int have_prev) {
  __shared__ float red[4];
  const float res = 4.f * block_reduce(part_row(scal, kRowRes), res_blocks, false, red) / (float)n;
  if (threadIdx.x == 0) {
    const bool good = res < 0.25f;                        // false for NaN
    const float rs = sqrtf(scal[13]);
    scal[14] = res;
    scal[15] = (!good && have_prev) ? 1.f : 0.f;
    scal[12] = rs;
    scal[11] = 1.f / rs;
  }
}

template <int NSP, int NPO>
int ns_product(const uint16_t* a0, const uint16_t* a1, const uint16_t* b0, const uint16_t* b1, int batch, int64_t plane,
               int64_t ldp, int n, const EpiAffine<NPO>& epi, hipStream_t st) {
  Args<NSP> g{};
  for (int p = 0; p < NSP; ++p) {
    g.a[0][p] = a0 + p * plane;
    g.b[0][p] = b0 + p * plane;
    g.a[1][p] = (batch > 1 ? a1 : a0) + p * plane;
    g.b[1][p] = (batch > 1 ? b1 : b0) + p * plane;
  }
  g.lda = g.ldb = ldp;
  g.M = g.N = n;
  g.ksteps = (n + 63) / 64;
  g.ksteps_per_split = g.ksteps;
  const int T = (n + 63) / 64;
  g.tiles_m = g.tiles_n = T;
  // two products in one launch are more workgroups than CUs: two stages (64 KiB at two planes) let two of them
  // share a CU; a single product runs one workgroup per CU and keeps a third stage in flight
  if (batch > 1) return launch<64, 64, 2, 4, NSP, true, 2, EpiAffine<NPO>>(g, epi, upper_tiles(T), 1, batch, st);
  return launch<64, 64, 2, 4, NSP, true, 3, EpiAffine<NPO>>(g, epi, upper_tiles(T), 1, batch, st);
}

// the same without symmetry: every tile of the T x T grid, nothing mirrored
template <int NSP, int NPO>
int ns_product_full(const uint16_t* a0, const uint16_t* a1, const uint16_t* b0, const uint16_t* b1, int batch, int64_t plane,
                    int64_t ldp, int n, const EpiAffine<NPO>& epi, hipStream_t st) {
  Args<NSP> g{};
  for (int p = 0; p < NSP; ++p) {
    g.a[0][p] = a0 + p * plane;
    g.b[0][p] = b0 + p * plane;
    g.a[1][p] = (batch > 1 ? a1 : a0) + p * plane;
    g.b[1][p] = (batch > 1 ? b1 : b0) + p * plane;
  }
  g.lda = g.ldb = ldp;
  g.M = g.N = n;
  g.ksteps = (n + 63) / 64;
  g.ksteps_per_split = g.ksteps;
  const int T = (n + 63) / 64;
  g.tiles_m = g.tiles_n = T;
  if (batch > 1) return launch<64, 64, 2, 4, NSP, false, 2, EpiAffine<NPO>>(g, epi, T * T, 1, batch, st);
  return launch<64, 64, 2, 4, NSP, false, 3, EpiAffine<NPO>>(g, epi, T * T, 1, batch, st);
}

template <int NPO>
EpiAffine<NPO> affine(int n, int64_t ldp) {
  EpiAffine<NPO> e{};
  e.nout = 1;
  e.ldf = n;
  e.ldp = ldp;
  e.M = e.N = n;
  e.keep = nullptr;
  return e;
}

template <int NPO>
void set_planes(typename EpiAffine<NPO>::Out& o, int z, uint16_t* base, int64_t plane) {
  for (int p = 0; p < NPO; ++p) o.pl[z][p] = base + p * plane;
}

template <int NPO>
void set_planes_t(typename EpiAffine<NPO>::Out& o, int z, uint16_t* base, int64_t plane) {
  for (int p = 0; p < NPO; ++p) o.plt[z][p] = base + p * plane;
}

// full != 0: the iterations after the fifth-order step multiply every tile and mirror nothing.  Mirroring replaces a
// product Y T by the symmetric matrix made of its upper triangle; Y and T commute only up to rounding, the difference
// feeds back into the iteration and is amplified by about (sqrt(cond) - 1) / 2 per step, which limits the mirrored form
// to well-conditioned matrices.  The plain form keeps every iterate in both orientations (the NT product takes its
// right factor row-major) at twice the tiles per product; only the final products, whose results leave the iteration,
// are mirrored again so that the factors come out symmetric to the last bit.
template <int NSP>
int ns_sqrt_planes(const float* C, int n, int64_t ldp, int iters, int power_steps, int safe, int full, uint16_t* planes,
                   float* f32ws, float* vec, float* scal, const float* prev_sqrt, const float* prev_inv, float* out_sqrt,
                   float* out_inv, uint16_t* out_sqrt_bf16, hipStream_t stream) {
  const int64_t nn = (int64_t)n * n;
  const int64_t plane = (int64_t)n * ldp;               // elements of one bf16 plane
  const int64_t sm = plane * NSP;                       // one split matrix
  uint16_t* Y0 = planes;
  uint16_t* A2 = planes + sm;
  uint16_t* Qp = planes + 2 * sm;
  uint16_t* Pp = planes + 3 * sm;
  uint16_t* Y1 = planes + 4 * sm;
  uint16_t* Tp = planes + 5 * sm;
  uint16_t* Yb[2] = {planes + 6 * sm, planes + 8 * sm};
  uint16_t* Zb[2] = {planes + 7 * sm, planes + 9 * sm};
  float* T0 = f32ws;
  float* Q0 = f32ws + nn;
  float* Tf = f32ws + 2 * nn;

  int mv_blocks = (n + 3) / 4;
  if (mv_blocks > kPartRow) mv_blocks = kPartRow;
  for (int j = 0; j < power_steps; ++j)
    ns_matvec_kernel<<<mv_blocks, 256, 0, stream>>>(C, n, vec + (j & 1) * n, vec + ((j + 1) & 1) * n, scal, j);
  ns_scale_planes_kernel<NSP><<<evox::stream_grid(nn, 256), 256, 0, stream>>>(C, n, ldp, Y0, plane, T0, Q0, scal, vec,
                                                                             power_steps, safe, mv_blocks);
  int e = evox::last_launch_status();
  if (e) return e;
  {
    // A2 = Y0 Y0 and q = Q0 + c4 A2 from one product
    auto epi = affine<NSP>(n, ldp);
    epi.nout = 2;
    epi.o[0].alpha = 1.f;
    set_planes<NSP>(epi.o[0], 0, A2, plane);
    epi.o[1].alpha = kP4c4;
    epi.o[1].beta = 1.f;
    epi.o[1].cin[0] = Q0;
    set_planes<NSP>(epi.o[1], 0, Qp, plane);
    e = ns_product<NSP, NSP>(Y0, nullptr, Y0, nullptr, 1, plane, ldp, n, epi, stream);
    if (e) return e;
  }
  {
    // p = T0 + A2 q   (= Z1); when the fifth-order step is the whole factorisation, C^{-1/2} = p / sqrt(s) leaves here
    auto epi = affine<NSP>(n, ldp);
    epi.o[0].alpha = 1.f;
    epi.o[0].beta = 1.f;
    epi.o[0].cin[0] = T0;
    set_planes<NSP>(epi.o[0], 0, Pp, plane);
    if (iters == 1) {
      epi.nout = 2;
      epi.o[1].alpha = 1.f;
      epi.o[1].beta = 1.f;
      epi.o[1].cin[0] = T0;
      epi.o[1].f32[0] = out_inv;
      epi.o[1].scale[0] = scal + 11;
    }
    e = ns_product<NSP, NSP>(A2, nullptr, Qp, nullptr, 1, plane, ldp, n, epi, stream);
    if (e) return e;
  }
  if (iters == 1) {
    // sqrt C = sqrt(s) Y0 p, then the probe residual and the guard
    auto epi = affine<1>(n, ldp);
    epi.o[0].alpha = 1.f;
    epi.o[0].f32[0] = out_sqrt;
    epi.o[0].scale[0] = scal + 12;
    epi.o[0].pl[0][0] = out_sqrt_bf16;
    e = ns_product<NSP, 1>(Y0, nullptr, Pp, nullptr, 1, plane, ldp, n, epi, stream);
    if (e) return e;
    float* w = Tf;                                        // n floats of scratch
    ns_probe_mv_kernel<<<mv_blocks, 256, 0, stream>>>(out_sqrt, n, nullptr, w, scal, 0);
    ns_probe_mv_kernel<<<mv_blocks, 256, 0, stream>>>(out_inv, n, w, nullptr, scal, 1);
    ns_probe_guard_kernel<<<evox::stream_grid(nn, 256), 256, 0, stream>>>(scal, mv_blocks, prev_sqrt, prev_inv, out_sqrt,
                                                                         out_inv, out_sqrt_bf16, n, ldp);
    return evox::last_launch_status();
  }
  {
    // Y1 = Y0 p
    auto epi = affine<NSP>(n, ldp);
    epi.o[0].alpha = 1.f;
    set_planes<NSP>(epi.o[0], 0, Y1, plane);
    e = ns_product<NSP, NSP>(Y0, nullptr, Pp, nullptr, 1, plane, ldp, n, epi, stream);
    if (e) return e;
  }
  const uint16_t* Y = Y1;
  const uint16_t* Z = Pp;
  if (full) {
    // Y1 and Z1 = p are polynomials in Y0 computed by mirrored products: symmetric to the last bit, their own transposes.
    const uint16_t *Yt = Y1, *Zt = Pp;
    uint16_t* Tt = Y0;                                    // Y0 is dead after the first step
    uint16_t* set_a[4] = {Yb[0], Zb[0], Yb[1], Zb[1]};    // Y, Yt, Z, Zt of the odd iterations
    uint16_t* set_b[4] = {A2, Qp, Pp, Y1};                // ... of the even ones (free from the second iteration on)
    for (int it = 1; it < iters; ++it) {
      const bool last = it + 1 == iters;
      {
        // T = 1.5 I - 0.5 Z Y, both orientations
        auto epi = affine<NSP>(n, ldp);
        epi.o[0].alpha = -0.5f;
        epi.o[0].gamma = 1.5f;
        set_planes<NSP>(epi.o[0], 0, Tp, plane);
        set_planes_t<NSP>(epi.o[0], 0, Tt, plane);
        if (last) epi.o[0].f32[0] = Tf;
        e = ns_product_full<NSP, NSP>(Z, nullptr, Yt, nullptr, 1, plane, ldp, n, epi, stream);
        if (e) return e;
      }
      if (!last) {
        uint16_t** o = (it & 1) ? set_a : set_b;
        auto epi = affine<NSP>(n, ldp);
        epi.o[0].alpha = 1.f;
        set_planes<NSP>(epi.o[0], 0, o[0], plane);
        set_planes_t<NSP>(epi.o[0], 0, o[1], plane);
        set_planes<NSP>(epi.o[0], 1, o[2], plane);
        set_planes_t<NSP>(epi.o[0], 1, o[3], plane);
        e = ns_product_full<NSP, NSP>(Y, Tp, Tt, Zt, 2, plane, ldp, n, epi, stream);   // Y T and T Z
        if (e) return e;
        Y = o[0]; Yt = o[1]; Z = o[2]; Zt = o[3];
      } else {
        int res_blocks = evox::stream_grid((int64_t)((n + kResStride - 1) / kResStride) * n, 256);
        if (res_blocks > kPartRow) res_blocks = kPartRow;
        ns_residual_kernel<<<res_blocks, 256, 0, stream>>>(Tf, n, scal);
        ns_flags_kernel<<<1, 256, 0, stream>>>(scal, n, res_blocks, prev_sqrt != nullptr && prev_inv != nullptr);
        auto epi = affine<1>(n, ldp);
        epi.o[0].alpha = 1.f;
        epi.o[0].f32[0] = out_sqrt;
        epi.o[0].f32[1] = out_inv;
        epi.o[0].scale[0] = scal + 12;
        epi.o[0].scale[1] = scal + 11;
        epi.o[0].prev[0] = prev_sqrt;
        epi.o[0].prev[1] = prev_inv;
        epi.o[0].pl[0][0] = out_sqrt_bf16;
        epi.keep = scal + 15;
        e = ns_product<NSP, 1>(Y, Tp, Tt, Zt, 2, plane, ldp, n, epi, stream);          // upper tiles of Y T and T Z, mirrored
        if (e) return e;
      }
    }
    return 0;
  }
  for (int it = 1; it < iters; ++it) {
    const bool last = it + 1 == iters;
    {
      // T = 1.5 I - 0.5 Z Y (planes for the next products; fp32 for the residual of the last one)
      auto epi = affine<NSP>(n, ldp);
      epi.o[0].alpha = -0.5f;
      epi.o[0].gamma = 1.5f;
      set_planes<NSP>(epi.o[0], 0, Tp, plane);
      if (last) epi.o[0].f32[0] = Tf;
      e = ns_product<NSP, NSP>(Z, nullptr, Y, nullptr, 1, plane, ldp, n, epi, stream);
      if (e) return e;
    }
    if (!last) {
      auto epi = affine<NSP>(n, ldp);
      epi.o[0].alpha = 1.f;
      set_planes<NSP>(epi.o[0], 0, Yb[it & 1], plane);
      set_planes<NSP>(epi.o[0], 1, Zb[it & 1], plane);
      e = ns_product<NSP, NSP>(Y, Tp, Tp, Z, 2, plane, ldp, n, epi, stream);      // Y T and T Z
      if (e) return e;
      Y = Yb[it & 1];
      Z = Zb[it & 1];
    } else {
      int res_blocks = evox::stream_grid((int64_t)((n + kResStride - 1) / kResStride) * n, 256);
      if (res_blocks > kPartRow) res_blocks = kPartRow;
      ns_residual_kernel<<<res_blocks, 256, 0, stream>>>(Tf, n, scal);
      ns_flags_kernel<<<1, 256, 0, stream>>>(scal, n, res_blocks, prev_sqrt != nullptr && prev_inv != nullptr);
      // sqrt C = sqrt(s) Y T, C^{-1/2} = Z T / sqrt(s): fp32 images, and the square root once more rounded to bf16
      auto epi = affine<1>(n, ldp);
      epi.o[0].alpha = 1.f;
      epi.o[0].f32[0] = out_sqrt;
      epi.o[0].f32[1] = out_inv;
      epi.o[0].scale[0] = scal + 12;
      epi.o[0].scale[1] = scal + 11;
      epi.o[0].prev[0] = prev_sqrt;
      epi.o[0].prev[1] = prev_inv;
      epi.o[0].pl[0][0] = out_sqrt_bf16;
      epi.keep = scal + 15;
      e = ns_product<NSP, 1>(Y, Tp, Tp, Z, 2, plane, ldp, n, epi, stream);        // Y T and T Z
      if (e) return e;
    }
  }
  return 0;
}

// cond(C) from both sides after a factorisation: with Z = C^{-1/2}, |Z|_inf^2 >= 1 / lambda_min (a bound) and
// |Z v|^2 <= 1 / lambda_min for the unit vector v carried between calls (an estimate that improves as v turns
// towards the smallest eigenvector).  scal[9] = s |Z v|^2, scal[10] = s |Z|_inf^2 with s = scal[13] ~ lambda_max.
__global__ __launch_bounds__(256) void ns_cond_kernel(float* scal, const float* scal2, int mv_blocks, float* vec2, int n) {
  __shared__ float red[4];
  const float inf_norm = block_reduce(part_row(scal2, kRowInf), mv_blocks, true, red);
  const float sq = block_reduce(part_row(scal2, 0), mv_blocks, false, red);
  if (threadIdx.x == 0) {
    scal[9] = scal[13] * sq;
    scal[10] = scal[13] * inf_norm * inf_norm;
  }
  const float inv = rsqrtf(fmaxf(sq, 1e-38f));
  for (int j = threadIdx.x; j < n; j += blockDim.x) vec2[j] = vec2[n + j] * inv;
}

}  // namespace

// Condition estimates of the matrix whose inverse square root is `inv` (see ns_cond_kernel).  vec2: 2n floats, the
// first n a unit vector carried between calls; scal: the block of the factorisation that produced `inv`;
// scal2: a second block of evox_hip_ns_scal_floats() floats (scratch).
EVOX_API int evox_hip_ns_cond(const float* inv, int n, float* vec2, float* scal, float* scal2, void* stream) {
  if (n <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  int mv_blocks = (n + 3) / 4;
  if (mv_blocks > kPartRow) mv_blocks = kPartRow;
  ns_matvec_kernel<<<mv_blocks, 256, 0, st>>>(inv, n, vec2, vec2 + n, scal2, 0);
  ns_cond_kernel<<<1, 256, 0, st>>>(scal, scal2, mv_blocks, vec2, n);
  return evox::last_launch_status();
}

// Work space of evox_hip_ns_sqrt_bf16: planes = 10 * nsp * n * ldp bf16 elements (zero-filled once by the caller: the
// pad columns are never written), f32ws = 3 * n * n floats.  ldp: a multiple of 64 elements >= n rounded up to 64.
// out_sqrt_bf16 (optional): n x ldp, pad columns zero.  iters >= 1 (the first is the fifth-order step; with
// iters == 1 it is the whole factorisation and the residual is estimated with a probe vector).
// full != 0: plain (un-mirrored) products in the iterations after the first step, for ill-conditioned matrices.
EVOX_API int evox_hip_ns_sqrt_planes(const float* C, int n, int64_t ldp, int nsp, int iters, int power_steps, int safe,
                                     int full, void* planes, float* f32ws, float* vec, float* scal, const float* prev_sqrt,
                                     const float* prev_inv, float* out_sqrt, float* out_inv, void* out_sqrt_bf16,
                                     void* stream) {
  if (n <= 0 || iters < 1 || power_steps < 1 || power_steps > kMaxPow || (nsp != 2 && nsp != 3)) return (int)hipErrorInvalidValue;
  if (ldp < (int64_t)((n + 63) / 64) * 64 || (ldp & 7) || ((uintptr_t)planes & 15) || ((uintptr_t)out_sqrt_bf16 & 15))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  if (nsp == 2)
    return ns_sqrt_planes<2>(C, n, ldp, iters, power_steps, safe, full, (uint16_t*)planes, f32ws, vec, scal, prev_sqrt,
                             prev_inv, out_sqrt, out_inv, (uint16_t*)out_sqrt_bf16, st);
  return ns_sqrt_planes<3>(C, n, ldp, iters, power_steps, safe, full, (uint16_t*)planes, f32ws, vec, scal, prev_sqrt,
                           prev_inv, out_sqrt, out_inv, (uint16_t*)out_sqrt_bf16, st);
}

EVOX_API int evox_hip_ns_sqrt_bf16(const float* C, int n, int64_t ldp, int nsp, int iters, int power_steps, int safe,
                                   void* planes, float* f32ws, float* vec, float* scal, const float* prev_sqrt,
                                   const float* prev_inv, float* out_sqrt, float* out_inv, void* out_sqrt_bf16,
                                   void* stream) {
  return evox_hip_ns_sqrt_planes(C, n, ldp, nsp, iters, power_steps, safe, 0, planes, f32ws, vec, scal, prev_sqrt, prev_inv,
                                 out_sqrt, out_inv, out_sqrt_bf16, stream);
}
