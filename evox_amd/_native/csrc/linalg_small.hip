// Small dense factorisations on the device, one workgroup per matrix (batched over blockIdx.x), no host involvement:
// the order-100 matrices of the evolution strategies are latency-bound in the vendor solvers (an eigendecomposition at
// n = 100 takes 1.35 ms there, longer than a copy to the host, LAPACK and a copy back), and a host round trip cannot be
// recorded into a graph.
//
//   jacobi      one-sided (Hestenes) Jacobi: right rotations that make the columns of a working matrix W orthogonal,
//               optionally accumulated into V.  Round-robin ordering: k/2 disjoint column pairs per step, dealt over the
//               sixteen wavefronts; a pair costs three dot products and the rotation of two columns.
//                 eigh   W = C V0 (V0 = previous eigenvectors, or I): at convergence V holds the eigenvectors and
//                        lambda_j = v_j . w_j.  A covariance matrix that moved by a fraction of a percent since V0 was
//                        computed needs one or two sweeps instead of eight.
//                 svd    the rows of X (k x dim) are the columns of W: at convergence their norms are the singular
//                        values and the normalised rows are V^T, written out in descending order.
//   cholesky    right-looking, lower factor.
//   qr          Q of the thin QR by classical Gram-Schmidt with re-orthogonalisation (positive diagonal of R).
//   expm        scaling and squaring with a degree-8 Taylor polynomial, all in LDS.
//
// Reference call sites: es_variants/cma_es.py:191-196 (eigh), asebo.py:76,97 (svd, cholesky), guided_es.py:76 (qr),
// nes.py:55,108 (cholesky, expm).
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int kLT = 1024;                  // threads per workgroup
constexpr int kLW = kLT / 64;
constexpr int kLdsFloats = 36 * 1024;      // 144 KiB of working storage

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kLW; ++w) t += red[w];
  return t;
}

struct JacobiArgs {
  float* W;            // batch x (k columns of length m), column-major, pitch ldw; rotated in place
  float* V;            // batch x (k x k), column-major, or nullptr
  const float* C;      // eigh: batch x (k x k) symmetric input (row- or column-major alike); W = C V0 is built first
  const float* V0;     // eigh: starting eigenvectors (column-major k x k) or nullptr (identity)
  float* evals;        // eigh: lambda_j = v_j . w_j; svd: singular values (column norms); batch x k
  float* out_rows;     // svd: batch x (k x m) normalised columns in descending order of their norm, or nullptr
  long long strideW, strideV, strideC;
  int m, k, ldw, max_sweeps;
  float tol;
};

// sum over the 16 lanes of a row of the wavefront (every lane of the row receives it): four DPP adds
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // row_ror:8
  return v;
}

// A column pair is handled by one 16-lane row of a wavefront: 64 pairs are in flight per workgroup, so a step of the
// round-robin schedule (k / 2 disjoint pairs) is one pass up to k = 128.  LDS: the working matrices live in LDS (the
// template keeps the accesses plain LDS instructions); otherwise they stay in (cache-resident) device memory.
template <bool LDS>
__global__ __launch_bounds__(kLT) void jacobi_kernel(JacobiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[kLW];
  __shared__ float s_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane >> 4, sl = lane & 15;
  const int m = a.m, k = a.k, b = blockIdx.x;
  float* Wg = a.W + (long long)b * a.strideW;
  float* Vg = a.V != nullptr ? a.V + (long long)b * a.strideV : nullptr;
  const bool eigh = a.C != nullptr;
  float* W = LDS ? lds : Wg;
  float* V = Vg == nullptr ? nullptr : (LDS ? lds + (long long)m * k : Vg);
  const int ldw = LDS ? m : a.ldw;
  if (eigh) {
    const float* C = a.C + (long long)b * a.strideC;
    const float* V0 = a.V0 != nullptr ? a.V0 + (long long)b * a.strideV : nullptr;
    for (int e = tid; e < k * k; e += kLT) {
      const int j = e / k, i = e - j * k;              // column j, row i
      float acc;
      if (V0 == nullptr) {
        acc = 0.5f * (C[(long long)i * k + j] + C[(long long)j * k + i]);
      } else {
        acc = 0.f;
        for (int t = 0; t < k; ++t) acc = fmaf(C[(long long)t * k + i], V0[(long long)j * k + t], acc);   // C = C^T
      }
      W[(long long)j * ldw + i] = acc;
      V[(long long)j * k + i] = V0 != nullptr ? V0[(long long)j * k + i] : (i == j ? 1.f : 0.f);
    }
  } else {
    if (LDS)
      for (int e = tid; e < m * k; e += kLT) {
        const int j = e / m, i = e - j * m;
        W[(long long)j * ldw + i] = Wg[(long long)j * a.ldw + i];
      }
    if (V != nullptr)
      for (int e = tid; e < k * k; e += kLT) V[e] = (e / k == e % k) ? 1.f : 0.f;
  }
  __syncthreads();
  const int K = (k + 1) & ~1;                          // positions of the round-robin schedule (one may be empty)
  for (int sweep = 0; sweep < a.max_sweeps; ++sweep) {
    float worst = 0.f;
    for (int step = 0; step < K - 1; ++step) {
      for (int pr = wave * 4 + sub; pr < K / 2; pr += kLW * 4) {
        int p = step + pr, q = step + K - 1 - pr;         // both below 2 (K - 1): one conditional subtraction each
        p = p >= K - 1 ? p - (K - 1) : p;
        q = q >= K - 1 ? q - (K - 1) : q;
        if (pr == 0) { p = K - 1; q = step; }
        const bool live = p < k && q < k;
        float* wp = W + (long long)(live ? p : 0) * ldw;
        float* wq = W + (long long)(live ? q : 0) * ldw;
        // up to kKeep elements of the two columns stay in registers between the dot products and the rotation
        constexpr int kKeep = 8;
        float xs[kKeep], ys[kKeep];
        float al = 0.f, be = 0.f, ga = 0.f;
        if (live) {
#pragma unroll
          for (int u = 0; u < kKeep; ++u) {
            const int i = sl + 16 * u;
            xs[u] = i < m ? wp[i] : 0.f;
            ys[u] = i < m ? wq[i] : 0.f;
            al = fmaf(xs[u], xs[u], al);
            be = fmaf(ys[u], ys[u], be);
            ga = fmaf(xs[u], ys[u], ga);
          }
          for (int i = sl + 16 * kKeep; i < m; i += 16) {
            const float x = wp[i], y = wq[i];
            al = fmaf(x, x, al);
            be = fmaf(y, y, be);
            ga = fmaf(x, y, ga);
          }
        }
        al = row16_sum(al);
        be = row16_sum(be);
        ga = row16_sum(ga);
        const float prod = al * be;
        const float rel = prod > 0.f ? fabsf(ga) * rsqrtf(prod) : 0.f;
        worst = fmaxf(worst, rel);
        if (live && rel > a.tol) {
          const float zeta = (be - al) / (2.f * ga);
          const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(fmaf(zeta, zeta, 1.f)));
          const float c = rsqrtf(fmaf(t, t, 1.f)), s = c * t;
#pragma unroll
          for (int u = 0; u < kKeep; ++u) {
            const int i = sl + 16 * u;
            if (i < m) {
              wp[i] = fmaf(c, xs[u], -s * ys[u]);
              wq[i] = fmaf(s, xs[u], c * ys[u]);
            }
          }
          for (int i = sl + 16 * kKeep; i < m; i += 16) {
            const float x = wp[i], y = wq[i];
            wp[i] = fmaf(c, x, -s * y);
            wq[i] = fmaf(s, x, c * y);
          }
          if (V != nullptr) {
            float* vp = V + (long long)p * k;
            float* vq = V + (long long)q * k;
            for (int i = sl; i < k; i += 16) {
              const float x = vp[i], y = vq[i];
              vp[i] = fmaf(c, x, -s * y);
              vq[i] = fmaf(s, x, c * y);
            }
          }
        }
      }
      __syncthreads();
    }
    // every wavefront's worst relative off-diagonal of the sweep: the sweep after a clean one is skipped
    worst = wave_max(worst);
    if (lane == 0) red[wave] = worst;
    __syncthreads();
    if (tid == 0) {
      float w = 0.f;
      for (int v = 0; v < kLW; ++v) w = fmaxf(w, red[v]);
      s_off = w;
    }
    __syncthreads();
    // convergence is quadratic: the rotations of a sweep whose worst pair was w leave pairs of about w^2 behind, so a
    // sweep that saw nothing above sqrt(tol) was the last one needed (a pure checking sweep would cost 40 % of a full one)
    if (!(s_off * s_off > a.tol)) break;
  }
  // ---- results ----
  float* ev = a.evals + (long long)b * k;
  for (int j = wave; j < k; j += kLW) {
    const float* wj = W + (long long)j * ldw;
    float acc = 0.f;
    if (eigh) {
      const float* vj = V + (long long)j * k;
      for (int i = lane; i < k; i += 64) acc = fmaf(vj[i], wj[i], acc);
    } else {
      for (int i = lane; i < m; i += 64) acc = fmaf(wj[i], wj[i], acc);
    }
    acc = wave_sum(acc);
    const float val = eigh ? acc : sqrtf(acc);
    if (lane == 0) ev[j] = val;
  }
  __syncthreads();
  if (eigh) {
    if (LDS)
      for (int e = tid; e < k * k; e += kLT) Vg[e] = V[e];
    return;
  }
  if (a.out_rows != nullptr) {
    // normalised columns in descending order of their norm (ties by index): the V^T of a thin SVD
    float* out = a.out_rows + (long long)b * k * m;
    for (int j = wave; j < k; j += kLW) {
      const float sj = ev[j];
      int r = 0;
      for (int t = 0; t < k; ++t) {
        const float st = ev[t];
        r += (st > sj || (st == sj && t < j)) ? 1 : 0;
      }
      const float inv = sj > 0.f ? 1.f / sj : 0.f;
      const float* wj = W + (long long)j * ldw;
      for (int i = lane; i < m; i += 64) out[(long long)r * m + i] = wj[i] * inv;
    }
    __syncthreads();
    if (tid == 0) {
      // singular values in the same (descending) order: a selection sort of at most a few hundred values
      for (int i = 0; i < k; ++i) {
        int best = i;
        for (int t = i + 1; t < k; ++t)
          if (ev[t] > ev[best]) best = t;
        const float tmp = ev[i];
        ev[i] = ev[best];
        ev[best] = tmp;
      }
    }
  } else if (LDS) {
    for (int e = tid; e < m * k; e += kLT) {
      const int j = e / m, i = e - j * m;
      Wg[(long long)j * a.ldw + i] = W[(long long)j * ldw + i];
    }
    if (Vg != nullptr)
      for (int e = tid; e < k * k; e += kLT) Vg[e] = V[e];
  }
}

// Lower Cholesky factor of a symmetric positive definite n x n matrix (row-major in and out, upper part zeroed).
// A non-positive pivot turns the rest of the factor into NaNs (as the vendor routine reports an error).
__global__ __launch_bounds__(kLT) void cholesky_kernel(const float* __restrict__ Ain, float* __restrict__ Lout, int n,
                                                       long long stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const float* A = Ain + (long long)blockIdx.x * stride;
  float* Lg = Lout + (long long)blockIdx.x * stride;
  const bool in_lds = (long long)n * n <= kLdsFloats;
  float* L = in_lds ? lds : Lg;
  for (int e = tid; e < n * n; e += kLT) {
    const int i = e / n, j = e - i * n;
    L[e] = j <= i ? 0.5f * (A[(long long)i * n + j] + A[(long long)j * n + i]) : 0.f;
  }
  __syncthreads();
  for (int c = 0; c < n; ++c) {
    const float piv = L[(long long)c * n + c];
    const float d = piv > 0.f ? sqrtf(piv) : __builtin_nanf("");
    __syncthreads();
    const float inv = 1.f / d;
    for (int i = c + tid; i < n; i += kLT) L[(long long)i * n + c] = i == c ? d : L[(long long)i * n + c] * inv;
    __syncthreads();
    // trailing update of the lower triangle: rows c+1.., columns c+1..row
    const int r = n - c - 1;
    for (int e = tid; e < r * r; e += kLT) {
      const int i = c + 1 + e / r, j = c + 1 + e % r;
      if (j <= i) L[(long long)i * n + j] = fmaf(-L[(long long)i * n + c], L[(long long)j * n + c], L[(long long)i * n + j]);
    }
    __syncthreads();
  }
  if (in_lds)
    for (int e = tid; e < n * n; e += kLT) Lg[e] = L[e];
}

// Q of the thin QR of A (m x k, k <= m), columns stored as rows: A and Q are k x m row-major (row j = column j).
// Classical Gram-Schmidt with one re-orthogonalisation; R has a positive diagonal.  A column that is linearly dependent
// on its predecessors (norm below 1e-12 of its own) becomes zero.
__global__ __launch_bounds__(kLT) void qr_kernel(const float* __restrict__ Ain, float* __restrict__ Qout, int m, int k,
                                                 long long stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[kLW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* A = Ain + (long long)blockIdx.x * stride;
  float* Q = Qout + (long long)blockIdx.x * stride;
  float* v = lds;                  // m floats: the column being orthogonalised
  float* r = lds + ((m + 63) & ~63);   // k projections
  for (int j = 0; j < k; ++j) {
    float n0 = 0.f;
    for (int i = tid; i < m; i += kLT) {
      const float x = A[(long long)j * m + i];
      v[i] = x;
      n0 = fmaf(x, x, n0);
    }
    n0 = block_sum(n0, red);
    for (int pass = 0; pass < 2; ++pass) {
      for (int q = wave; q < j; q += kLW) {
        const float* qq = Q + (long long)q * m;
        float acc = 0.f;
        for (int i = lane; i < m; i += 64) acc = fmaf(qq[i], v[i], acc);
        acc = wave_sum(acc);
        if (lane == 0) r[q] = acc;
      }
      __syncthreads();
      for (int i = tid; i < m; i += kLT) {
        float x = v[i];
        for (int q = 0; q < j; ++q) x = fmaf(-r[q], Q[(long long)q * m + i], x);
        v[i] = x;
      }
      __syncthreads();
    }
    float n1 = 0.f;
    for (int i = tid; i < m; i += kLT) n1 = fmaf(v[i], v[i], n1);
    n1 = block_sum(n1, red);
    const float inv = n1 > 1e-24f * n0 && n1 > 0.f ? rsqrtf(n1) : 0.f;
    for (int i = tid; i < m; i += kLT) Q[(long long)j * m + i] = v[i] * inv;
    __threadfence_block();
    __syncthreads();
  }
}

// exp(A) for n x n row-major A (n * n * 3 floats must fit the working storage): B = A / 2^s with s fixed at 6, Taylor
// polynomial of degree 8 by Horner, six squarings.  Accurate to single precision for |A| up to about 30.
__global__ __launch_bounds__(kLT) void expm_kernel(const float* __restrict__ Ain, float* __restrict__ Eout, int n,
                                                   long long stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int nn = n * n;
  const float* A = Ain + (long long)blockIdx.x * stride;
  float* E = Eout + (long long)blockIdx.x * stride;
  float* B = lds;
  float* P = lds + nn;
  float* T = lds + 2 * nn;
  constexpr int kSquarings = 6, kDegree = 8;
  const float sc = 1.f / (float)(1 << kSquarings);
  for (int e = tid; e < nn; e += kLT) {
    const float bv = A[e] * sc;
    B[e] = bv;
    P[e] = bv * (1.f / kDegree) + ((e / n == e % n) ? 1.f : 0.f);       // I + B / 8
  }
  __syncthreads();
  // Z = alpha X Y + diag I: a 4 x 4 block of Z per thread — per step of the inner product four (broadcast) values of X
  // and one 16-byte read of Y feed sixteen multiply-adds, a quarter of the LDS traffic of one output per thread
  const int tn = (n + 3) >> 2;
  auto matmul = [&](const float* X, const float* Y, float* Z, float alpha, float diag) {
    for (int e = tid; e < tn * tn; e += kLT) {
      const int i0 = (e / tn) * 4, j0 = (e % tn) * 4;
      float acc[4][4] = {};
      if (j0 + 3 < n && (n & 3) == 0) {
        for (int t = 0; t < n; ++t) {
          const float4 y = *reinterpret_cast<const float4*>(Y + t * n + j0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = i0 + r < n ? X[(i0 + r) * n + t] : 0.f;
            acc[r][0] = fmaf(x, y.x, acc[r][0]);
            acc[r][1] = fmaf(x, y.y, acc[r][1]);
            acc[r][2] = fmaf(x, y.z, acc[r][2]);
            acc[r][3] = fmaf(x, y.w, acc[r][3]);
          }
        }
      } else {
        for (int t = 0; t < n; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = i0 + r < n ? X[(i0 + r) * n + t] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(x, j0 + c < n ? Y[t * n + j0 + c] : 0.f, acc[r][c]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (i0 + r < n && j0 + c < n) Z[(i0 + r) * n + j0 + c] = fmaf(alpha, acc[r][c], i0 + r == j0 + c ? diag : 0.f);
    }
    __syncthreads();
  };
  // Horner: P <- I + (B P) / d for d = 7 .. 1
  float* cur = P;
  float* nxt = T;
  for (int d = kDegree - 1; d >= 1; --d) {
    matmul(B, cur, nxt, 1.f / (float)d, 1.f);
    float* t = cur; cur = nxt; nxt = t;
  }
  for (int q = 0; q < kSquarings; ++q) {
    matmul(cur, cur, nxt, 1.f, 0.f);
    float* t = cur; cur = nxt; nxt = t;
  }
  for (int e = tid; e < nn; e += kLT) E[e] = cur[e];
}

template <class K>
int configure(K kern) {
  static bool done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) return (int)hipErrorInvalidDevice;
  if (!done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (kLdsFloats + 1024) * 4);
    if (e != hipSuccess) return (int)e;
    done[dev] = true;
  }
  return 0;
}

constexpr int kLdsBytes = (kLdsFloats + 1024) * 4;

int launch_jacobi(const JacobiArgs& a, long long need_floats, int batch, hipStream_t st) {
  if (need_floats <= kLdsFloats) {
    int e = configure(jacobi_kernel<true>);
    if (e) return e;
    jacobi_kernel<true><<<batch, kLT, kLdsBytes, st>>>(a);
  } else {
    jacobi_kernel<false><<<batch, kLT, 0, st>>>(a);
  }
  return last_launch_status();
}

}  // namespace

EVOX_API int evox_hip_small_max_lds_floats() { return kLdsFloats; }

// Symmetric eigendecomposition, batched: C (batch x n x n), V0 (optional starting eigenvectors, column-major: V0[b][j][i] =
// component i of vector j) -> V (same layout), evals (batch x n, in the order of the columns of V — not sorted), W:
// batch x n x n scratch.
EVOX_API int evox_hip_small_eigh(const float* C, const float* V0, float* V, float* evals, float* W, int n, int batch,
                                 int max_sweeps, float tol, void* stream) {
  if (n <= 0 || batch <= 0 || max_sweeps <= 0 || C == nullptr || V == nullptr || W == nullptr || evals == nullptr)
    return (int)hipErrorInvalidValue;
  JacobiArgs a{};
  a.W = W; a.V = V; a.C = C; a.V0 = V0; a.evals = evals; a.out_rows = nullptr;
  a.strideW = a.strideV = a.strideC = (long long)n * n;
  a.m = a.k = n; a.ldw = n; a.max_sweeps = max_sweeps; a.tol = tol;
  return launch_jacobi(a, 2ll * n * n, batch, (hipStream_t)stream);
}

// V^T of the thin SVD of X (batch x k x m, row-major, k <= m, overwritten with the rotated rows): Vt (batch x k x m) and
// the singular values (batch x k), both in descending order.
EVOX_API int evox_hip_small_svd_vt(float* X, float* Vt, float* svals, int k, int m, int batch, int max_sweeps, float tol,
                                   void* stream) {
  if (k <= 0 || m <= 0 || batch <= 0 || X == nullptr || Vt == nullptr || svals == nullptr) return (int)hipErrorInvalidValue;
  JacobiArgs a{};
  a.W = X; a.V = nullptr; a.C = nullptr; a.V0 = nullptr; a.evals = svals; a.out_rows = Vt;
  a.strideW = (long long)k * m; a.strideV = 0; a.strideC = 0;
  a.m = m; a.k = k; a.ldw = m; a.max_sweeps = max_sweeps; a.tol = tol;
  return launch_jacobi(a, (long long)k * m, batch, (hipStream_t)stream);
}

EVOX_API int evox_hip_small_cholesky(const float* A, float* L, int n, int batch, void* stream) {
  if (n <= 0 || batch <= 0) return (int)hipErrorInvalidValue;
  int e = configure(cholesky_kernel);
  if (e) return e;
  cholesky_kernel<<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(A, L, n, (long long)n * n);
  return last_launch_status();
}

// A, Q: batch x k x m row-major, row j = column j of the m x k matrix.  m + k <= the working storage.
EVOX_API int evox_hip_small_qr_q(const float* A, float* Q, int m, int k, int batch, void* stream) {
  if (m <= 0 || k <= 0 || k > m || batch <= 0 || (long long)m + k + 64 > kLdsFloats) return (int)hipErrorInvalidValue;
  int e = configure(qr_kernel);
  if (e) return e;
  qr_kernel<<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(A, Q, m, k, (long long)m * k);
  return last_launch_status();
}

EVOX_API int evox_hip_small_expm(const float* A, float* E, int n, int batch, void* stream) {
  if (n <= 0 || batch <= 0 || 3ll * n * n > kLdsFloats) return (int)hipErrorInvalidValue;
  int e = configure(expm_kernel);
  if (e) return e;
  expm_kernel<<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(A, E, n, (long long)n * n);
  return last_launch_status();
}
