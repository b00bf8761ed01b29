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

#include <mutex>
#include <set>
#include <utility>

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

// ---------------------------------------------------------------------------------------------------------------------
// Symmetric POSITIVE DEFINITE eigendecomposition: one-sided Jacobi on the Cholesky factor.
//
// C = L L^T; right rotations make the columns of W = L J1 J2 ... orthogonal; at convergence W = U S with the eigenvectors
// of C in U (the normalised columns) and the eigenvalues S^2 (the squared column norms).  Against rotating W = C V and
// accumulating V this moves HALF the data per rotation (no V) and needs 7-8 sweeps whatever the conditioning (the plain
// form needs 11-19 between cond 1e2 and 1e6).  A direction along which C is singular to working precision comes back
// with eigenvalue 0 and a zero vector (see cholesky_lds).
// With the eigenvectors V0 of a nearby matrix the same is done for A = V0^T C V0, which is nearly diagonal — three
// sweeps for a covariance matrix that moved by a few percent — and the result is V0 U.
//
// Layout: column-major in LDS with a pitch that is a multiple of four floats (16-byte reads of a column) and not of 32
// (the Cholesky step reads along rows).  A column pair belongs to 8 lanes; a lane keeps its 4·NV elements of both columns
// in registers between the three dot products and the rotation.
// ---------------------------------------------------------------------------------------------------------------------

// sum over the 8 lanes of a group (every lane receives it): three DPP adds
__device__ __forceinline__ float oct_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
  return v;
}

__host__ __device__ inline int spd_pitch(int n) {
  int ld = (n + 3) & ~3;
  if ((ld & 31) == 0) ld += 4;
  return ld;
}

// In-place lower Cholesky factor, left-looking: X[j*ld + i] (i > j) holds the lower triangle of the matrix on entry and of
// the factor on exit; the diagonal of the matrix is in diag[] (left alone), the factor's is written to X[j*ld + j].  Row i
// of column j is the work of one group of 8 lanes: its dot product with row j over the finished columns, and — each
// group for itself, from the values it reads anyway — the pivot.  One barrier per column.
// STRICT: a non-positive pivot gives NaNs from there on.  Otherwise a pivot that has vanished to rounding (below 1e-6 of
// its diagonal entry: the matrix is singular to working precision along that direction) makes its column ZERO, which is
// what the column of a semi-definite matrix is in exact arithmetic; dividing its rounding noise by a floored pivot
// instead would feed the noise, amplified, into every later column.
template <bool STRICT>
__device__ void cholesky_lds(float* X, const float* diag, int n, int ld) {
  const int g = threadIdx.x >> 3, l = threadIdx.x & 7;
  for (int j = 0; j < n; ++j) {
    for (int i0 = j; i0 < n; i0 += kLT / 8) {
      const int i = i0 + g;
      if (i0 + (g & ~7) >= n) continue;                     // a wavefront (8 groups) without a row: straight to the barrier
      const bool live = i < n;
      const int ii = live ? i : j;
      float aij = 0.f, ajj = 0.f;
      for (int t = l; t < j; t += 32) {                     // four independent pairs of reads in flight
        float lj[4], li[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int tt = t + 8 * u;
          lj[u] = tt < j ? X[tt * ld + j] : 0.f;
          li[u] = tt < j ? X[tt * ld + ii] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          aij = fmaf(li[u], lj[u], aij);
          ajj = fmaf(lj[u], lj[u], ajj);
        }
      }
      aij = oct_sum(aij);
      ajj = oct_sum(ajj);
      const float djj = diag[j], piv = djj - ajj;
      const bool keep = STRICT || piv > 1e-6f * djj;
      const float d = keep ? sqrtf(piv) : 0.f;
      if (live && l == 0) X[j * ld + i] = i == j ? d : (keep ? __fdividef(X[j * ld + i] - aij, d) : 0.f);
    }
    __syncthreads();
  }
}

struct SpdArgs {
  const float* C;      // batch x (n x n), symmetric
  const float* V0;     // batch x (n x n) starting eigenvectors, vector j at V0[j*n ..], or nullptr
  float* V;            // batch x (n x n) eigenvectors, same layout
  float* evals;        // batch x n, in the order of the vectors (not sorted)
  float* scratch;      // batch x (n x n), used with V0
  int* sweeps;         // batch sweep counts, or nullptr
  int n, max_sweeps;
  float tol;
};

template <int NV>      // 16-byte pieces of a column per lane: pitch <= 32 * NV
__global__ __launch_bounds__(kLT) void eigh_spd_kernel(SpdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[kLW];
  __shared__ float s_bcast;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.n, ld = spd_pitch(n), b = blockIdx.x;
  const float* C = a.C + (long long)b * n * n;
  const float* V0 = a.V0 != nullptr ? a.V0 + (long long)b * n * n : nullptr;
  float* Vout = a.V + (long long)b * n * n;
  float* X = lds;                      // n columns of pitch ld
  float* Y = lds + n * ld;             // V0, the same layout (warm start only)
  float* diag = V0 != nullptr ? Y + n * ld : X + n * ld;
  float* nrm = diag + n;
  const int nb4 = (n + 3) >> 2;

  if (V0 != nullptr) {
    float* S = a.scratch + (long long)b * n * n;
    for (int e = tid; e < n * ld; e += kLT) {
      const int j = e / ld, i = e - j * ld;
      Y[e] = i < n ? V0[j * n + i] : 0.f;
    }
    __syncthreads();
    // T = C V0 into X: column block (4 vectors) x row i per work item; C is read along its rows (C = C^T), coalesced
    for (int e = tid; e < n * nb4; e += kLT) {
      const int jb = e / n, i = e - jb * n, j0 = jb * 4;
      const float* y0 = Y + j0 * ld;
      const float* y1 = Y + min(j0 + 1, n - 1) * ld;
      const float* y2 = Y + min(j0 + 2, n - 1) * ld;
      const float* y3 = Y + min(j0 + 3, n - 1) * ld;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int t = 0; t < ld; t += 4) {          // rows n .. ld-1 of Y are zero; C is not read there
        const float4 v0 = *reinterpret_cast<const float4*>(y0 + t), v1 = *reinterpret_cast<const float4*>(y1 + t);
        const float4 v2 = *reinterpret_cast<const float4*>(y2 + t), v3 = *reinterpret_cast<const float4*>(y3 + t);
        const float c0 = t < n ? C[t * n + i] : 0.f;
        const float c1 = t + 1 < n ? C[(t + 1) * n + i] : 0.f;
        const float c2 = t + 2 < n ? C[(t + 2) * n + i] : 0.f;
        const float c3 = t + 3 < n ? C[(t + 3) * n + i] : 0.f;
        a0 = fmaf(c0, v0.x, a0); a0 = fmaf(c1, v0.y, a0); a0 = fmaf(c2, v0.z, a0); a0 = fmaf(c3, v0.w, a0);
        a1 = fmaf(c0, v1.x, a1); a1 = fmaf(c1, v1.y, a1); a1 = fmaf(c2, v1.z, a1); a1 = fmaf(c3, v1.w, a1);
        a2 = fmaf(c0, v2.x, a2); a2 = fmaf(c1, v2.y, a2); a2 = fmaf(c2, v2.z, a2); a2 = fmaf(c3, v2.w, a2);
        a3 = fmaf(c0, v3.x, a3); a3 = fmaf(c1, v3.y, a3); a3 = fmaf(c2, v3.z, a3); a3 = fmaf(c3, v3.w, a3);
      }
      X[j0 * ld + i] = a0;
      if (j0 + 1 < n) X[(j0 + 1) * ld + i] = a1;
      if (j0 + 2 < n) X[(j0 + 2) * ld + i] = a2;
      if (j0 + 3 < n) X[(j0 + 3) * ld + i] = a3;
    }
    for (int e = tid; e < n * (ld - n); e += kLT) {        // the padding rows of X
      const int j = e / (ld - n), i = n + e - j * (ld - n);
      X[j * ld + i] = 0.f;
    }
    __syncthreads();
    // A = V0^T T, lower triangle, to the scratch matrix: S[q*n + p] = v_p . t_q
    for (int e = tid; e < n * nb4; e += kLT) {
      const int qb = e / n, p = e - qb * n, q0 = qb * 4;
      if (p < q0) continue;
      const float* yp = Y + p * ld;
      const float* x0 = X + q0 * ld;
      const float* x1 = X + min(q0 + 1, n - 1) * ld;
      const float* x2 = X + min(q0 + 2, n - 1) * ld;
      const float* x3 = X + min(q0 + 3, n - 1) * ld;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int t = 0; t < ld; t += 4) {
        const float4 y = *reinterpret_cast<const float4*>(yp + t);
        const float4 v0 = *reinterpret_cast<const float4*>(x0 + t), v1 = *reinterpret_cast<const float4*>(x1 + t);
        const float4 v2 = *reinterpret_cast<const float4*>(x2 + t), v3 = *reinterpret_cast<const float4*>(x3 + t);
        a0 = fmaf(y.x, v0.x, a0); a0 = fmaf(y.y, v0.y, a0); a0 = fmaf(y.z, v0.z, a0); a0 = fmaf(y.w, v0.w, a0);
        a1 = fmaf(y.x, v1.x, a1); a1 = fmaf(y.y, v1.y, a1); a1 = fmaf(y.z, v1.z, a1); a1 = fmaf(y.w, v1.w, a1);
        a2 = fmaf(y.x, v2.x, a2); a2 = fmaf(y.y, v2.y, a2); a2 = fmaf(y.z, v2.z, a2); a2 = fmaf(y.w, v2.w, a2);
        a3 = fmaf(y.x, v3.x, a3); a3 = fmaf(y.y, v3.y, a3); a3 = fmaf(y.z, v3.z, a3); a3 = fmaf(y.w, v3.w, a3);
      }
      S[q0 * n + p] = a0;
      if (q0 + 1 < n) S[(q0 + 1) * n + p] = a1;
      if (q0 + 2 < n) S[(q0 + 2) * n + p] = a2;
      if (q0 + 3 < n) S[(q0 + 3) * n + p] = a3;
    }
    __threadfence_block();
    __syncthreads();                                        // S is complete and visible to the workgroup
    for (int e = tid; e < n * ld; e += kLT) {
      const int j = e / ld, i = e - j * ld;
      X[e] = i > j && i < n ? S[j * n + i] : 0.f;
      if (i == j) diag[j] = S[j * n + j];
    }
  } else {
    for (int e = tid; e < n * ld; e += kLT) {
      const int j = e / ld, i = e - j * ld;
      X[e] = i > j && i < n ? 0.5f * (C[i * n + j] + C[j * n + i]) : 0.f;
      if (i == j) diag[j] = C[j * n + j];
    }
  }
  __syncthreads();
  cholesky_lds<false>(X, diag, n, ld);

  // ---- one-sided Jacobi on the columns of the factor ----
  const int K = (n + 1) & ~1;
  const int pr = tid >> 3, l = tid & 7;
  int sweep = 0;
  for (; sweep < a.max_sweeps; ++sweep) {
    float worst = 0.f;
    for (int step = 0; step < K - 1; ++step) {
      if (pr < K / 2) {
        int p = step + pr, q = step + K - 1 - pr;           // both below 2 (K - 1): one conditional subtraction each
        p = p >= K - 1 ? p - (K - 1) : p;
        q = q >= K - 1 ? q - (K - 1) : q;
        if (pr == 0) { p = K - 1; q = step; }
        if (p < n && q < n) {
          float* wp = X + p * ld;
          float* wq = X + q * ld;
          float4 xs[NV], ys[NV];
          float al = 0.f, be = 0.f, ga = 0.f;
#pragma unroll
          for (int u = 0; u < NV; ++u) {
            const int i = 4 * (l + 8 * u);
            if (i < ld) {
              xs[u] = *reinterpret_cast<const float4*>(wp + i);
              ys[u] = *reinterpret_cast<const float4*>(wq + i);
            } else {
              xs[u] = make_float4(0.f, 0.f, 0.f, 0.f);
              ys[u] = xs[u];
            }
            al = fmaf(xs[u].x, xs[u].x, al); al = fmaf(xs[u].y, xs[u].y, al); al = fmaf(xs[u].z, xs[u].z, al); al = fmaf(xs[u].w, xs[u].w, al);
            be = fmaf(ys[u].x, ys[u].x, be); be = fmaf(ys[u].y, ys[u].y, be); be = fmaf(ys[u].z, ys[u].z, be); be = fmaf(ys[u].w, ys[u].w, be);
            ga = fmaf(xs[u].x, ys[u].x, ga); ga = fmaf(xs[u].y, ys[u].y, ga); ga = fmaf(xs[u].z, ys[u].z, ga); ga = fmaf(xs[u].w, ys[u].w, ga);
          }
          al = oct_sum(al);
          be = oct_sum(be);
          ga = oct_sum(ga);
          const float prod = al * be;
          const float rel = prod > 0.f ? fabsf(ga) * rsqrtf(prod) : 0.f;
          worst = fmaxf(worst, rel);
          if (rel > a.tol) {
            const float zeta = __fdividef(be - al, 2.f * ga);
            const float t = __fdividef(copysignf(1.f, zeta), fabsf(zeta) + sqrtf(fmaf(zeta, zeta, 1.f)));
            const float c = rsqrtf(fmaf(t, t, 1.f)), s = c * t;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
              const int i = 4 * (l + 8 * u);
              if (i < ld) {
                float4 x = xs[u], y = ys[u], xo, yo;
                xo.x = fmaf(c, x.x, -s * y.x); yo.x = fmaf(s, x.x, c * y.x);
                xo.y = fmaf(c, x.y, -s * y.y); yo.y = fmaf(s, x.y, c * y.y);
                xo.z = fmaf(c, x.z, -s * y.z); yo.z = fmaf(s, x.z, c * y.z);
                xo.w = fmaf(c, x.w, -s * y.w); yo.w = fmaf(s, x.w, c * y.w);
                *reinterpret_cast<float4*>(wp + i) = xo;
                *reinterpret_cast<float4*>(wq + i) = yo;
              }
            }
          }
        }
      }
      __syncthreads();
    }
    worst = wave_max(worst);
    if (lane == 0) red[wave] = worst;
    __syncthreads();
    if (tid == 0) {
      float w = 0.f;
      for (int v = 0; v < kLW; ++v) w = fmaxf(w, red[v]);
      s_bcast = w;
    }
    __syncthreads();
    // quadratic convergence: a sweep whose worst pair was below sqrt(tol) leaves nothing above tol behind
    if (!(s_bcast * s_bcast > a.tol)) { ++sweep; break; }
  }
  if (a.sweeps != nullptr && tid == 0) a.sweeps[b] = sweep;

  // ---- results: eigenvalue = squared norm of a column, eigenvector = the normalised column (times V0) ----
  float* ev = a.evals + (long long)b * n;
  for (int j = pr; j < n; j += kLT / 8) {
    const float* wj = X + j * ld;
    float acc = 0.f;
    for (int i = 4 * l; i < ld; i += 32) {
      const float4 x = *reinterpret_cast<const float4*>(wj + i);
      acc = fmaf(x.x, x.x, acc); acc = fmaf(x.y, x.y, acc); acc = fmaf(x.z, x.z, acc); acc = fmaf(x.w, x.w, acc);
    }
    acc = oct_sum(acc);
    if (l == 0) {
      ev[j] = acc;
      nrm[j] = acc > 0.f ? rsqrtf(acc) : 0.f;
    }
  }
  __syncthreads();
  if (V0 == nullptr) {
    for (int e = tid; e < n * n; e += kLT) {
      const int j = e / n, i = e - j * n;
      Vout[e] = X[j * ld + i] * nrm[j];
    }
    return;
  }
  // V[j][i] = sum_t V0[t][i] u_j[t]: four vectors j per work item, lanes along i
  for (int e = tid; e < n * nb4; e += kLT) {
    const int jb = e / n, i = e - jb * n, j0 = jb * 4;
    const float* x0 = X + j0 * ld;
    const float* x1 = X + min(j0 + 1, n - 1) * ld;
    const float* x2 = X + min(j0 + 2, n - 1) * ld;
    const float* x3 = X + min(j0 + 3, n - 1) * ld;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int t = 0; t < n; ++t) {
      const float y = Y[t * ld + i];
      a0 = fmaf(y, x0[t], a0);
      a1 = fmaf(y, x1[t], a1);
      a2 = fmaf(y, x2[t], a2);
      a3 = fmaf(y, x3[t], a3);
    }
    Vout[j0 * n + i] = a0 * nrm[j0];
    if (j0 + 1 < n) Vout[(j0 + 1) * n + i] = a1 * nrm[j0 + 1];
    if (j0 + 2 < n) Vout[(j0 + 2) * n + i] = a2 * nrm[j0 + 2];
    if (j0 + 3 < n) Vout[(j0 + 3) * n + i] = a3 * nrm[j0 + 3];
  }
}

// Lower Cholesky factor with the working matrix in LDS (row-major in and out, upper part zeroed): cholesky_lds above.
// A non-positive pivot turns the rest of the factor into NaNs (as the vendor routine reports an error).
__global__ __launch_bounds__(kLT) void cholesky_lds_kernel(const float* __restrict__ Ain, float* __restrict__ Lout, int n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, ld = spd_pitch(n);
  const float* A = Ain + (long long)blockIdx.x * n * n;
  float* L = Lout + (long long)blockIdx.x * n * n;
  float* X = lds;
  float* diag = lds + n * ld;
  for (int e = tid; e < n * ld; e += kLT) {
    const int j = e / ld, i = e - j * ld;
    X[e] = i > j && i < n ? 0.5f * (A[i * n + j] + A[j * n + i]) : 0.f;
    if (i == j) diag[j] = A[j * n + j];
  }
  __syncthreads();
  cholesky_lds<true>(X, diag, n, ld);
  for (int e = tid; e < n * n; e += kLT) {
    const int i = e / n, j = e - i * n;
    L[e] = j <= i ? X[j * ld + i] : 0.f;
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

// raises the dynamic LDS limit of a kernel once per device and kernel
int configure_ptr(const void* kern) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kern})) return 0;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (kLdsFloats + 1024) * 4);
  if (e != hipSuccess) return (int)e;
  done.insert({dev, kern});
  return 0;
}
template <class K>
int configure(K kern) { return configure_ptr(reinterpret_cast<const void*>(kern)); }

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

// The same for symmetric POSITIVE DEFINITE C by one-sided Jacobi on the Cholesky factor (eigh_spd_kernel); scratch:
// batch x n x n, touched only with V0; sweeps: batch ints (the sweeps taken) or null.  hipErrorInvalidValue when the
// working matrices do not fit the LDS (n > 190, or n > 134 with V0): the caller then takes evox_hip_small_eigh.
EVOX_API int evox_hip_small_eigh_spd_fits(int n, int warm) {
  const long long need = (long long)(warm ? 2 : 1) * n * spd_pitch(n) + 2ll * n;
  return n > 0 && spd_pitch(n) <= 192 && need <= kLdsFloats ? 1 : 0;
}

EVOX_API int evox_hip_small_eigh_spd(const float* C, const float* V0, float* V, float* evals, float* scratch, int* sweeps,
                                     int n, int batch, int max_sweeps, float tol, void* stream) {
  if (batch <= 0 || max_sweeps <= 0 || C == nullptr || V == nullptr || evals == nullptr ||
      (V0 != nullptr && scratch == nullptr) || !evox_hip_small_eigh_spd_fits(n, V0 != nullptr))
    return (int)hipErrorInvalidValue;
  SpdArgs a{C, V0, V, evals, scratch, sweeps, n, max_sweeps, tol};
  if (spd_pitch(n) <= 128) {
    int e = configure(eigh_spd_kernel<4>);
    if (e) return e;
    eigh_spd_kernel<4><<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(a);
  } else {
    int e = configure(eigh_spd_kernel<6>);
    if (e) return e;
    eigh_spd_kernel<6><<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(a);
  }
  return last_launch_status();
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
  if ((long long)n * spd_pitch(n) + n <= kLdsFloats) {
    int e = configure(cholesky_lds_kernel);
    if (e) return e;
    cholesky_lds_kernel<<<batch, kLT, kLdsBytes, (hipStream_t)stream>>>(A, L, n);
    return last_launch_status();
  }
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
