// Reference-vector-guided selection (RVEA) without n × nv intermediates.
//
// Every solution is associated with the reference vector of smallest angle, scored by the angle-penalised
// distance  APD = (1 + m θ · angle / γ_v) · ‖f‖,  and every vector keeps its best-scored solution.  With
// tensor operations that is an n × nv cosine matrix, its arc cosine, a membership mask and a masked
// arg-min; here one pass streams LDS tiles of the unit reference vectors past each solution (registers:
// its translated objectives, best cosine and index), and the per-vector arg-min is a 64-bit atomic
// minimum on (APD bits, solution index).  γ (each vector's smallest angle to another vector) comes from
// the same streaming pattern.
//
// Layout of a workgroup: 256 threads = 64 rows × 4 slices of every 256-vector tile (LDS broadcasts).
#include "hip_common.h"

using namespace evox;

namespace {

// rows: (n, M) translated objectives (NaN rows are skipped) or unit vectors; vn: (nv, M) unit vectors.
// SELF: rows are the vectors themselves — skip j == i and write gamma = smallest angle to another vector.
template <int M, bool SELF>
__global__ __launch_bounds__(256) void rvea_assoc_kernel(const float* __restrict__ rows, int n, const float* __restrict__ vn,
                                                         int nv, const float* __restrict__ gamma, float m_theta,
                                                         const float* __restrict__ theta_dev,
                                                         float* __restrict__ gamma_out,
                                                         unsigned long long* __restrict__ best) {
  __shared__ float tile[M][256];
  __shared__ float p_c[4][64];
  __shared__ int p_i[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;
  float mine[M];
  float norm2 = 0.f;
  bool bad = i >= n;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    mine[k] = i < n ? rows[(int64_t)i * M + k] : 0.f;
    bad = bad || (mine[k] != mine[k]);
    norm2 += mine[k] * mine[k];
  }
  const float inv_norm = 1.0f / sqrtf(norm2);
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] *= inv_norm;
  // angles through the chord between unit vectors, 2·asin(‖a − b‖ / 2): exact for the small angles between
  // neighbouring reference vectors, where the arc cosine of a rounded dot product is not
  float best_d = INFINITY; int best_j = 0x7fffffff;
  for (int j0 = 0; j0 < nv; j0 += 256) {
    __syncthreads();
    const int jj = j0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = jj < nv ? vn[(int64_t)jj * M + k] : 0.f;
    __syncthreads();
    const int lim = min(64, nv - j0 - s * 64);
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) { const float d = mine[k] - tile[k][c]; d2 += d * d; }
      d2 = fminf(d2, 2.0f);                               // angles beyond a right angle count as one; NaN → 2
      if ((!SELF || j0 + c != i) && d2 < best_d) { best_d = d2; best_j = j0 + c; }
    }
  }
  p_c[s][r] = best_d; p_i[s][r] = best_j;
  __syncthreads();
  if (s != 0 || i >= n) return;
  for (int t = 1; t < 4; ++t)
    if (p_c[t][r] < best_d || (p_c[t][r] == best_d && p_i[t][r] < best_j)) { best_d = p_c[t][r]; best_j = p_i[t][r]; }
  if (SELF) {
    gamma_out[i] = best_j == 0x7fffffff ? 0.f : 2.0f * asinf(0.5f * sqrtf(best_d));
    return;
  }
  if (bad || best_j == 0x7fffffff) return;
  const float angle = 2.0f * asinf(0.5f * sqrtf(best_d));
  // the penalty weight M·θ comes as a launch argument or — θ on the device, e.g. computed from a generation counter that a
  // recorded graph advances — is read here
  const float mt = theta_dev ? (float)M * theta_dev[0] : m_theta;
  float apd = (1.0f + mt * angle / gamma[best_j]) * sqrtf(norm2);
  if (apd != apd) apd = INFINITY;
  // non-negative floats order like their bit patterns; ties go to the smaller solution index
  const unsigned long long key = ((unsigned long long)__float_as_uint(fmaxf(apd, 0.f)) << 32) | (unsigned int)i;
  atomicMin(&best[best_j], key);
}

// Association of points with reference directions by perpendicular distance (NSGA-III niching, theta-DEA clustering):
// for point p and unit direction v, c = clamp(<p, v> / |p|, -1, 1), d1 = |p| c (along v), d2 = |p| sqrt(1 - c^2)
// (off v).  Every point gets the direction with the smallest d2 (ties: the lowest index), its d1 and d2; a point with a
// NaN coordinate gets direction 0 and d2 = +inf.  The tensor formulation materialises five n x nv matrices.
template <int M>
__global__ __launch_bounds__(256) void ref_assoc_kernel(const float* __restrict__ rows, int n, const float* __restrict__ vn,
                                                        int nv, int64_t* __restrict__ group, float* __restrict__ d1_out,
                                                        float* __restrict__ d2_out) {
  __shared__ float tile[M][256];
  __shared__ float p_k[4][64], p_c[4][64];
  __shared__ int p_i[4][64];
  const int r = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;
  float mine[M];
  float norm2 = 0.f;
  bool bad = i >= n;
#pragma unroll
  for (int k = 0; k < M; ++k) {
    mine[k] = i < n ? rows[(int64_t)i * M + k] : 0.f;
    bad = bad || (mine[k] != mine[k]);
    norm2 += mine[k] * mine[k];
  }
  const float norm = sqrtf(norm2);
  const float inv_norm = 1.0f / norm;
#pragma unroll
  for (int k = 0; k < M; ++k) mine[k] *= inv_norm;
  float best_key = INFINITY, best_c = 0.f;
  int best_j = 0x7fffffff;
  for (int j0 = 0; j0 < nv; j0 += 256) {
    __syncthreads();
    const int jj = j0 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = jj < nv ? vn[(int64_t)jj * M + k] : 0.f;
    __syncthreads();
    const int lim = min(64, nv - j0 - s * 64);
    for (int q = 0; q < lim; ++q) {
      const int c = s * 64 + q;
      // through the chord between the unit vectors, h = |p^ - v|^2 / 2 = 1 - c: 1 - c^2 = h (2 - h) without the
      // cancellation that 1 - c*c suffers for the small angles that matter here
      float ch = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) { const float d = mine[k] - tile[k][c]; ch = fmaf(d, d, ch); }
      const float h = fminf(0.5f * ch, 2.f);
      const float key = fmaxf(h * (2.f - h), 0.f);
      if (key < best_key) { best_key = key; best_c = 1.f - h; best_j = j0 + c; }
    }
  }
  p_k[s][r] = best_key; p_c[s][r] = best_c; p_i[s][r] = best_j;
  __syncthreads();
  if (s != 0 || i >= n) return;
  for (int t = 1; t < 4; ++t)
    if (p_k[t][r] < best_key || (p_k[t][r] == best_key && p_i[t][r] < best_j)) {
      best_key = p_k[t][r]; best_c = p_c[t][r]; best_j = p_i[t][r];
    }
  if (bad || best_j == 0x7fffffff || !(norm == norm)) {
    group[i] = 0;
    d1_out[i] = __builtin_nanf("");
    d2_out[i] = INFINITY;
    return;
  }
  group[i] = best_j;
  d1_out[i] = norm * best_c;
  const float d2 = norm * sqrtf(best_key);
  d2_out[i] = d2 == d2 ? d2 : INFINITY;                    // the zero vector: 0 * (1 / 0) has no direction
}

template <int M>
int launch_assoc(const float* rows, int n, const float* vn, int nv, int64_t* group, float* d1, float* d2, hipStream_t st) {
  ref_assoc_kernel<M><<<(n + 63) / 64, 256, 0, st>>>(rows, n, vn, nv, group, d1, d2);
  return last_launch_status();
}

template <int M>
int launch(const float* obj, int n, const float* vn, int nv, float m_theta, const float* theta_dev, float* gamma,
           unsigned long long* best, hipStream_t st) {
  rvea_assoc_kernel<M, true><<<(nv + 63) / 64, 256, 0, st>>>(vn, nv, vn, nv, nullptr, 0.f, nullptr, gamma, nullptr);
  rvea_assoc_kernel<M, false><<<(n + 63) / 64, 256, 0, st>>>(obj, n, vn, nv, gamma, m_theta, theta_dev, nullptr, best);
  return last_launch_status();
}

}  // namespace

// obj: (n, m) objectives minus the ideal point, clamped positive, NaN rows allowed; vn: (nv, m) unit reference
// vectors; gamma: scratch (nv); best: (nv) 64-bit words preset to all ones — on return the low 32 bits of
// best[v] hold the selected solution, all ones where no solution is associated with v.
static int rvea_select_any(const float* obj, int n, const float* vn, int nv, int m, float theta, const float* theta_dev,
                           float* gamma, unsigned long long* best, void* stream) {
  if (n <= 0 || nv <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const float mt = (float)m * theta;
  switch (m) {
    case 2: return launch<2>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 3: return launch<3>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 4: return launch<4>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 5: return launch<5>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 6: return launch<6>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 7: return launch<7>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    case 8: return launch<8>(obj, n, vn, nv, mt, theta_dev, gamma, best, st);
    default: return (int)hipErrorInvalidValue;
  }
}

// rows: (n, m) points; vn: (nv, m) unit directions; group / d1 / d2: out (n).
EVOX_API int evox_hip_ref_associate(const float* rows, int n, const float* vn, int nv, int m, int64_t* group, float* d1,
                                    float* d2, void* stream) {
  if (n <= 0) return 0;
  if (nv <= 0) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  switch (m) {
    case 2: return launch_assoc<2>(rows, n, vn, nv, group, d1, d2, st);
    case 3: return launch_assoc<3>(rows, n, vn, nv, group, d1, d2, st);
    case 4: return launch_assoc<4>(rows, n, vn, nv, group, d1, d2, st);
    case 5: return launch_assoc<5>(rows, n, vn, nv, group, d1, d2, st);
    case 6: return launch_assoc<6>(rows, n, vn, nv, group, d1, d2, st);
    case 7: return launch_assoc<7>(rows, n, vn, nv, group, d1, d2, st);
    case 8: return launch_assoc<8>(rows, n, vn, nv, group, d1, d2, st);
    default: return (int)hipErrorInvalidValue;
  }
}

EVOX_API int evox_hip_rvea_select(const float* obj, int n, const float* vn, int nv, int m, float theta, float* gamma,
                                  unsigned long long* best, void* stream) {
  return rvea_select_any(obj, n, vn, nv, m, theta, nullptr, gamma, best, stream);
}

// The same with the angle-penalty weight θ read from the device (one float).
EVOX_API int evox_hip_rvea_select_dev(const float* obj, int n, const float* vn, int nv, int m, const float* theta,
                                      float* gamma, unsigned long long* best, void* stream) {
  if (theta == nullptr) return (int)hipErrorInvalidValue;
  return rvea_select_any(obj, n, vn, nv, m, 0.f, theta, gamma, best, stream);
}
