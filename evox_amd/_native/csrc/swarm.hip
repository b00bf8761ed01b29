// Particle-swarm update as two launches (three for long swarms):
//   1. pso_best_kernel  – one workgroup scans the swarm: new personal-best fitness, arg-min,
//                         and whether the global best improves.  Beyond 32 768 particles the scan is
//                         spread over up to 1024 workgroups (pso_best_part_kernel) whose partial minima
//                         one workgroup merges (pso_best_final_kernel): one workgroup streams 10⁶
//                         fitness pairs in about a millisecond, as long as the whole move takes.
//   2. pso_move_kernel  – elementwise over N x D: personal-best rows, velocity and position
//                         with two Philox uniforms per coordinate generated in registers
//                         (element e of the logical (2, N, D) draw), clamping to the bounds.  With a
//                         dimension divisible by four (and 16-byte aligned operands) a lane moves four
//                         coordinates of one row with 16-byte loads and stores and one row division.
#include <algorithm>
#include <cstdint>

#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

constexpr int kBestParts = 1024;          // most workgroups of the spread scan; ws holds 2 + 2·kBestParts words
constexpr int64_t kBestSpread = 32768;    // swarms beyond this size take the spread scan

__device__ inline void best_merge(float& best, int& best_i, float ov, int oi) {
  if (ov < best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
}

// The (minimum, lowest index) of a workgroup's lanes, valid in thread 0.
__device__ inline void best_of_workgroup(float& best, int& best_i, float* s_val, int* s_idx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    best_merge(best, best_i, ov, oi);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = best_i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int w = 1; w < nw; ++w) best_merge(best, best_i, s_val[w], s_idx[w]);
  }
}

__device__ inline void best_publish(float best, int best_i, const float* gbest_fit, float* o_gfit, int* ws) {
  const float g = gbest_fit[0];
  const int take = (best_i != 0x7fffffff) && (best < g);
  ws[0] = best_i == 0x7fffffff ? 0 : best_i;
  ws[1] = take;
  o_gfit[0] = take ? best : g;
}

__global__ __launch_bounds__(1024) void pso_best_kernel(const float* __restrict__ fit, const float* __restrict__ pbest_fit,
                                                        const float* __restrict__ gbest_fit, float* __restrict__ o_pfit,
                                                        float* __restrict__ o_gfit, int* __restrict__ ws, int64_t n) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  float best = INFINITY;
  int best_i = 0x7fffffff;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float f = fit[i], p = pbest_fit[i];
    const float m = f < p ? f : p;
    o_pfit[i] = m;
    best_merge(best, best_i, m, (int)i);
  }
  best_of_workgroup(best, best_i, s_val, s_idx);
  if (threadIdx.x == 0) best_publish(best, best_i, gbest_fit, o_gfit, ws);
}

__global__ __launch_bounds__(256) void pso_best_part_kernel(const float* __restrict__ fit, const float* __restrict__ pbest_fit,
                                                            float* __restrict__ o_pfit, int* __restrict__ ws, int64_t n) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  float best = INFINITY;
  int best_i = 0x7fffffff;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float f = fit[i], p = pbest_fit[i];
    const float m = f < p ? f : p;
    o_pfit[i] = m;
    best_merge(best, best_i, m, (int)i);
  }
  best_of_workgroup(best, best_i, s_val, s_idx);
  if (threadIdx.x == 0) {
    ws[2 + blockIdx.x] = __float_as_int(best);
    ws[2 + kBestParts + blockIdx.x] = best_i;
  }
}

__global__ __launch_bounds__(1024) void pso_best_final_kernel(const float* __restrict__ gbest_fit, float* __restrict__ o_gfit,
                                                              int* __restrict__ ws, int parts) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  float best = INFINITY;
  int best_i = 0x7fffffff;
  if ((int)threadIdx.x < parts) {
    best = __int_as_float(ws[2 + threadIdx.x]);
    best_i = ws[2 + kBestParts + threadIdx.x];
    if (best_i == 0x7fffffff) best = INFINITY;
  }
  best_of_workgroup(best, best_i, s_val, s_idx);
  if (threadIdx.x == 0) best_publish(best, best_i, gbest_fit, o_gfit, ws);
}

template <bool VEC>
__global__ __launch_bounds__(256) void pso_move_kernel(
    const int32_t* __restrict__ key, const float* __restrict__ pop, const float* __restrict__ vel,
    const float* __restrict__ fit, const float* __restrict__ pbest, const float* __restrict__ pbest_fit,
    const float* __restrict__ gbest, const float* __restrict__ lb, const float* __restrict__ ub,
    const int* __restrict__ ws, float* __restrict__ o_pop, float* __restrict__ o_vel, float* __restrict__ o_pbest,
    float* __restrict__ o_gbest, int64_t n, int d, float w, float phi_p, float phi_g) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const int64_t total = n * (int64_t)d;
  const int64_t nquad = (total + 3) >> 2;
  const int best_i = ws[0];
  const bool take = ws[1] != 0;
  const bool best_improved = fit[best_i] < pbest_fit[best_i];
  const float* best_row = (best_improved ? pop : pbest) + (int64_t)best_i * d;
  if constexpr (VEC) {
    // d % 4 == 0: the four coordinates of a quad lie in one row, and the second random tensor starts on a block boundary
    const bool narrow = total < ((int64_t)1 << 31);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e = 4 * q;
      const int64_t row = narrow ? (int64_t)((uint32_t)e / (uint32_t)d) : e / d;
      const int col = (int)(e - row * d);
      const f32x4 rp = unit4(philox_block(k0, k1, (uint64_t)q, kStreamDraw));
      const f32x4 rg = unit4(philox_block(k0, k1, (uint64_t)(total + e) >> 2, kStreamDraw));
      const float4 x = *reinterpret_cast<const float4*>(pop + e);
      const float4 v = *reinterpret_cast<const float4*>(vel + e);
      const bool better = fit[row] < pbest_fit[row];
      float4 pb = x;
      if (!better) pb = *reinterpret_cast<const float4*>(pbest + e);
      const float4 gcur = *reinterpret_cast<const float4*>((take ? best_row : gbest) + col);
      const float4 lo = *reinterpret_cast<const float4*>(lb + col);
      const float4 hi = *reinterpret_cast<const float4*>(ub + col);
      float4 nv, nx;
      nv.x = w * v.x + phi_p * lane_of(rp, 0) * (pb.x - x.x) + phi_g * lane_of(rg, 0) * (gcur.x - x.x);
      nv.y = w * v.y + phi_p * lane_of(rp, 1) * (pb.y - x.y) + phi_g * lane_of(rg, 1) * (gcur.y - x.y);
      nv.z = w * v.z + phi_p * lane_of(rp, 2) * (pb.z - x.z) + phi_g * lane_of(rg, 2) * (gcur.z - x.z);
      nv.w = w * v.w + phi_p * lane_of(rp, 3) * (pb.w - x.w) + phi_g * lane_of(rg, 3) * (gcur.w - x.w);
      nx.x = fminf(fmaxf(x.x + nv.x, lo.x), hi.x);
      nx.y = fminf(fmaxf(x.y + nv.y, lo.y), hi.y);
      nx.z = fminf(fmaxf(x.z + nv.z, lo.z), hi.z);
      nx.w = fminf(fmaxf(x.w + nv.w, lo.w), hi.w);
      *reinterpret_cast<float4*>(o_pbest + e) = pb;
      *reinterpret_cast<float4*>(o_vel + e) = nv;
      *reinterpret_cast<float4*>(o_pop + e) = nx;
      if (row == 0) *reinterpret_cast<float4*>(o_gbest + col) = gcur;
    }
    return;
  }
  const int shift = (int)(total & 3);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 rp = unit4(philox_block(k0, k1, (uint64_t)q, kStreamDraw));
    // the second random tensor starts at element `total` of the same stream
    const uint64_t b2 = (uint64_t)(total + 4 * q) >> 2;
    const f32x4 ga = unit4(philox_block(k0, k1, b2, kStreamDraw));
    f32x4 gb = ga;
    if (shift) gb = unit4(philox_block(k0, k1, b2 + 1, kStreamDraw));
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int64_t e = 4 * q + l;
      if (e >= total) break;
      const int64_t row = e / d;
      const int col = (int)(e - row * d);
      const int l2 = l + shift;
      const float r_p = lane_of(rp, l);
      const float r_g = l2 < 4 ? lane_of(ga, l2) : lane_of(gb, l2 - 4);
      const float x = pop[e], v = vel[e];
      const bool better = fit[row] < pbest_fit[row];
      const float pb = better ? x : pbest[e];
      const float gcur = take ? best_row[col] : gbest[col];
      const float nv = w * v + phi_p * r_p * (pb - x) + phi_g * r_g * (gcur - x);
      const float lo = lb[col], hi = ub[col];
      float nx = fminf(fmaxf(x + nv, lo), hi);
      o_pbest[e] = pb;
      o_vel[e] = nv;
      o_pop[e] = nx;
      if (row == 0) o_gbest[col] = gcur;
    }
  }
}

}  // namespace

EVOX_API int evox_hip_pso_step(const int32_t* key, const float* pop, const float* vel, const float* fit,
                               const float* pbest, const float* pbest_fit, const float* gbest, const float* gbest_fit,
                               const float* lb, const float* ub, float* o_pop, float* o_vel, float* o_pbest,
                               float* o_pfit, float* o_gbest, float* o_gfit, int* ws, int64_t n, int d, float w,
                               float phi_p, float phi_g, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n > kBestSpread) {
    const int parts = (int)std::min<int64_t>(kBestParts, (n + 1023) / 1024);
    pso_best_part_kernel<<<parts, 256, 0, st>>>(fit, pbest_fit, o_pfit, ws, n);
    pso_best_final_kernel<<<1, 1024, 0, st>>>(gbest_fit, o_gfit, ws, parts);
  } else {
    pso_best_kernel<<<1, 1024, 0, st>>>(fit, pbest_fit, gbest_fit, o_pfit, o_gfit, ws, n);
  }
  const int64_t nquad = (n * (int64_t)d + 3) >> 2;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = d % 4 == 0 && aligned(pop) && aligned(vel) && aligned(pbest) && aligned(gbest) && aligned(lb) &&
                   aligned(ub) && aligned(o_pop) && aligned(o_vel) && aligned(o_pbest) && aligned(o_gbest);
  if (vec)
    pso_move_kernel<true><<<stream_grid(nquad, 256), 256, 0, st>>>(key, pop, vel, fit, pbest, pbest_fit, gbest, lb, ub, ws,
                                                                  o_pop, o_vel, o_pbest, o_gbest, n, d, w, phi_p, phi_g);
  else
    pso_move_kernel<false><<<stream_grid(nquad, 256), 256, 0, st>>>(key, pop, vel, fit, pbest, pbest_fit, gbest, lb, ub, ws,
                                                                   o_pop, o_vel, o_pbest, o_gbest, n, d, w, phi_p, phi_g);
  return last_launch_status();
}

// Competitive swarm optimiser: one pass over the random pairs.  perm[0:half] meets perm[half:2·half];
// the loser of each pair learns from the winner,
//     v' = r1·v + r2·(x_w − x_l) + phi·r3·(mean − x_l),   x' = clip(x_l + v'),
// with r1, r2, r3 the Philox uniforms of key at elements (k·half + pair)·dim + c (the layout of the tensor
// formulation's (3, half, dim) draw).  With write_all the winners' rows are copied through, so new_pop / new_vel
// need no initialisation when every particle is in a pair (an even swarm); without it they are copies of the
// inputs made by the caller and only the losers' rows are rewritten.  A dimension divisible by four moves four
// coordinates per lane (16-byte accesses, one Philox block per random tensor instead of four).
// Outputs: cand (half × dim, the losers' new positions) and students (the losers' row numbers).
namespace {
template <bool VEC>
__global__ __launch_bounds__(256) void cso_step_kernel(const int32_t* __restrict__ key, const float* __restrict__ pop,
                                                       const float* __restrict__ vel, const float* __restrict__ fit,
                                                       const int64_t* __restrict__ perm, const float* __restrict__ center,
                                                       float phi, const float* __restrict__ lb, const float* __restrict__ ub,
                                                       int half, int dim, int write_all, float* __restrict__ new_pop,
                                                       float* __restrict__ new_vel, float* __restrict__ cand,
                                                       int64_t* __restrict__ students) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  if constexpr (VEC) {
    const int qpr = dim >> 2;                                        // quads per row
    const int64_t nquad = (int64_t)half * qpr;
    const bool narrow = nquad < ((int64_t)1 << 31);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
      const int r = narrow ? (int)((uint32_t)q / (uint32_t)qpr) : (int)(q / qpr);
      const int c = (int)(q - (int64_t)r * qpr) * 4;
      const int64_t a = perm[r], b = perm[half + r];
      const bool first_wins = fit[a] < fit[b];
      const int64_t w = first_wins ? a : b, l = first_wins ? b : a;
      auto uni4 = [&](int k) {
        const uint64_t idx = ((uint64_t)k * half + r) * dim + c;      // a multiple of four
        return unit4(philox_block(k0, k1, idx >> 2, kStreamDraw));
      };
      const float4 xl = *reinterpret_cast<const float4*>(pop + l * dim + c);
      const float4 vl = *reinterpret_cast<const float4*>(vel + l * dim + c);
      const float4 xw = *reinterpret_cast<const float4*>(pop + w * dim + c);
      const float4 lo = *reinterpret_cast<const float4*>(lb + c);
      const float4 hi = *reinterpret_cast<const float4*>(ub + c);
      const f32x4 u0 = uni4(0), u1 = uni4(1);
      float4 v, x;
      v.x = u0.x * vl.x + u1.x * (xw.x - xl.x);
      v.y = u0.y * vl.y + u1.y * (xw.y - xl.y);
      v.z = u0.z * vl.z + u1.z * (xw.z - xl.z);
      v.w = u0.w * vl.w + u1.w * (xw.w - xl.w);
      if (phi != 0.f) {
        const f32x4 u2 = uni4(2);
        const float4 m = *reinterpret_cast<const float4*>(center + c);
        v.x += phi * u2.x * (m.x - xl.x);
        v.y += phi * u2.y * (m.y - xl.y);
        v.z += phi * u2.z * (m.z - xl.z);
        v.w += phi * u2.w * (m.w - xl.w);
      }
      x.x = fminf(fmaxf(xl.x + v.x, lo.x), hi.x);
      x.y = fminf(fmaxf(xl.y + v.y, lo.y), hi.y);
      x.z = fminf(fmaxf(xl.z + v.z, lo.z), hi.z);
      x.w = fminf(fmaxf(xl.w + v.w, lo.w), hi.w);
      *reinterpret_cast<float4*>(new_pop + l * dim + c) = x;
      *reinterpret_cast<float4*>(new_vel + l * dim + c) = v;
      *reinterpret_cast<float4*>(cand + (int64_t)r * dim + c) = x;
      if (write_all) {
        *reinterpret_cast<float4*>(new_pop + w * dim + c) = xw;
        *reinterpret_cast<float4*>(new_vel + w * dim + c) = *reinterpret_cast<const float4*>(vel + w * dim + c);
      }
      if (c == 0) students[r] = l;
    }
    return;
  }
  const int64_t total = (int64_t)half * dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / dim), c = (int)(e - (int64_t)r * dim);
    const int64_t a = perm[r], b = perm[half + r];
    const bool first_wins = fit[a] < fit[b];
    const int64_t w = first_wins ? a : b, l = first_wins ? b : a;
    auto uni = [&](int k) {
      const uint64_t idx = ((uint64_t)k * half + r) * dim + c;
      const u32x4 blk = philox_block(k0, k1, idx >> 2, kStreamDraw);
      return bits_to_unit(lane_of(blk, (int)(idx & 3)));
    };
    const float xl = pop[l * dim + c], xw = pop[w * dim + c];
    float v = uni(0) * vel[l * dim + c] + uni(1) * (xw - xl);
    if (phi != 0.f) v += phi * uni(2) * (center[c] - xl);
    const float x = fminf(fmaxf(xl + v, lb[c]), ub[c]);
    new_pop[l * dim + c] = x;
    new_vel[l * dim + c] = v;
    cand[e] = x;
    if (write_all) {
      new_pop[w * dim + c] = xw;
      new_vel[w * dim + c] = vel[w * dim + c];
    }
    if (c == 0) students[r] = l;
  }
}
}  // namespace

EVOX_API int evox_hip_cso_step(const int32_t* key, const float* pop, const float* vel, const float* fit,
                               const int64_t* perm, const float* center, float phi, const float* lb, const float* ub,
                               int half, int dim, int write_all, float* new_pop, float* new_vel, float* cand,
                               int64_t* students, void* stream) {
  if (half <= 0 || dim <= 0) return 0;
  if (phi != 0.f && center == nullptr) return (int)hipErrorInvalidValue;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = dim % 4 == 0 && aligned(pop) && aligned(vel) && aligned(lb) && aligned(ub) && aligned(new_pop) &&
                   aligned(new_vel) && aligned(cand) && (phi == 0.f || aligned(center));
  if (vec)
    cso_step_kernel<true><<<stream_grid((int64_t)half * (dim / 4), 256), 256, 0, (hipStream_t)stream>>>(
        key, pop, vel, fit, perm, center, phi, lb, ub, half, dim, write_all, new_pop, new_vel, cand, students);
  else
    cso_step_kernel<false><<<stream_grid((int64_t)half * dim, 256), 256, 0, (hipStream_t)stream>>>(
        key, pop, vel, fit, perm, center, phi, lb, ub, half, dim, write_all, new_pop, new_vel, cand, students);
  return last_launch_status();
}

// The move shared by the swarm variants (CLPSO, SL-PSO, DMS-PSO-EL, small-world PSO, FS-PSO's elite half): up to three
// attractors, each a table row chosen per particle, each weighted by a constant and a Philox uniform,
//     v' = s_i * ( a * u0 * v  +  sum_k (c_k * u_k) * (T_k[row_k(i)][c] - x) ),   x' = clip(x + v'),
// u_k = element offset_k + i * dim + c of the stream of key_k (the layout of the tensor formulation's draws; a null
// key stands for the factor 1), row_k(i) = i, index_k[i] or 0 (one row for all), s_i an optional per-particle factor
// (constriction).  One launch in place of the dozen elementwise launches of the tensor formulation, same random numbers.
namespace {

constexpr int kMaxAttract = 3;

struct AttractTerm {
  const float* table;
  const int64_t* index;
  const int32_t* key;
  long long offset;
  int mode;                                            // 0: row i, 1: row index[i], 2: row 0
  float coef;
};

struct Attract {
  const float* x; const float* v; const float* lb; const float* ub; const float* row_scale;
  float* ox; float* ov;
  const int32_t* inertia_key;
  long long inertia_offset;
  long long n;
  int dim, nterms;
  float inertia;
  AttractTerm t[kMaxAttract];
};

__device__ __forceinline__ float unit_at(const int32_t* key, uint64_t e) {
  const u32x4 blk = philox_block((uint32_t)key[0], (uint32_t)key[1], e >> 2, kStreamDraw);
  return bits_to_unit(lane_of(blk, (int)(e & 3)));
}

__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
__device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
// elements e … e + 3 of a key's uniform stream, e a multiple of four: one Philox block
__device__ __forceinline__ void units4_at(const int32_t* key, uint64_t e, float (&o)[4]) {
  const f32x4 u = unit4(philox_block((uint32_t)key[0], (uint32_t)key[1], e >> 2, kStreamDraw));
  o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = u.w;
}

// VEC: the dimension and every stream offset are multiples of four and all operands 16-byte aligned — a lane moves four
// coordinates of one row (16-byte accesses, one Philox block per random factor instead of four).
template <bool VEC>
__global__ __launch_bounds__(256) void swarm_attract_kernel(Attract a) {
  const int64_t total = a.n * (int64_t)a.dim;
  if constexpr (VEC) {
    const int64_t nquad = total >> 2;
    const bool narrow = total < ((int64_t)1 << 31);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e = 4 * q;
      const int64_t r = narrow ? (int64_t)((uint32_t)e / (uint32_t)a.dim) : e / a.dim;
      const int c = (int)(e - r * a.dim);
      float x[4], v[4], nv[4], u[4], lo[4], hi[4];
      load4(a.x + e, x);
      load4(a.v + e, v);
      if (a.inertia_key) {
        units4_at(a.inertia_key, (uint64_t)(a.inertia_offset + e), u);
#pragma unroll
        for (int j = 0; j < 4; ++j) nv[j] = (a.inertia * u[j]) * v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) nv[j] = a.inertia * v[j];
      }
#pragma unroll
      for (int k = 0; k < kMaxAttract; ++k) {
        if (k < a.nterms) {
          const AttractTerm& t = a.t[k];
          const int64_t row = t.mode == 0 ? r : (t.mode == 1 ? t.index[r] : 0);
          float T[4];
          load4(t.table + row * a.dim + c, T);
          if (t.key) {
            units4_at(t.key, (uint64_t)(t.offset + e), u);
#pragma unroll
            for (int j = 0; j < 4; ++j) nv[j] = nv[j] + (t.coef * u[j]) * (T[j] - x[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) nv[j] = nv[j] + t.coef * (T[j] - x[j]);
          }
        }
      }
      if (a.row_scale) {
        const float sc = a.row_scale[r];
#pragma unroll
        for (int j = 0; j < 4; ++j) nv[j] = sc * nv[j];
      }
      load4(a.lb + c, lo);
      load4(a.ub + c, hi);
      store4(a.ov + e, nv);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = fminf(fmaxf(x[j] + nv[j], lo[j]), hi[j]);
      store4(a.ox + e, x);
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / a.dim;
    const int c = (int)(e - r * a.dim);
    const float x = a.x[e];
    float w = a.inertia;
    if (a.inertia_key) w = w * unit_at(a.inertia_key, (uint64_t)(a.inertia_offset + e));
    float nv = w * a.v[e];
#pragma unroll
    for (int k = 0; k < kMaxAttract; ++k) {
      if (k < a.nterms) {
        const AttractTerm& t = a.t[k];
        const int64_t row = t.mode == 0 ? r : (t.mode == 1 ? t.index[r] : 0);
        float f = t.coef;
        if (t.key) f = f * unit_at(t.key, (uint64_t)(t.offset + e));
        nv = nv + f * (t.table[row * a.dim + c] - x);
      }
    }
    if (a.row_scale) nv = a.row_scale[r] * nv;
    a.ov[e] = nv;
    a.ox[e] = fminf(fmaxf(x + nv, a.lb[c]), a.ub[c]);
  }
}

}  // namespace

// terms: nterms records of (table, index, key, offset, mode, coef) given as parallel arrays.
EVOX_API int evox_hip_swarm_attract(const float* x, const float* v, const float* lb, const float* ub, const float* row_scale,
                                    float* ox, float* ov, int64_t n, int dim, float inertia, const int32_t* inertia_key,
                                    int64_t inertia_offset, int nterms, const void* const* tables, const void* const* indices,
                                    const void* const* keys, const int64_t* offsets, const int* modes, const float* coefs,
                                    void* stream) {
  if (n <= 0 || dim <= 0) return 0;
  if (nterms < 0 || nterms > kMaxAttract) return (int)hipErrorInvalidValue;
  Attract a{};
  a.x = x; a.v = v; a.lb = lb; a.ub = ub; a.row_scale = row_scale; a.ox = ox; a.ov = ov;
  a.inertia_key = inertia_key; a.inertia_offset = inertia_offset; a.n = n; a.dim = dim; a.nterms = nterms; a.inertia = inertia;
  for (int k = 0; k < nterms; ++k) {
    if (tables[k] == nullptr || modes[k] < 0 || modes[k] > 2 || (modes[k] == 1 && indices[k] == nullptr))
      return (int)hipErrorInvalidValue;
    a.t[k].table = (const float*)tables[k];
    a.t[k].index = (const int64_t*)indices[k];
    a.t[k].key = (const int32_t*)keys[k];
    a.t[k].offset = offsets[k];
    a.t[k].mode = modes[k];
    a.t[k].coef = coefs[k];
  }
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = dim % 4 == 0 && aligned(x) && aligned(v) && aligned(lb) && aligned(ub) && aligned(ox) && aligned(ov) &&
             (inertia_key == nullptr || inertia_offset % 4 == 0);
  for (int k = 0; k < nterms; ++k) vec = vec && aligned(tables[k]) && (keys[k] == nullptr || offsets[k] % 4 == 0);
  if (vec)
    swarm_attract_kernel<true><<<stream_grid(n * (int64_t)(dim / 4), 256), 256, 0, (hipStream_t)stream>>>(a);
  else
    swarm_attract_kernel<false><<<stream_grid(n * (int64_t)dim, 256), 256, 0, (hipStream_t)stream>>>(a);
  return last_launch_status();
}

// Fully informed particle swarm: a particle is drawn to the randomly weighted centroid of its neighbours' personal bests,
//     pm = sum_k (w_ik * phi_ik) * pbest[nbr_ik][c] / sum_k (w_ik * phi_ik),   phi_ik = u * max_phi * mask_ik,
//     v' = chi_i * (v + phi_i * (pm - x)),   x' = clip(x + v'),
// u = element (i * K + k) * dim + c of the key's uniform stream (the (n, K, dim) draw of the tensor formulation).
namespace {
__global__ __launch_bounds__(256) void fips_move_kernel(const int32_t* __restrict__ key, const float* __restrict__ x,
                                                        const float* __restrict__ v, const float* __restrict__ pbest,
                                                        const int64_t* __restrict__ nbr, const float* __restrict__ mask,
                                                        const float* __restrict__ w, const float* __restrict__ chi,
                                                        const float* __restrict__ phi, const float* __restrict__ lb,
                                                        const float* __restrict__ ub, float* __restrict__ ox,
                                                        float* __restrict__ ov, int64_t n, int K, int dim, float max_phi) {
  const int64_t total = n * (int64_t)dim;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / dim;
    const int c = (int)(e - i * dim);
    float num = 0.f, den = 0.f;
    for (int k = 0; k < K; ++k) {
      const float ph = unit_at(key, (uint64_t)((i * K + k) * dim + c)) * max_phi * mask[i * K + k];
      const float wp = w[i * K + k] * ph;
      num += pbest[nbr[i * K + k] * dim + c] * wp;
      den += wp;
    }
    const float xv = x[e];
    const float nv = chi[i] * (v[e] + phi[i] * (num / den - xv));
    ov[e] = nv;
    ox[e] = fminf(fmaxf(xv + nv, lb[c]), ub[c]);
  }
}
}  // namespace

EVOX_API int evox_hip_fips_move(const int32_t* key, const float* x, const float* v, const float* pbest, const int64_t* nbr,
                                const float* mask, const float* w, const float* chi, const float* phi, const float* lb,
                                const float* ub, float* ox, float* ov, int64_t n, int K, int dim, float max_phi, void* stream) {
  if (n <= 0 || dim <= 0) return 0;
  if (K <= 0) return (int)hipErrorInvalidValue;
  fips_move_kernel<<<stream_grid(n * (int64_t)dim, 256), 256, 0, (hipStream_t)stream>>>(key, x, v, pbest, nbr, mask, w, chi, phi,
                                                                                       lb, ub, ox, ov, n, K, dim, max_phi);
  return last_launch_status();
}
