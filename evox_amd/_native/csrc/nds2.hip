// Non-dominated sort of two objectives in ONE launch of one workgroup.
//
// With the points in lexicographic order (first objective, then second) a dominator never sits behind the point it
// dominates, and point q is dominated by an earlier point p exactly when
//     p.f1 <  q.f1  and  p.f2 <= q.f2      or      p.f1 == q.f1  and  p.f2 < q.f2 .
// So q belongs to the current front iff   A > q.f2   and   B >= q.f2 ,   A the smallest f2 among the remaining points of
// the earlier f1-groups, B the smallest among the remaining earlier points of q's own group — both fall out of one
// prefix-minimum over the remaining points: with P the exclusive prefix-minimum at q,  q is dominated iff
// A <= q.f2 or P < q.f2  (P < A means B = P; otherwise B >= A and B < q.f2 would imply A <= q.f2 anyway).  For the first
// point of a group A = P; the few others read the inclusive prefix-minimum at the end of the previous group from LDS.
// "No remaining point in front" is a NaN (a legitimate second objective may be +inf): fminf passes its other operand
// through and every comparison with it fails.
// Identical points need no special care: neither dominates the other (B < q.f2 fails for the twin), and whatever
// dominates one dominates both.
// A front is therefore one block-wide scan: hundreds of fronts cost a few microseconds each instead of three launches.
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int kThreads = 1024;

template <int E>
__global__ __launch_bounds__(kThreads) void nds2_scan_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                             int* __restrict__ gstart, int n, int target,
                                                             int* __restrict__ rk, int* __restrict__ sizes,
                                                             int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) float s_p[];          // inclusive prefix-minimum, n floats
  __shared__ float s_w[kThreads / 64];
  __shared__ int s_c[kThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = tid * E;
  float val[E];                                     // second objective of the points still unranked, NaN otherwise
  unsigned first = 0u;                              // bit e: first of its f1-group
  __shared__ int s_groups;                          // is any f1-group longer than one point?
  __shared__ int s_g[kThreads / 64];
  if (tid == 0) s_groups = 0;
  __syncthreads();
  // group starts: a running maximum of "my position if I open a group" (one block-wide scan, once)
  int open[E];
  int run_g = -1;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int q = base + e;
    val[e] = NAN;
    open[e] = -1;
    if (q < n) {
      val[e] = f2[q];
      const bool opens = q == 0 || f1[q] != f1[q - 1];
      if (opens) { first |= 1u << e; run_g = q; } else s_groups = 1;
      open[e] = run_g;
    }
  }
  {
    int inc = run_g;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc = max(inc, up);
    }
    int excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = -1;
    if (lane == 63) s_g[wave] = inc;
    __syncthreads();
    for (int w = 0; w < wave; ++w) excl = max(excl, s_g[w]);
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (base + e < n) gstart[base + e] = max(open[e], excl);
  }
  __syncthreads();
  const bool groups = s_groups != 0;                // uniform
  int ranked = 0, front = 0;
  while (ranked < target && front < n) {
    // inclusive prefix-minimum over the remaining points of this thread ("no point yet" is a NaN: fminf returns its
    // other operand, comparisons fail), and the thread's total
    float pm[E];
    float run = NAN;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      run = fminf(run, val[e]);
      pm[e] = run;
    }
    float inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(inc, o, 64);
      if (lane >= o) inc = fminf(inc, up);
    }
    float excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = NAN;
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    // everything in the waves in front of this one: lanes 0 … 15 hold the wave totals, four exchange steps fold them
    float before = (lane < kThreads / 64 && lane < wave) ? s_w[lane & (kThreads / 64 - 1)] : NAN;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) before = fminf(before, __shfl_xor(before, o, 64));
    before = __shfl(before, 0, 64);
    excl = fminf(excl, before);
    if (groups) {                                     // the look-ups below need the prefix-minimum of other threads
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (base + e < n) s_p[base + e] = fminf(excl, pm[e]);
      __syncthreads();
    }
    int count = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float v = val[e];
      const float p_excl = e == 0 ? excl : fminf(excl, pm[e - 1]);
      float a = p_excl;
      if (groups && v == v && !((first >> e) & 1u)) {
        const int gs = gstart[base + e];
        a = gs > 0 ? s_p[gs - 1] : NAN;
      }
      const bool mine = v == v && !(a <= v || p_excl < v);          // unranked and not dominated
      if (mine) {
        rk[base + e] = front;
        val[e] = NAN;
        ++count;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o, 64);
    if (lane == 0) s_c[wave] = count;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) total += s_c[w];
    if (tid == 0) sizes[front] = total;
    ++front;
    ranked += total;
    if (total == 0) break;                            // only points with NaN are left
  }
  if (tid == 0) { info[0] = front; info[1] = ranked; }
}

template <int E>
int launch(const float* f1, const float* f2, int* gstart, int n, int target, int* rk, int* sizes, int* info, hipStream_t st) {
  const size_t lds = (size_t)n * sizeof(float);
  auto kern = nds2_scan_kernel<E>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  kern<<<1, kThreads, lds, st>>>(f1, f2, gstart, n, target, rk, sizes, info);
  return last_launch_status();
}

}  // namespace

// (32 points per thread spill registers and lose to the pair-test rounds)
EVOX_API int64_t evox_hip_nds2_scan_max() { return 24 * kThreads; }

// f1, f2: the two objectives of the points in lexicographic order (no NaN).  gstart: scratch (n).  rk: out, front index
// (unranked points keep what the caller stored, e.g. -1); sizes: out (up to n), points per front; info: out (fronts,
// points ranked).  Peeling stops once `target` points are ranked.
EVOX_API int evox_hip_nds2_scan(const float* f1, const float* f2, int* gstart, int n, int target, int* rk, int* sizes, int* info,
                                void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (n <= 8 * kThreads) return launch<8>(f1, f2, gstart, n, target, rk, sizes, info, st);
  if (n <= 16 * kThreads) return launch<16>(f1, f2, gstart, n, target, rk, sizes, info, st);
  if (n <= 24 * kThreads) return launch<24>(f1, f2, gstart, n, target, rk, sizes, info, st);
  return (int)hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------
// Two objectives, any number of points and fronts: front index = length of the longest chain of dominators, computed
// like the piles of a patience sort.  With the points in lexicographic order (first objective, then second) a point q in
// front of p dominates p exactly when key(q) < key(p) for key = (second objective, first objective) compared
// lexicographically.  tails[k] = the smallest key seen so far in front k is strictly increasing in k (the smallest key of
// front k+1 belongs to a point that some earlier point of front k dominates), so the front of a point is the number of
// tails below its key — a binary search — raised by the dominators among the points handled together with it.
// One wavefront walks the points 64 at a time: 64 binary searches on the tails (LDS), one ordered pass over the 64 points
// (a point's final front is broadcast to the lanes behind it), 64 LDS minima.  O(n log fronts) comparisons instead of
// fronts * n, and no dependence on the number of fronts: ranking all of 2e5 uniform points (883 fronts) takes 11 ms
// instead of 23.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kPatienceFronts = 16384;      // tails kept in LDS (128 KiB); more fronts: the caller takes another path

__device__ __forceinline__ uint32_t ordered_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__global__ __launch_bounds__(64) void nds2_patience_kernel(const float* __restrict__ f1, const float* __restrict__ f2, int n,
                                                           int* __restrict__ rank, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tails[];
  const int lane = threadIdx.x;
  int fronts = 0;                                                    // uniform
  // the next chunk's objectives are fetched while this one is worked on
  float a_next = lane < n ? f1[lane] : 0.f, b_next = lane < n ? f2[lane] : 0.f;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool live = i < n;
    const float a = a_next, b = b_next;
    if (i + 64 < n) {
      a_next = f1[i + 64];
      b_next = f2[i + 64];
    }
    const unsigned long long key = live ? ((unsigned long long)ordered_bits(b) << 32) | ordered_bits(a) : ~0ull;
    // fronts whose smallest key lies below this one's: each of them holds a dominator among the earlier chunks
    // (branch-free: the number of tails below the key, by halving steps on a power-of-two span)
    int r = 0;
    if (fronts > 0) {
      int step = 1 << (31 - __builtin_clz(fronts));                 // largest power of two <= fronts
      for (; step > 0; step >>= 1) {
        const int probe = r + step;                                  // tails[probe - 1] < key  =>  at least `probe` tails below
        const unsigned long long t = tails[min(probe, fronts) - 1];
        r = (probe <= fronts && t < key) ? probe : r;
      }
    }
    // dominators inside the chunk, in order: when lane q is reached its front is final.  Padding lanes hold the largest
    // key and sit behind the live ones, so the pass may run over all 64 lanes.
    const uint32_t key_lo = (uint32_t)key, key_hi = (uint32_t)(key >> 32);
#pragma unroll 8
    for (int q = 0; q < 63; ++q) {
      const uint32_t qlo = __builtin_amdgcn_readlane(key_lo, q), qhi = __builtin_amdgcn_readlane(key_hi, q);
      const int rq1 = __builtin_amdgcn_readlane(r, q) + 1;
      const unsigned long long kq = ((unsigned long long)qhi << 32) | qlo;
      const bool dominated = lane > q && kq < key;
      r = dominated ? max(r, rq1) : r;
    }
    // new fronts start empty, then every point lowers the tail of its front
    // highest front of the chunk + 1 (uniform): row maxima by DPP, then the four rows
    int top = live ? r + 1 : 0;
    top = max(top, __builtin_amdgcn_update_dpp(0, top, 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
    top = max(top, __builtin_amdgcn_update_dpp(0, top, 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
    top = max(top, __builtin_amdgcn_update_dpp(0, top, 0x124, 0xF, 0xF, false));     // row_ror:4
    top = max(top, __builtin_amdgcn_update_dpp(0, top, 0x128, 0xF, 0xF, false));     // row_ror:8
    top = max(max(__builtin_amdgcn_readlane(top, 0), __builtin_amdgcn_readlane(top, 16)),
              max(__builtin_amdgcn_readlane(top, 32), __builtin_amdgcn_readlane(top, 48)));
    if (top > kPatienceFronts) {
      if (lane == 0) info[1] = 1;                                    // more fronts than the LDS holds
      return;
    }
    // (one wavefront: its LDS operations take effect in program order, so no barrier — which would also wait for the
    // stores of the ranks — only a fence for the compiler)
    for (int k = fronts + lane; k < top; k += 64) tails[k] = ~0ull;
    __builtin_amdgcn_wave_barrier();
    if (live) {
      atomicMin(&tails[r], key);
      rank[i] = r;
    }
    __builtin_amdgcn_wave_barrier();
    fronts = max(fronts, top);
  }
  if (lane == 0) info[0] = fronts;
}

}  // namespace

EVOX_API int evox_hip_nds2_patience_max_fronts() { return kPatienceFronts; }

// f1, f2: the two objectives of the points in lexicographic order (first objective, then second; no NaN, -0 folded into
// +0).  rank: out, front index of every point.  info: [0] number of fronts, [1] set when there are more fronts than
// evox_hip_nds2_patience_max_fronts() (rank is then incomplete); zeroed by the caller.
EVOX_API int evox_hip_nds2_patience(const float* f1, const float* f2, int n, int* rank, int* info, void* stream) {
  if (n <= 0) return 0;
  static bool configured[16] = {};
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
  const int lds = kPatienceFronts * (int)sizeof(unsigned long long);
  if (dev >= 0 && dev < 16 && !configured[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)nds2_patience_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  nds2_patience_kernel<<<1, 64, lds, (hipStream_t)stream>>>(f1, f2, n, rank, info);
  return last_launch_status();
}
