// Matrix-free non-dominated sorting.
//
//   count[i] = #{ j : j dominates i }              (nds_count)   O(N^2 m) pair tests, no N x N matrix
//   repeat:  front = { i unranked : count[i] == 0 } -> compacted, rank[i] = k           (nds_front)
//            count[i] -= #{ j in front : j dominates i } for every unranked i           (nds_peel)
//
// Each lane owns one point i (objectives in VGPRs) and walks the candidate dominators j in the same
// order as every other lane, so a candidate's objectives are wave-uniform: they are fetched with
// scalar loads into SGPRs and compared straight against the lane's VGPRs - no LDS staging and two
// or three VALU instructions per pair test.
//
// Points are sorted by ascending first objective and stored objective-major (caller).  For block b
// every candidate placed before the first row whose f0 equals the block's smallest f0 has a strictly
// smaller f0 than all of the block's points, so for those the test shrinks to the remaining m-1
// "<=" comparisons (phase A).  The short tail up to the last row with f0 <= the block's largest f0
// gets the full test (phase B).  Candidates behind that can never dominate.  Narrow problems split
// the candidate range over blockIdx.y and combine with integer atomics (exact, order-free).
#include "hip_common.h"

using namespace evox;

namespace {

constexpr int NT = 256;
constexpr int CH = 16;          // candidates fetched per batch of scalar loads

// Layout: objective-major ("SoA"), obj[k * ld + j], ld a multiple of CH.  One s_load_dwordx16 then
// brings one objective of 16 consecutive candidates into SGPRs.
template <int M, int K0, bool STRICT>
__device__ __forceinline__ bool test1(const float* __restrict__ obj, int64_t ld, int j, const float (&b)[M]) {
  bool le = true, lt = false;
#pragma unroll
  for (int k = K0; k < M; ++k) {
    const float a = obj[k * ld + j];
    le = le & (a <= b[k]);
    lt = lt | (a < b[k]);
  }
  return STRICT ? (le & lt) : le;
}

// Number of candidates j in [lo, hi) passing the test against this lane's point: all-<= over
// objectives K0..M-1, and with STRICT additionally "< in at least one".
template <int M, int K0, bool STRICT>
__device__ __forceinline__ int scan_rows(const float* __restrict__ obj, int64_t ld, int lo, int hi, const float (&b)[M]) {
  int cnt = 0;
  int j = lo;
  const int head = min(hi, (lo + CH - 1) / CH * CH);
  for (; j < head; ++j) cnt += test1<M, K0, STRICT>(obj, ld, j, b) ? 1 : 0;
  for (; j + CH <= hi; j += CH) {
    float a[M][CH];
#pragma unroll
    for (int k = K0; k < M; ++k) {
      const float4* __restrict__ src = reinterpret_cast<const float4*>(obj + k * ld + j);
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const float4 v = src[q];
        a[k][4 * q] = v.x; a[k][4 * q + 1] = v.y; a[k][4 * q + 2] = v.z; a[k][4 * q + 3] = v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      bool le = true, lt = false;
#pragma unroll
      for (int k = K0; k < M; ++k) { le = le & (a[k][u] <= b[k]); lt = lt | (a[k][u] < b[k]); }
      cnt += (STRICT ? (le & lt) : le) ? 1 : 0;
    }
  }
  for (; j < hi; ++j) cnt += test1<M, K0, STRICT>(obj, ld, j, b) ? 1 : 0;
  return cnt;
}

// first position in [lo, hi) whose key is >= v (STRICT=false) or > v (STRICT=true); keys ascending,
// NaN keys last
template <bool STRICT>
__device__ __forceinline__ int bound(const float* __restrict__ key, int lo, int hi, float v) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float k = key[mid];
    const bool before = STRICT ? (k <= v) : (k < v);
    if (before) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// obj sorted by objective 0.  count must be zeroed by the caller.
template <int M>
__global__ __launch_bounds__(NT) void nds_count_kernel(const float* __restrict__ obj, int64_t ld, int n,
                                                       int* __restrict__ count) {
  const int first = blockIdx.x * NT;
  const int last = min(n, first + NT) - 1;
  const float key_lo = obj[first], key_hi = obj[last];
  if (key_lo != key_lo) return;                       // a block of NaN keys: nothing dominates them
  // uniform bounds (every lane computes the same values; the compiler keeps them scalar)
  const int j_strict = bound<false>(obj, 0, first, key_lo);                      // f0 < every f0 of the block
  const int j_end = (key_hi != key_hi) ? n : bound<true>(obj, last + 1, n, key_hi);

  const int i = first + threadIdx.x;
  float b[M];
#pragma unroll
  for (int k = 0; k < M; ++k) b[k] = i < n ? obj[k * ld + i] : 0.f;
  const bool clean = b[0] == b[0];

  // this y-slice's share of [0, j_end), in whole batches
  int per = (j_end + gridDim.y - 1) / gridDim.y;
  per = (per + CH - 1) / CH * CH;
  const int lo = min(j_end, (int)blockIdx.y * per), hi = min(j_end, lo + per);
  const int cnt_a = scan_rows<M, 1, false>(obj, ld, lo, min(hi, j_strict), b);
  const int cnt_b = scan_rows<M, 0, true>(obj, ld, max(lo, j_strict), hi, b);
  const int cnt = (clean ? cnt_a : 0) + cnt_b;
  if (i < n && cnt) atomicAdd(&count[i], cnt);
}

// Collect the current front: unranked points whose count reached zero.  Their objectives are copied
// into a compact objective-major array so that the peel pass streams them with scalar loads.  The
// copy keeps the sorted order (two passes: per-block counts, then offsets + ballot ranks), and
// blockend[b] = number of front members among the points of blocks 0..b, which lets the peel pass
// skip every member that sits behind a block's points and drop the first-objective comparison
// for the members in front of them.
__global__ __launch_bounds__(NT) void nds_mark_kernel(const int* __restrict__ count, int* __restrict__ rank,
                                                      int* __restrict__ blockcnt, int n, int cur) {
  __shared__ int wcnt[NT / 64];
  const int i = blockIdx.x * NT + threadIdx.x;
  const bool flag = i < n && rank[i] < 0 && count[i] == 0;
  if (flag) rank[i] = cur;
  const unsigned long long bal = __ballot(flag);
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

template <int M>
__global__ __launch_bounds__(NT) void nds_compact_kernel(const float* __restrict__ obj, int64_t ld,
                                                         const int* __restrict__ rank, const int* __restrict__ blockcnt,
                                                         int* __restrict__ blockend, float* __restrict__ front_obj,
                                                         int* __restrict__ nf, int n, int cur) {
  __shared__ int red[NT / 64];
  __shared__ int wcnt[NT / 64];
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += NT) part += blockcnt[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  const int i = blockIdx.x * NT + threadIdx.x;
  const bool flag = i < n && rank[i] == cur;
  const unsigned long long bal = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = part; wcnt[wave] = __popcll(bal); }
  __syncthreads();
  const int offset = red[0] + red[1] + red[2] + red[3];
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wcnt[w];
  if (flag) {
    const int pos = offset + before + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
    for (int k = 0; k < M; ++k) front_obj[k * ld + pos] = obj[k * ld + i];
  }
  if (threadIdx.x == 0) {
    const int end = offset + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    blockend[blockIdx.x] = end;
    if (blockIdx.x == gridDim.x - 1) *nf = end;
  }
}

// A front member j and a still-unranked point i are never equal (equal points share their front),
// so "j dominates i" is the all-<= test alone.  Members are in sorted order: those before the first
// row with the block's smallest f0 need no f0 comparison, those behind the last row with the
// block's largest f0 cannot dominate.  A small population leaves most CUs idle with one block per
// 256 points, so the member range is also split over blockIdx.y (integer atomics: exact).
template <int M>
__global__ __launch_bounds__(NT) void nds_peel_kernel(const float* __restrict__ obj, int64_t ld, int n,
                                                      const float* __restrict__ front_obj,
                                                      const int* __restrict__ blockend, int* __restrict__ count,
                                                      const int* __restrict__ rank) {
  const int first = blockIdx.x * NT;
  const int last = min(n, first + NT) - 1;
  const int i = first + threadIdx.x;
  const bool live = i < n && rank[i] < 0;
  if (__syncthreads_or(live) == 0) return;            // block-uniform exit
  const float key_lo = obj[first], key_hi = obj[last];
  if (key_lo != key_lo) return;                       // a block of NaN keys: nothing dominates them
  const int j_strict = bound<false>(obj, 0, first, key_lo);
  const int j_end = (key_hi != key_hi) ? n : bound<true>(obj, last + 1, n, key_hi);
  const int m_strict = j_strict >= NT ? blockend[j_strict / NT - 1] : 0;   // members wholly in front of j_strict
  const int m_end = blockend[(j_end - 1) / NT];                           // members of every block touching [0, j_end)
  if (m_end == 0) return;
  float b[M];
#pragma unroll
  for (int k = 0; k < M; ++k) b[k] = live ? obj[k * ld + i] : -INFINITY;
  const bool clean = b[0] == b[0];
  int per = (m_end + (int)gridDim.y - 1) / (int)gridDim.y;
  per = (per + CH - 1) / CH * CH;
  const int lo = min(m_end, (int)blockIdx.y * per), hi = min(m_end, lo + per);
  if (lo >= hi) return;
  const int dec_a = scan_rows<M, 1, false>(front_obj, ld, lo, min(hi, m_strict), b);
  const int dec_b = scan_rows<M, 0, false>(front_obj, ld, max(lo, m_strict), hi, b);
  const int dec = (clean ? dec_a : 0) + dec_b;
  if (live && dec) {
    if (gridDim.y > 1) atomicSub(&count[i], dec);
    else count[i] -= dec;
  }
}

// Many fronts in a small population: a round of three launches costs its launch latencies (about 15 us), whatever
// the work.  Here ONE workgroup of 1024 threads finishes the sort from the current state: a thread keeps four
// consecutive points (objectives and remaining dominator count) in registers, the front being removed sits in LDS
// and is read by all lanes at the same address.  Per round: flags, one block scan, copy of the front into LDS (in
// index order, so sorted by the first objective like the points), then every thread tests its points against the
// members (all-<=; a member never equals an unranked point).  A wave stops at the first member whose first
// objective exceeds that of its last point, and a wave whose 256 points are all ranked skips the members
// altogether.  A ranked point's count is -1 and only falls, so it needs no branch.
constexpr int kPaThreads = 1024;
constexpr int kPaQ = 4;
constexpr int kPaMaxN = kPaThreads * kPaQ;

template <int M>
__global__ __launch_bounds__(kPaThreads) void nds_peel_all_kernel(const float* __restrict__ obj, int64_t ld, int n,
                                                                  const int* __restrict__ count, int* __restrict__ rank,
                                                                  int cur, int ranked, int target, int* __restrict__ sizes,
                                                                  int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) float s_front[];          // M x cap floats, then cap front sizes
  __shared__ int s_wave[2][kPaThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cap = (n + 3) & ~3;
  int* s_sizes = reinterpret_cast<int*>(s_front + M * cap);
  float b[kPaQ][M];
  int cnt[kPaQ], got[kPaQ];
#pragma unroll
  for (int q = 0; q < kPaQ; ++q) {
    const int i = tid * kPaQ + q;
    const bool live = i < n && rank[i] < 0;
    cnt[q] = live ? count[i] : -1;
    got[q] = -1;
#pragma unroll
    for (int k = 0; k < M; ++k) b[q][k] = live ? obj[k * ld + i] : 0.f;
  }
  // largest first objective among this wave's points (the points are sorted by it; NaN keys sort last)
  float wmax = obj[min(n - 1, (wave + 1) * 64 * kPaQ - 1)];
  if (wmax != wmax) wmax = INFINITY;
  // No global store inside the loop (a barrier would wait for it to land: microseconds per round): front indices
  // stay in registers and front sizes in LDS until the end.  Two barriers per round: the per-wave counts alternate
  // between two buffers, and a wave still reading round r's members has not reached the first barrier of round
  // r+1, behind which the next members are written.
  int f = cur;
  while (true) {
    int mine = 0;
#pragma unroll
    for (int q = 0; q < kPaQ; ++q) mine += cnt[q] == 0 ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    int* sw = s_wave[f & 1];
    if (lane == 63) sw[wave] = incl;
    __syncthreads();
    int pos = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < kPaThreads / 64; ++w) { const int c = sw[w]; total += c; if (w < wave) pos += c; }
    total = __builtin_amdgcn_readfirstlane(total);     // the same in every lane: keep it (and the loops over it) scalar
    if (total == 0) break;                             // nothing left that can be ranked
#pragma unroll
    for (int q = 0; q < kPaQ; ++q) {
      if (cnt[q] == 0) {
#pragma unroll
        for (int k = 0; k < M; ++k) s_front[k * cap + pos] = b[q][k];
        got[q] = f;
        cnt[q] = -1;
        ++pos;
      }
    }
    // pad the member list to a multiple of four with NaN (compares false against everything)
    if (tid < ((4 - (total & 3)) & 3)) s_front[total + tid] = __builtin_nanf("");
    if (tid == 0) s_sizes[f - cur] = total;
    ranked += total;
    ++f;
    if (ranked >= target) break;
    __syncthreads();
    bool any_live = false;
#pragma unroll
    for (int q = 0; q < kPaQ; ++q) any_live |= cnt[q] > 0;
    if (__ballot(any_live) == 0ull) continue;          // wave-uniform
    for (int j = 0; j < total; j += 4) {
      float4 a[M];
#pragma unroll
      for (int k = 0; k < M; ++k) a[k] = *reinterpret_cast<const float4*>(s_front + k * cap + j);
      if (a[0].x > wmax) break;                        // wave-uniform: this member and all later ones lie behind
#pragma unroll
      for (int q = 0; q < kPaQ; ++q) {
        bool d0 = true, d1 = true, d2 = true, d3 = true;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          d0 &= a[k].x <= b[q][k]; d1 &= a[k].y <= b[q][k]; d2 &= a[k].z <= b[q][k]; d3 &= a[k].w <= b[q][k];
        }
        cnt[q] -= (d0 ? 1 : 0) + (d1 ? 1 : 0) + (d2 ? 1 : 0) + (d3 ? 1 : 0);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kPaQ; ++q)
    if (got[q] >= 0) rank[tid * kPaQ + q] = got[q];
  for (int t = tid; t < f - cur; t += kPaThreads) sizes[cur + t] = s_sizes[t];
  if (tid == 0) { info[0] = f; info[1] = ranked; }
}

// The whole sort of a small population (n <= 1024) in one launch and without a read-back: one point per thread,
// dominator counts by all pairs from an LDS copy of the points (no ordering needed), then the same peeling loop
// with the LDS copy reused for the front.  Writes the final int64 front indices in the caller's row order (points
// behind the front that reached `target` share the next index) and (last front kept apart, points in front of it,
// its size) into info.
constexpr int kSmThreads = 1024;

template <int M>
__global__ __launch_bounds__(kSmThreads) void nds_small_kernel(const float* __restrict__ x, int64_t ldx, int n, int target,
                                                               int64_t* __restrict__ rank_out, int* __restrict__ info) {
  __shared__ __attribute__((aligned(16))) float s_pts[M][kSmThreads];
  __shared__ int s_wave[2][kSmThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool have = tid < n;
  const int n4 = (n + 3) & ~3;
  float b[M];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    b[k] = have ? x[(int64_t)tid * ldx + k] : __builtin_nanf("");
    if (tid < n4) s_pts[k][tid] = b[k];
  }
  __syncthreads();
  int cnt = 0;
  for (int j = 0; j < n4; j += 4) {
    float4 a[M];
#pragma unroll
    for (int k = 0; k < M; ++k) a[k] = *reinterpret_cast<const float4*>(&s_pts[k][j]);
    bool le0 = true, le1 = true, le2 = true, le3 = true, lt0 = false, lt1 = false, lt2 = false, lt3 = false;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      le0 &= a[k].x <= b[k]; le1 &= a[k].y <= b[k]; le2 &= a[k].z <= b[k]; le3 &= a[k].w <= b[k];
      lt0 |= a[k].x < b[k]; lt1 |= a[k].y < b[k]; lt2 |= a[k].z < b[k]; lt3 |= a[k].w < b[k];
    }
    cnt += ((le0 & lt0) ? 1 : 0) + ((le1 & lt1) ? 1 : 0) + ((le2 & lt2) ? 1 : 0) + ((le3 & lt3) ? 1 : 0);
  }
  if (!have) cnt = -1;
  int got = -1, f = 0, ranked = 0, total = 0, before = 0, last_total = 0;
  while (true) {
    const bool flag = cnt == 0;
    const unsigned long long bal = __ballot(flag);
    int* sw = s_wave[f & 1];
    if (lane == 0) sw[wave] = __popcll(bal);
    __syncthreads();                                   // also: every reader of the points / previous front is done
    int pos = __popcll(bal & ((1ull << lane) - 1ull));
    total = 0;
#pragma unroll
    for (int w = 0; w < kSmThreads / 64; ++w) { const int c = sw[w]; total += c; if (w < wave) pos += c; }
    total = __builtin_amdgcn_readfirstlane(total);     // the same in every lane: keep it (and the loops over it) scalar
    if (total == 0) break;
    if (flag) {
#pragma unroll
      for (int k = 0; k < M; ++k) s_pts[k][pos] = b[k];
      got = f;
      cnt = -1;
    }
    if (tid < ((4 - (total & 3)) & 3)) s_pts[0][total + tid] = __builtin_nanf("");
    before = ranked;
    last_total = total;
    ranked += total;
    ++f;
    if (ranked >= target) break;
    __syncthreads();
    if (__ballot(cnt > 0) == 0ull) continue;           // wave-uniform
    for (int j = 0; j < total; j += 4) {
      float4 a[M];
#pragma unroll
      for (int k = 0; k < M; ++k) a[k] = *reinterpret_cast<const float4*>(&s_pts[k][j]);
      bool d0 = true, d1 = true, d2 = true, d3 = true;
#pragma unroll
      for (int k = 0; k < M; ++k) { d0 &= a[k].x <= b[k]; d1 &= a[k].y <= b[k]; d2 &= a[k].z <= b[k]; d3 &= a[k].w <= b[k]; }
      cnt -= (d0 ? 1 : 0) + (d1 ? 1 : 0) + (d2 ? 1 : 0) + (d3 ? 1 : 0);
    }
  }
  if (have) rank_out[tid] = got >= 0 ? got : f;
  if (tid == 0) {
    info[0] = f - 1;
    info[1] = before;
    info[2] = last_total;
  }
}

template <int M>
int launch_small(const float* x, int64_t ldx, int n, int target, int64_t* rank_out, int* info, hipStream_t s) {
  nds_small_kernel<M><<<1, kSmThreads, 0, s>>>(x, ldx, n, target, rank_out, info);
  return last_launch_status();
}

template <int M>
int launch_peel_all(const float* obj, int64_t ld, int n, const int* count, int* rank, int cur, int ranked, int target,
                    int* sizes, int* info, hipStream_t s) {
  const size_t lds = (size_t)(M + 1) * ((n + 3) & ~3) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)nds_peel_all_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  nds_peel_all_kernel<M><<<1, kPaThreads, lds, s>>>(obj, ld, n, count, rank, cur, ranked, target, sizes, info);
  return last_launch_status();
}

template <int M>
int launch_count(const float* obj, int64_t ld, int n, int* count, int ysplit, hipStream_t s) {
  dim3 grid((n + NT - 1) / NT, ysplit);
  nds_count_kernel<M><<<grid, NT, 0, s>>>(obj, ld, n, count);
  return last_launch_status();
}

template <int M>
int launch_round(const float* obj, int64_t ld, int n, int* count, int* rank, float* front_obj, int* nf, int* blockcnt,
                 int* blockend, int cur, hipStream_t s) {
  const int bx = (n + NT - 1) / NT;
  nds_mark_kernel<<<bx, NT, 0, s>>>(count, rank, blockcnt, n, cur);
  nds_compact_kernel<M><<<bx, NT, 0, s>>>(obj, ld, rank, blockcnt, blockend, front_obj, nf, n, cur);
  int st = last_launch_status();
  if (st) return st;
  int by = 1;
  while (bx * by < 1024 && by < 16) by *= 2;
  nds_peel_kernel<M><<<dim3(bx, by), NT, 0, s>>>(obj, ld, n, front_obj, blockend, count, rank);
  return last_launch_status();
}

}  // namespace

#define EVOX_NDS_DISPATCH(FN, ...)                 \
  switch (m) {                                     \
    case 2: return FN<2>(__VA_ARGS__);             \
    case 3: return FN<3>(__VA_ARGS__);             \
    case 4: return FN<4>(__VA_ARGS__);             \
    case 5: return FN<5>(__VA_ARGS__);             \
    case 6: return FN<6>(__VA_ARGS__);             \
    case 7: return FN<7>(__VA_ARGS__);             \
    case 8: return FN<8>(__VA_ARGS__);             \
    default: return (int)hipErrorInvalidValue;     \
  }

// obj: m x ld objective-major (ld >= n, a multiple of 16, 64-byte aligned base), columns sorted by
// the first objective.
EVOX_API int evox_hip_nds_count(const float* obj, int64_t ld, int n, int m, int* count, int ysplit, void* stream) {
  if (n <= 0 || ysplit < 1 || ld < n || ld % 16) return (int)hipErrorInvalidValue;
  EVOX_NDS_DISPATCH(launch_count, obj, ld, n, count, ysplit, (hipStream_t)stream)
}

// One peeling round: gather front `cur` in sorted order (its size is written to *nf) and subtract its
// dominations from the counts of the points still unranked.  blockcnt / blockend: ceil(n/256) ints each.
EVOX_API int evox_hip_nds_round(const float* obj, int64_t ld, int n, int m, int* count, int* rank, float* front_obj,
                                int* nf, int* blockcnt, int* blockend, int cur, void* stream) {
  if (n <= 0 || ld < n || ld % 16) return (int)hipErrorInvalidValue;
  EVOX_NDS_DISPATCH(launch_round, obj, ld, n, count, rank, front_obj, nf, blockcnt, blockend, cur, (hipStream_t)stream)
}

// `rounds` consecutive peeling rounds from one host call (fronts cur .. cur+rounds-1; their sizes go to
// nf[0 .. rounds)).  A round whose front is empty changes nothing, so issuing rounds past the last front is safe.
EVOX_API int evox_hip_nds_rounds(const float* obj, int64_t ld, int n, int m, int* count, int* rank, float* front_obj,
                                 int* nf, int* blockcnt, int* blockend, int cur, int rounds, void* stream) {
  if (rounds < 0) return (int)hipErrorInvalidValue;
  for (int r = 0; r < rounds; ++r) {
    const int st = evox_hip_nds_round(obj, ld, n, m, count, rank, front_obj, nf + r, blockcnt, blockend, cur + r, stream);
    if (st) return st;
  }
  return 0;
}

// x: n x m row-major (row stride ldx), n <= 1024.  See nds_small_kernel.
EVOX_API int evox_hip_nds_small(const float* x, int64_t ldx, int n, int m, int target, int64_t* rank_out, int* info,
                                void* stream) {
  if (n <= 0 || n > kSmThreads || ldx < m) return (int)hipErrorInvalidValue;
  EVOX_NDS_DISPATCH(launch_small, x, ldx, n, target, rank_out, info, (hipStream_t)stream)
}

// The rest of the sort from one workgroup (n <= 4096): fronts cur, cur+1, ... until `target` points are ranked or no
// unranked point is free of dominators.  count / rank: the state after front cur-1 was removed (cur = 0: straight
// after evox_hip_nds_count); ranked: points in fronts < cur.  sizes[f] receives the size of every front collected
// here, info[0] the number of fronts collected in all, info[1] the number of ranked points.
EVOX_API int evox_hip_nds_peel_all(const float* obj, int64_t ld, int n, int m, const int* count, int* rank, int cur,
                                   int ranked, int target, int* sizes, int* info, void* stream) {
  if (n <= 0 || n > kPaMaxN || ld < n || cur < 0) return (int)hipErrorInvalidValue;
  EVOX_NDS_DISPATCH(launch_peel_all, obj, ld, n, count, rank, cur, ranked, target, sizes, info, (hipStream_t)stream)
}

// Crowding distance of one front from per-objective orders.  The caller ranks every objective with
// the counting-rank kernel (order and inverse permutation in one launch each); a member's neighbours
// in objective k are then order_k[rank_k[i] ± 1]:
//     dist_i = Σ_k (v_k[next] − v_k[prev]) / (max_k − min_k),   +inf at either end of any objective
// (the boundary points of NSGA-II).  vals / order / rank: m × count, objective-major.
namespace {

__global__ __launch_bounds__(256) void crowding_kernel(const float* __restrict__ vals,
                                                       const int64_t* __restrict__ order,
                                                       const int64_t* __restrict__ rank, int count, int m,
                                                       float* __restrict__ dist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float total = 0.f;
  for (int k = 0; k < m; ++k) {
    const float* v = vals + (size_t)k * count;
    const int64_t* o = order + (size_t)k * count;
    const int64_t r = rank[(size_t)k * count + i];
    if (r == 0 || r == count - 1) {
      total = INFINITY;
      break;
    }
    const float span = v[o[count - 1]] - v[o[0]];
    total += (v[o[r + 1]] - v[o[r - 1]]) / span;
  }
  dist[i] = total;
}

}  // namespace

EVOX_API int evox_hip_crowding(const float* vals, const int64_t* order, const int64_t* rank, int count, int m,
                               float* dist, void* stream) {
  if (count <= 0 || m <= 0) return 0;
  crowding_kernel<<<(count + 255) / 256, 256, 0, (hipStream_t)stream>>>(vals, order, rank, count, m, dist);
  return last_launch_status();
}
