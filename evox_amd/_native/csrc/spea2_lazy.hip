// SPEA2 archive truncation in rounds that use the whole chip, with the survivors of the one-at-a-time loop.
//
// The loop removes, one at a time, the point whose nearest live neighbour is closest (smallest index among equals).
// Removing a point can only push the nearest-neighbour distance of the others up, so — as for IBEA's removal
// (ibea_lazy.hip) — the K points with the smallest distance can be played through on their own for as long as their
// smallest stays below the (K+1)-th of the round's start.  One round:
//   select : one workgroup finds the K points S and the bound T by a search on the bits of the distances,
//   lists  : one workgroup per member of S collects its 8 nearest live neighbours (all of the chip at once),
//   play   : one wave removes members of S in order; a member whose neighbour went takes the next live entry of its
//            list, and the round ends early when a list runs out,
//   rescan : every live point whose neighbour is gone (inside or outside S) gets its nearest live neighbour again
//            (three small launches: list them, stream all points past them in LDS tiles, write the results).
// No n × n distance matrix exists; distances are recomputed from the objective vectors with the expression of
// distance.hip::cdist_small_kernel, so the values — and the survivors — equal those of the matrix form.
#include "hip_common.h"
#include "lazy_select.h"

using namespace evox;
using namespace evox::lazy;

namespace {

constexpr int kList = 8;            // neighbours listed per member of S
constexpr int kKS = 512;            // points a round plays through: the play loop has no per-member arithmetic, so a
constexpr int kPerLaneS = kKS / 64; // larger S than IBEA's costs little per removal and halves the number of rounds
constexpr int kMaxPoints = 65536;   // the play launch keeps one bit per point in LDS

enum { kRemoved = 0, kRounds = 1, kRoundCount = 2, kSCount = 3, kTauBits = 4, kTIdx = 5, kTau = 6, kTauStep = 7,
       kTieRounds = 8, kShortRounds = 9, kFlagged = 10, kCtrlWords = 16 };

template <int M>
__device__ __forceinline__ float dist(const float (&a)[M], const float (&b)[M]) {
  return sqrtf(squared_distance<M>(a, b));                // the rounding sequence of distance.hip::cdist_small_kernel
}

// Rescan: the nearest live neighbour (distance, index; smallest index among equals) of every live point whose recorded
// neighbour is gone — or of every live point (all != 0) — in three launches.
// (1) the points concerned are listed (one atomic per wave);
template <int DUMMY>
__global__ __launch_bounds__(256) void spea2_flag_kernel(int n, const uint8_t* __restrict__ alive, const int* __restrict__ nn,
                                                         int* __restrict__ ctrl, int all, int* __restrict__ flagged,
                                                         unsigned long long* __restrict__ best) {
  if (!all && ctrl[kRoundCount] <= 0) return;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool flag = false;
  if (i < n && alive[i]) {
    const int a = nn[i];
    flag = all || a < 0 || !alive[a];
  }
  const unsigned long long mask = __ballot(flag);
  if (mask == 0) return;
  const int leader = (int)__builtin_ctzll(mask);
  int base = 0;
  if (lane == leader) base = atomicAdd(&ctrl[kFlagged], (int)__popcll(mask));
  base = __shfl(base, leader, 64);
  if (flag) {
    const int pos = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
    flagged[pos] = i;
    best[pos] = ~0ull;
  }
}

// (2) 64 listed points per workgroup pass against one slice of the columns, streamed through LDS in tiles of 256 (wave s
// takes columns 64 s … 64 s + 63 of a tile, so its LDS reads are broadcasts); the result is merged with a 64-bit minimum
// on (distance bits, index) — distances are non-negative, so their bits order like integers;
template <int M>
__global__ __launch_bounds__(256) void spea2_scan_kernel(const float* __restrict__ obj, int n, const uint8_t* __restrict__ alive,
                                                         const int* __restrict__ ctrl, const int* __restrict__ flagged,
                                                         unsigned long long* __restrict__ best) {
  __shared__ float tile[M][256];
  __shared__ uint8_t tile_alive[256];
  const int count = ctrl[kFlagged];
  if (count <= 0) return;
  const int per = (n + (int)gridDim.y - 1) / (int)gridDim.y;
  const int c0 = blockIdx.y * per, c1 = min(n, c0 + per);
  const int r = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (int g = blockIdx.x; g * 64 < count; g += gridDim.x) {
    const int fr = g * 64 + r;
    const int p = fr < count ? flagged[fr] : -1;
    float mine[M];
#pragma unroll
    for (int k = 0; k < M; ++k) mine[k] = p >= 0 ? obj[(int64_t)p * M + k] : 0.f;
    float v = INFINITY;
    int a = 0;
    for (int j0 = c0; j0 < c1; j0 += 256) {
      __syncthreads();
      const int jj = j0 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < M; ++k) tile[k][threadIdx.x] = jj < c1 ? obj[(int64_t)jj * M + k] : 0.f;
      tile_alive[threadIdx.x] = jj < c1 ? alive[jj] : 0;
      __syncthreads();
#pragma unroll 4
      for (int q = 0; q < 64; ++q) {
        const int c = sl * 64 + q;
        float oj[M];
#pragma unroll
        for (int k = 0; k < M; ++k) oj[k] = tile[k][c];
        const float d = dist<M>(mine, oj);
        const bool ok = tile_alive[c] != 0 && j0 + c != p;
        if (ok && d < v) { v = d; a = j0 + c; }
      }
    }
    if (p >= 0 && v < INFINITY)
      atomicMin(&best[fr], ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)a);
  }
}

// (3) and written to the per-point arrays.
template <int DUMMY>
__global__ __launch_bounds__(256) void spea2_finish_kernel(const int* __restrict__ ctrl, const int* __restrict__ flagged,
                                                           const unsigned long long* __restrict__ best,
                                                           float* __restrict__ rowmin, int* __restrict__ nn) {
  const int count = ctrl[kFlagged];
  for (int fr = blockIdx.x * 256 + threadIdx.x; fr < count; fr += gridDim.x * 256) {
    const unsigned long long b = best[fr];
    const int p = flagged[fr];
    if (b == ~0ull) { rowmin[p] = INFINITY; nn[p] = -1; }
    else { rowmin[p] = __uint_as_float((unsigned)(b >> 32)); nn[p] = (int)(unsigned)(b & 0xFFFFFFFFull); }
  }
}

// S and T of a round.  Thread t owns the points t·E … t·E + E − 1.  s_list: out (kKS) indices of S.
template <int E>
__global__ __launch_bounds__(kSel) void spea2_lazy_select_kernel(const float* __restrict__ rowmin, int n, int n_remove,
                                                                 int* __restrict__ ctrl, int* __restrict__ s_list) {
  __shared__ int s_cnt[2][kSel / 64];
  __shared__ int s_scan[kSel / 64];
  __shared__ int s_n, s_tidx;
  const int tid = threadIdx.x, lane = tid & 63;
  if (ctrl[kRemoved] >= n_remove) {                       // uniform: complete, the other launches of the round do nothing
    if (tid == 0) { ctrl[kSCount] = 0; ctrl[kRoundCount] = 0; ctrl[kFlagged] = 0; }
    return;
  }
  const unsigned tau_prev = (unsigned)ctrl[kTau], step_prev = (unsigned)ctrl[kTauStep];
  unsigned key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid * E + e;
    key[e] = j < n ? member_key(rowmin[j]) : kDeadKey;   // removed points carry +inf
  }
  if (tid == 0) { s_n = 0; s_tidx = -1; }
  int flip = 0, bisections = 0;
  const int live = block_count_below<E>(key, kDeadKey, s_cnt[flip]); flip ^= 1;
  unsigned tau = kDeadKey;
  int below = live;
  bool tie = false;
  if (live > kKS) tau = find_tau<E, kKS>(key, tau_prev, step_prev, s_cnt, flip, bisections, below, tie);
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (key[e] < tau) s_list[atomicAdd(&s_n, 1)] = tid * E + e;
  if (tie) {
    // the points with key == τ join in index order until S is full; the first one left out is T
    int mine = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) mine += key[e] == tau ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    int pos = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) pos += s_scan[w];
    const int room = kKS - below;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (key[e] == tau) {
        if (pos < room) s_list[below + pos] = tid * E + e;
        else if (pos == room) s_tidx = tid * E + e;
        ++pos;
      }
  }
  __syncthreads();
  if (tid == 0) {
    ctrl[kSCount] = tie ? kKS : s_n;
    ctrl[kFlagged] = 0;
    ctrl[kTauBits] = (int)tau;
    ctrl[kTIdx] = s_tidx;
    if (tau != kDeadKey) {
      ctrl[kTauStep] = (int)(tau_prev != 0 && tau > tau_prev ? tau - tau_prev : 0);
      ctrl[kTau] = (int)tau;
    }
    if (tie) ctrl[kTieRounds] += 1;
  }
}

// The kList nearest live neighbours of member blockIdx.x of S: ascending distance, the larger index first among equal
// distances (the play launch removes small indices first among equals, so these entries live longest).
template <int M>
__global__ __launch_bounds__(256) void spea2_lazy_list_kernel(const float* __restrict__ obj, int n, const uint8_t* __restrict__ alive,
                                                              const int* __restrict__ ctrl, const int* __restrict__ s_list,
                                                              float* __restrict__ list_d, int* __restrict__ list_i) {
  __shared__ float s_d[4];
  __shared__ int s_i[4];
  const int s = blockIdx.x;
  if (s >= ctrl[kSCount]) return;
  const int p = s_list[s];
  float op[M];
#pragma unroll
  for (int k = 0; k < M; ++k) op[k] = obj[(int64_t)p * M + k];
  float ld[kList];
  int li[kList];
#pragma unroll
  for (int t = 0; t < kList; ++t) { ld[t] = INFINITY; li[t] = -1; }
  for (int j0 = threadIdx.x; j0 < n; j0 += 1024) {        // four points at a time: their loads are in flight together
    float dd[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 256 * u;
      const int jc = min(j, n - 1);
      float oj[M];
#pragma unroll
      for (int k = 0; k < M; ++k) oj[k] = obj[(int64_t)jc * M + k];
      const bool ok = j < n && alive[jc] != 0 && j != p;
      dd[u] = ok ? dist<M>(op, oj) : INFINITY;
      cc[u] = ok ? j : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float d = dd[u];
      int c = cc[u];
      if (d < ld[kList - 1] || (d == ld[kList - 1] && c > li[kList - 1])) {
#pragma unroll
        for (int t = 0; t < kList; ++t) {                // insertion by a chain of compare-exchanges
          const bool better = d < ld[t] || (d == ld[t] && c > li[t]);
          const float td = ld[t];
          const int ti = li[t];
          ld[t] = better ? d : td; li[t] = better ? c : ti;
          d = better ? td : d; c = better ? ti : c;
        }
      }
    }
  }
  // merge: kList times the best head of all threads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = 0; t < kList; ++t) {
    float v = ld[0];
    int a = li[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oa = __shfl_xor(a, o, 64);
      if (ov < v || (ov == v && oa > a)) { v = ov; a = oa; }
    }
    __syncthreads();
    if (lane == 0) { s_d[wave] = v; s_i[wave] = a; }
    __syncthreads();
    v = s_d[0]; a = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (s_d[w] < v || (s_d[w] == v && s_i[w] > a)) { v = s_d[w]; a = s_i[w]; }
    if (threadIdx.x == 0) { list_d[s * kList + t] = v; list_i[s * kList + t] = a; }
    if (a >= 0 && li[0] == a) {                           // the winner's thread moves on to its next entry
#pragma unroll
      for (int u = 0; u + 1 < kList; ++u) { ld[u] = ld[u + 1]; li[u] = li[u + 1]; }
      ld[kList - 1] = INFINITY; li[kList - 1] = -1;
    }
  }
}

// The removals among S, by one wave.  Distances are non-negative, so their bits order like integers.
__global__ __launch_bounds__(256) void spea2_lazy_play_kernel(int n, int n_remove, uint8_t* __restrict__ alive,
                                                             float* __restrict__ rowmin, int* __restrict__ nn,
                                                             int* __restrict__ ctrl, const int* __restrict__ s_list,
                                                             const float* __restrict__ list_d, const int* __restrict__ list_i) {
  __shared__ int2 l_e[kKS * kList];                        // neighbour lists: (index, bits of the distance)
  __shared__ int s_idx[kKS];
  __shared__ int s_order[kKS];
  __shared__ unsigned s_gone[kMaxPoints / 32];            // one bit per point: removed in this round
  const int lane = threadIdx.x & 63;
  const int cnt = ctrl[kSCount];
  if (cnt <= 0) return;
  const int done = ctrl[kRemoved];
  const int budget = n_remove - done;
  // the whole workgroup sets the stage, its first wave plays
  for (int w = threadIdx.x; w < (n + 31) / 32; w += 256) s_gone[w] = 0u;
  for (int e = threadIdx.x; e < kKS * kList; e += 256) {
    const bool have = e / kList < cnt;
    l_e[e] = have ? make_int2(list_i[e], __float_as_int(list_d[e])) : make_int2(-1, 0x7F800000);
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  constexpr int kInfKey = 0x7F800000;
  float f[kPerLaneS];
  int id[kPerLaneS], nb[kPerLaneS], ptr[kPerLaneS];
#pragma unroll
  for (int q = 0; q < kPerLaneS; ++q) {
    const int slot = lane + 64 * q;
    id[q] = slot < cnt ? s_list[slot] : 0x7fffffff;
    f[q] = slot < cnt ? rowmin[id[q]] : INFINITY;
    nb[q] = slot < cnt ? nn[id[q]] : -1;
    ptr[q] = 0;
    s_idx[slot] = id[q];
  }
  const int tau_key = (int)((unsigned)ctrl[kTauBits] ^ 0x80000000u);
  const int t_idx = ctrl[kTIdx];
  int removed = 0;
  bool short_list = false;
  while (removed < budget) {
    int key[kPerLaneS];
#pragma unroll
    for (int q = 0; q < kPerLaneS; ++q) key[q] = __float_as_int(f[q]);
    int lk = key[0];
#pragma unroll
    for (int q = 1; q < kPerLaneS; ++q) lk = min(lk, key[q]);
    const int wk = wave_min_dpp(lk);
    if (wk >= kInfKey) break;                               // S is used up
    // equal distances are the rule (both points of a closest pair): the smaller index goes, and it names x right away
    int cand = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < kPerLaneS; ++q) cand = key[q] == wk ? min(cand, id[q]) : cand;
    const int x = wave_min_dpp(cand);
    unsigned long long hit[kPerLaneS];
#pragma unroll
    for (int q = 0; q < kPerLaneS; ++q) hit[q] = __ballot(key[q] == wk && id[q] == x);
    int q_sel = 0;
    unsigned long long h = hit[0];
#pragma unroll
    for (int q = 1; q < kPerLaneS; ++q)
      if (hit[q] != 0) { q_sel = q; h = hit[q]; }           // scalar
    const int src = (int)__builtin_ctzll(h);
    const int x_slot = q_sel * 64 + src;
    if (wk > tau_key || (wk == tau_key && x >= t_idx)) break;   // no longer below T: a point outside S may be next
#pragma unroll
    for (int q = 0; q < kPerLaneS; ++q) f[q] = (q == q_sel && lane == src) ? INFINITY : f[q];
    if (lane == 0) atomicOr(&s_gone[x >> 5], 1u << (x & 31));   // no read on the critical path
    s_order[removed] = x_slot;
    ++removed;
    // members of S that lose their neighbour take the next live entry of their list
    bool out = false;
#pragma unroll
    for (int q = 0; q < kPerLaneS; ++q) {
      if (nb[q] == x && f[q] < INFINITY) {
        const int slot = lane + 64 * q;
        int p = ptr[q], c = -1, dbits = 0;
        while (p < kList) {
          // two entries and their bitmap words per round trip; x itself is gone without asking the bitmap
          const int2 e0 = l_e[slot * kList + p];
          const int2 e1 = l_e[slot * kList + min(p + 1, kList - 1)];
          const int c0 = e0.x, c1 = p + 1 < kList ? e1.x : -1;
          const unsigned w0 = c0 >= 0 ? s_gone[c0 >> 5] : 0u, w1 = c1 >= 0 ? s_gone[c1 >> 5] : 0u;
          const bool gone0 = c0 >= 0 && (c0 == x || ((w0 >> (c0 & 31)) & 1u));
          const bool gone1 = c1 >= 0 && (c1 == x || ((w1 >> (c1 & 31)) & 1u));
          if (!gone0) { c = c0; dbits = e0.y; break; }      // a live entry, or the end of a short list (c0 < 0)
          ++p;
          if (!gone1) { c = c1; dbits = e1.y; break; }
          ++p;
        }
        if (p < kList && c >= 0) { f[q] = __int_as_float(dbits); nb[q] = c; ptr[q] = p; }
        else out = true;                                    // the list ran out: this member waits for the rescan
      }
    }
    if (__ballot(out) != 0) { short_list = true; break; }
  }
  for (int t = lane; t < removed; t += 64) {
    const int x = s_idx[s_order[t]];
    alive[x] = 0; rowmin[x] = INFINITY; nn[x] = -1;
  }
#pragma unroll
  for (int q = 0; q < kPerLaneS; ++q)
    if (f[q] < INFINITY) { rowmin[id[q]] = f[q]; nn[id[q]] = nb[q]; }
  if (lane == 0) {
    ctrl[kRemoved] = done + removed;
    ctrl[kRounds] += 1;
    ctrl[kRoundCount] = removed;
    if (short_list) ctrl[kShortRounds] += 1;
  }
}

__global__ __launch_bounds__(64) void spea2_lazy_reset_kernel(int* ctrl) {
  if (threadIdx.x < kCtrlWords) ctrl[threadIdx.x] = 0;
}

template <int E>
void launch_select(const float* rowmin, int n, int n_remove, int* ctrl, int* s_list, hipStream_t st) {
  spea2_lazy_select_kernel<E><<<1, kSel, 0, st>>>(rowmin, n, n_remove, ctrl, s_list);
}

void launch_select_any(const float* rowmin, int n, int n_remove, int* ctrl, int* s_list, hipStream_t st) {
  if (n <= 4 * kSel) launch_select<4>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 8 * kSel) launch_select<8>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 12 * kSel) launch_select<12>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 16 * kSel) launch_select<16>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 20 * kSel) launch_select<20>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 24 * kSel) launch_select<24>(rowmin, n, n_remove, ctrl, s_list, st);
  else if (n <= 32 * kSel) launch_select<32>(rowmin, n, n_remove, ctrl, s_list, st);
  else launch_select<64>(rowmin, n, n_remove, ctrl, s_list, st);
}

template <int M>
int launch_rounds(const float* obj, int n, uint8_t* alive, float* rowmin, int* nn, int n_remove, int* ctrl, int* s_list,
                  float* list_d, int* list_i, int* flagged, unsigned long long* best, int reset, int rounds,
                  hipStream_t st) {
  const int blocks = (n + 255) / 256;
  auto rescan = [&](int all) {
    spea2_flag_kernel<0><<<blocks, 256, 0, st>>>(n, alive, nn, ctrl, all, flagged, best);
    spea2_scan_kernel<M><<<dim3(8, 32), 256, 0, st>>>(obj, n, alive, ctrl, flagged, best);
    spea2_finish_kernel<0><<<8, 256, 0, st>>>(ctrl, flagged, best, rowmin, nn);
  };
  if (reset) {
    spea2_lazy_reset_kernel<<<1, 64, 0, st>>>(ctrl);
    rescan(1);
  }
  for (int r = 0; r < rounds; ++r) {
    launch_select_any(rowmin, n, n_remove, ctrl, s_list, st);
    spea2_lazy_list_kernel<M><<<kKS, 256, 0, st>>>(obj, n, alive, ctrl, s_list, list_d, list_i);
    spea2_lazy_play_kernel<<<1, 256, 0, st>>>(n, n_remove, alive, rowmin, nn, ctrl, s_list, list_d, list_i);
    rescan(0);
  }
  return last_launch_status();
}

}  // namespace

EVOX_API int64_t evox_hip_spea2_lazy_max(int m) { return m >= 1 && m <= 8 ? kMaxPoints : 0; }
EVOX_API int64_t evox_hip_spea2_lazy_ctrl_words() { return kCtrlWords; }
// int32 words of the scratch for n points: best (n 64-bit words, first: alignment) + flagged (n) + S (kKS) + neighbour
// indices (kKS · kList) + neighbour distances (kKS · kList floats)
EVOX_API int64_t evox_hip_spea2_lazy_scratch_words(int n) { return 3 * (int64_t)n + kKS + 2 * kKS * kList; }

// obj: (n, m) fp32 points, all live on entry when reset != 0 (alive is then expected to hold ones; rowmin and nn are
// computed here).  Issues `rounds` rounds; rounds after the n_remove-th removal do nothing.  alive: in/out bytes.
// ctrl[0] = points removed so far, ctrl[1] = rounds that did work, ctrl[8] = rounds with the tie order, ctrl[9] = rounds
// cut short by an exhausted neighbour list.
EVOX_API int evox_hip_spea2_lazy_rounds(const float* obj, int n, int m, uint8_t* alive, float* rowmin, int* nn, int n_remove,
                                        int* ctrl, int* scratch, int reset, int rounds, void* stream) {
  if (n <= 0 || rounds <= 0) return 0;
  if (n > kMaxPoints) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(scratch);
  int* flagged = scratch + 2 * (int64_t)n;
  int* s_list = flagged + n;
  int* list_i = s_list + kKS;
  float* list_d = reinterpret_cast<float*>(list_i + kKS * kList);
  switch (m) {
    case 1: return launch_rounds<1>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 2: return launch_rounds<2>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 3: return launch_rounds<3>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 4: return launch_rounds<4>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 5: return launch_rounds<5>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 6: return launch_rounds<6>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 7: return launch_rounds<7>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    case 8: return launch_rounds<8>(obj, n, alive, rowmin, nn, n_remove, ctrl, s_list, list_d, list_i, flagged, best, reset, rounds, st);
    default: return (int)hipErrorInvalidValue;
  }
}
