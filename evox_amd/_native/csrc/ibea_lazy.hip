// IBEA's removal loop in rounds that use the whole chip, with the survivors of the one-at-a-time loop.
//
// Removing a member adds exp(−I(x, j) / (C_x κ)) ≥ 0 to every other member's fitness, so fitness values only grow
// while the loop runs.  Hence: take the K members with the smallest fitness (the set S) and the (K+1)-th smallest
// (value, index) pair T.  Whatever is removed, every member outside S stays ≥ T.  As long as the smallest member of
// S — with S's fitness values kept current — is below T, it is the global arg-min, and the one-at-a-time loop would
// remove exactly it.  So one round is
//   select : one workgroup finds S and T by bisection on the order-preserving bits of the fitness,
//   simulate: one wave plays the removals among the K members of S (four per lane, in registers; an arg-min is a
//             few DPP steps, no barrier, no LDS round trip) until the smallest is no longer below T,
//   apply  : a chip-wide launch adds the increments of the round's removed members, in removal order, to everyone.
// Additions happen in the order of the one-at-a-time loop with the same expression, so the fitness values, and with
// them the survivors, are bit-identical to `ibea_remove_reg_kernel`.  A round removes ≈ 0.6 K members on converging
// populations; the sequential part shrinks from n members per removal on one compute unit to K on one wave.
#include "hip_common.h"
#include "lazy_select.h"

using namespace evox;
using namespace evox::lazy;

namespace {

constexpr int kClones = 32;         // clones of a flooding point that a round keeps in S
constexpr int kCandPer = 8;          // candidates per thread of the select workgroup
constexpr int kCandCap = kCandPer * kSel;
constexpr int kNoSentinel = 1;      // flag: tie rounds never use the sentinel (for tests of the plain tie order)
constexpr int kNoCandidates = 2;    // flag: every round scans the whole population for S

// control words (int32) shared by the launches of one removal
enum { kRemoved = 0, kRounds = 1, kRoundCount = 2, kTieRounds = 3, kSelectTicks = 4, kSimTicks = 5, kBisections = 6, kTau = 7, kTauStep = 8, kSentinelRounds = 9, kGuess = 10, kGuessMult = 11, kCandCount = 12, kCandRounds = 13,
       kCtrlWords = 16 };

template <int M>
__device__ __forceinline__ float eps_ind(const float (&a)[M], const float (&b)[M]) {
  float v = a[0] - b[0];
#pragma unroll
  for (int k = 1; k < M; ++k) v = fmaxf(v, a[k] - b[k]);
  return v;
}

// The removals among the members of S, by one wave; returns how many.  The arg-min runs on integer keys with the order
// of the floats (three operations per member): integer minima fuse with their DPP moves and fold to a scalar, the
// ballots of "my key is the minimum" name lane and register slot of the member as scalars, and only equal keys (rare)
// need the index comparison.  +inf (removed, padding) is the largest key below the NaNs.  The loop is bound by the
// instruction issue of its one wave, so it only logs the slot of each removed member.
//
// SENT: the round started with more than K/2 members of exactly equal fitness that are exact copies of one point (a
// population flooded with clones).  The copies outside S gain exactly what the first of them, the sentinel T, gains, so
// T's fitness is played along and bounds them; everyone else outside S is bounded by the smallest key above τ at the
// start of the round (`next_key`).  Without this the round would end after one removal: removing a copy lifts all
// others by exp(0) = 1, above the stale τ.
template <int M, bool SENT>
__device__ __forceinline__ int simulate(const float* s_f, const int* s_idx, const float (*s_obj)[kK], const float* s_s,
                                        int* s_order, int lane, int budget, int tau_key, int t_idx, int next_key,
                                        const float* __restrict__ obj, const float* __restrict__ fit) {
  constexpr int kInfKey = 0x7F800000;
  float f[kPerLane], o[kPerLane][M];
  int id[kPerLane];
#pragma unroll
  for (int q = 0; q < kPerLane; ++q) {
    const int slot = lane + 64 * q;
    f[q] = s_f[slot]; id[q] = s_idx[slot];
#pragma unroll
    for (int k = 0; k < M; ++k) o[q][k] = s_obj[k][slot];
  }
  float f_t = 0.f, o_t[M];
#pragma unroll
  for (int k = 0; k < M; ++k) o_t[k] = 0.f;
  if (SENT) {
    f_t = fit[t_idx];
#pragma unroll
    for (int k = 0; k < M; ++k) o_t[k] = obj[(int64_t)t_idx * M + k];
  }
  int removed = 0;
  while (removed < budget) {
    int key[kPerLane];
#pragma unroll
    for (int q = 0; q < kPerLane; ++q) {
      const int b = __float_as_int(f[q]);
      key[q] = b ^ ((b >> 31) & 0x7fffffff);
    }
    int lk = key[0];
#pragma unroll
    for (int q = 1; q < kPerLane; ++q) lk = min(lk, key[q]);
    const int wk = wave_min_dpp(lk);
    if (wk >= kInfKey) break;                               // S is used up
    unsigned long long hit[kPerLane];
    int hits = 0;
#pragma unroll
    for (int q = 0; q < kPerLane; ++q) { hit[q] = __ballot(key[q] == wk); hits += __popcll(hit[q]); }
    if (hits == 1) {
#pragma unroll
      for (int q = 0; q < kPerLane; ++q) f[q] = key[q] == wk ? INFINITY : f[q];
    } else {                                                // equal values: the smallest index goes
      int cand = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < kPerLane; ++q) cand = key[q] == wk ? min(cand, id[q]) : cand;
      const int wi = wave_min_dpp(cand);
#pragma unroll
      for (int q = 0; q < kPerLane; ++q) {
        const bool me = key[q] == wk && id[q] == wi;
        hit[q] = __ballot(me);
        f[q] = me ? INFINITY : f[q];
      }
    }
    int q_sel = 0;
    unsigned long long h = hit[0];
#pragma unroll
    for (int q = 1; q < kPerLane; ++q)
      if (hit[q] != 0) { q_sel = q; h = hit[q]; }           // scalar
    const int x_slot = q_sel * 64 + (int)__builtin_ctzll(h);
    // no longer below T: a member outside S may be next (x's register copy is already struck out, but the registers
    // are not used after the loop)
    if (SENT) {
      const int bt = __float_as_int(f_t);
      const int key_t = bt ^ ((bt >> 31) & 0x7fffffff);
      if (wk >= next_key || wk > key_t || (wk == key_t && s_idx[x_slot] >= t_idx)) break;
    } else {
      if (wk > tau_key || (wk == tau_key && s_idx[x_slot] >= t_idx)) break;
    }
    s_order[removed] = x_slot;
    float ox[M];
#pragma unroll
    for (int k = 0; k < M; ++k) ox[k] = s_obj[k][x_slot];
    const float s = s_s[x_slot];
#pragma unroll
    for (int q = 0; q < kPerLane; ++q) f[q] += __expf(eps_ind<M>(ox, o[q]) * s);      // +inf stays +inf
    if (SENT) f_t += __expf(eps_ind<M>(ox, o_t) * s);
    ++removed;
  }
  return removed;
}

// One round's select and simulate.  Thread t owns the members t·E … t·E + E − 1 (index order = thread order).
// rem: (kK, M + 1) floats — objectives and −1 / (C κ) of the members removed in this round, in removal order.
template <int M, int E>
__global__ __launch_bounds__(kSel) void ibea_lazy_select_kernel(const float* __restrict__ obj, int n,
                                                                const float* __restrict__ C, float kappa,
                                                                float* __restrict__ fit, int n_remove,
                                                                int* __restrict__ ctrl, float* __restrict__ rem,
                                                                const int* __restrict__ cand, int flags) {
  __shared__ int s_cnt[2][kSel / 64];
  __shared__ int s_scan[kSel / 64];
  __shared__ int s_n;                 // members of S so far
  __shared__ int s_tidx;              // index part of T
  __shared__ int s_tclone;            // tie round: the clone after the first kClones
  __shared__ int s_idx[kK];
  __shared__ float s_f[kK];
  __shared__ float s_obj[M][kK];
  __shared__ float s_s[kK];
  __shared__ int s_order[kK];         // slots of the removed members, in removal order
  __shared__ int s_dup;               // tie round: every member with key τ is an exact copy of T's member
  __shared__ unsigned s_next;         // tie round: smallest key above τ

  const int done = ctrl[kRemoved];
  if (done >= n_remove) {             // uniform: the removal is complete, later rounds of the chunk do nothing
    if (threadIdx.x == 0) ctrl[kRoundCount] = 0;
    return;
  }
  const int budget = n_remove - done;
  const long long t_start = wall_clock64();
  int bisections = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned tau_prev = (unsigned)ctrl[kTau], step_prev = (unsigned)ctrl[kTauStep];
  const int cc = ctrl[kCandCount];
  int mult = ctrl[kGuessMult];
  if (mult == 0) mult = 4;
  if (ctrl[kGuess] != 0) {            // the last apply launch collected candidates: too few, or too many?
    if (cc <= kK) mult = min(mult * 2, 256);
    else if (cc > kCandCap / 2) mult = max(mult / 2, 1);
  }
  if (tid == 0) { s_n = 0; s_tidx = -1; s_dup = 1; s_next = kDeadKey; }
  if (tid < kK) { s_idx[tid] = 0x7fffffff; s_f[tid] = INFINITY; }
  int flip = 0;
  unsigned tau = kDeadKey, tau_warm = kDeadKey;
  int below = 0;
  bool tie = false, clones = false;

  // ---- select among the candidates the last apply launch listed: every member with a key below its guess G.  More
  // than K of them: the K + 1 smallest of the population are among them.  (A tie needs the index order: full scan.)
  bool from_cand = cc > kK && cc <= kCandCap && (flags & kNoCandidates) == 0;
  if (from_cand) {
    unsigned ck[kCandPer];
    int ci[kCandPer];
#pragma unroll
    for (int e = 0; e < kCandPer; ++e) {
      const int p = tid + e * kSel;
      ck[e] = kDeadKey; ci[e] = 0;
      if (p < cc) { ck[e] = (unsigned)cand[2 * p]; ci[e] = cand[2 * p + 1]; }
    }
    tau = find_tau<kCandPer>(ck, tau_prev, step_prev, s_cnt, flip, bisections, below, tie);
    tau_warm = tau;
    if (tie) from_cand = false;
    else {
#pragma unroll
      for (int e = 0; e < kCandPer; ++e)
        if (ck[e] < tau) s_idx[atomicAdd(&s_n, 1)] = ci[e];
    }
  }
  if (!from_cand) {
  unsigned key[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid * E + e;
    key[e] = j < n ? member_key(fit[j]) : kDeadKey;
  }
  const int alive = block_count_below<E>(key, kDeadKey, s_cnt[flip]); flip ^= 1;
  tau = kDeadKey;                     // alive ≤ K: S is everyone
  below = alive;
  tie = false;
  if (alive > kK) tau = find_tau<E>(key, tau_prev, step_prev, s_cnt, flip, bisections, below, tie);
  tau_warm = tau;
  if (tie) {
    // position of every member with key == τ among them, in index order
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
    unsigned above = kDeadKey;        // the smallest key above τ
#pragma unroll
    for (int e = 0; e < E; ++e) above = key[e] > tau ? min(above, key[e]) : above;
    above = wave_min_dpp(above);
    if (lane == 0) atomicMin(&s_next, above);
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < (tid >> 6); ++w) before += s_scan[w];
    const int room = kK - below;      // > K/2
    // Clones of a flooding point that a round would keep in S: removing one lifts the others by exactly 1, so about
    // (next fitness above − their fitness) of them go before anyone else is due; at least kClones, and one is left over
    // as the sentinel.
    int n_cl = kClones;
    {
      const float gap = key_value(s_next) - key_value(tau);
      if (!(gap < (float)(room - 2))) n_cl = room - 1;
      else n_cl = max(kClones, min(room - 1, (int)gap + 2));
    }
    {
      int pos = before;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (key[e] == tau) {
          if (pos == room) s_tidx = tid * E + e;
          if (pos == n_cl) s_tclone = tid * E + e;
          ++pos;
        }
    }
    __syncthreads();
    // are the members of the tie group copies of one point?
    const int tj = s_tclone;
    float o_t[M];
#pragma unroll
    for (int k = 0; k < M; ++k) o_t[k] = obj[(int64_t)tj * M + k];
    const float c_t = C[tj];
    bool same = true;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (key[e] == tau) {
        const int j = tid * E + e;
        same = same && C[j] == c_t;
#pragma unroll
        for (int k = 0; k < M; ++k) same = same && obj[(int64_t)j * M + k] == o_t[k];
      }
    if (!same) s_dup = 0;
    __syncthreads();
    clones = s_dup != 0 && (flags & kNoSentinel) == 0;
    if (clones) {
      // A flood of clones: only the first n_cl of them (by index) stand for the group in S, the next one is the
      // sentinel, and the freed places go to the members above τ: the search runs again with the other clones struck
      // out and yields τ₂ ≥ τ, the stale bound of everyone outside S who is not a clone.
      int pos = before;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (key[e] == tau) {
          if (pos >= n_cl) key[e] = kDeadKey;
          ++pos;
        }
      int below2;
      bool tie2;
      const unsigned tau2 = find_tau<E>(key, tau, step_prev, s_cnt, flip, bisections, below2, tie2);
      if (tid == 0) { s_tidx = s_tclone; s_next = tau2; }
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (key[e] < tau2) s_idx[atomicAdd(&s_n, 1)] = tid * E + e;
    } else {
      // the members with key == τ join in index order until S is full; the first one left out is T
      int pos = before;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (key[e] < tau) s_idx[atomicAdd(&s_n, 1)] = tid * E + e;
        if (key[e] == tau) {
          if (pos < room) s_idx[below + pos] = tid * E + e;
          ++pos;
        }
      }
    }
  } else {
    // members below τ go to S in any order
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (key[e] < tau) s_idx[atomicAdd(&s_n, 1)] = tid * E + e;
  }
  }  // full scan
  __syncthreads();
  if (tid < kK) {
    const int j = s_idx[tid];
    if (j < n) {
      s_f[tid] = fit[j];
      s_s[tid] = -1.0f / (C[j] * kappa);
#pragma unroll
      for (int k = 0; k < M; ++k) s_obj[k][tid] = obj[(int64_t)j * M + k];
    } else {
      s_s[tid] = 0.f;
#pragma unroll
      for (int k = 0; k < M; ++k) s_obj[k][tid] = 0.f;
    }
  }
  __syncthreads();
  if (tid >= 64) return;

  // ---- simulate: wave 0, no barrier from here on
  const long long t_sim = wall_clock64();
  const bool sentinel = clones;
  const int tau_key = (int)(tau ^ 0x80000000u);            // τ in the signed order of `simulate`
  const int removed = sentinel
      ? simulate<M, true>(s_f, s_idx, s_obj, s_s, s_order, lane, budget, tau_key, s_tidx, (int)(s_next ^ 0x80000000u), obj, fit)
      : simulate<M, false>(s_f, s_idx, s_obj, s_s, s_order, lane, budget, tau_key, s_tidx, 0, obj, fit);
  for (int t = lane; t < removed; t += 64) {
    const int slot = s_order[t];
    float* r = rem + (int64_t)t * (M + 1);
#pragma unroll
    for (int k = 0; k < M; ++k) r[k] = s_obj[k][slot];
    r[M] = s_s[slot];
    fit[s_idx[slot]] = INFINITY;
  }
  if (lane == 0) {
    ctrl[kRemoved] = done + removed;
    ctrl[kRounds] += 1;
    ctrl[kRoundCount] = removed;
    if (tie) ctrl[kTieRounds] += 1;
    if (sentinel) ctrl[kSentinelRounds] += 1;
    if (from_cand) ctrl[kCandRounds] += 1;
    // for the next round: τ, how far it moved, and the guess G below which the apply launch lists candidates
    const unsigned moved = tau_warm != kDeadKey && tau_prev != 0 && tau_warm > tau_prev ? tau_warm - tau_prev : 0;
    unsigned guess = 0;
    if (moved != 0 && !tie) {
      const unsigned long long g = (unsigned long long)tau_warm + (unsigned long long)moved * (unsigned)mult;
      guess = g < kDeadKey ? (unsigned)g : 0;
    }
    if (tau_warm != kDeadKey) { ctrl[kTau] = (int)tau_warm; ctrl[kTauStep] = (int)moved; }
    ctrl[kGuess] = (int)guess;
    ctrl[kGuessMult] = mult;
    ctrl[kCandCount] = 0;
    // time spent (100 MHz ticks) finding S and playing the removals, and bisection steps: for tuning
    ctrl[kSelectTicks] += (int)(t_sim - t_start);
    ctrl[kSimTicks] += (int)(wall_clock64() - t_sim);
    ctrl[kBisections] += bisections;
  }
}

// fit[j] += Σ_t exp(I(x_t, j) · s_t) over the members removed in the round, in removal order.  Members whose new key is
// below the guess G that the select launch left in ctrl are listed in `cand` as (key, index) for the next select launch.
template <int M>
__global__ __launch_bounds__(256) void ibea_lazy_apply_kernel(const float* __restrict__ obj, int n, float* __restrict__ fit,
                                                              int* __restrict__ ctrl, const float* __restrict__ rem,
                                                              int* __restrict__ cand) {
  __shared__ float s_rem[kK * (M + 1)];
  const int count = ctrl[kRoundCount];
  if (count <= 0) return;
  const unsigned guess = (unsigned)ctrl[kGuess];
  for (int i = threadIdx.x; i < count * (M + 1); i += 256) s_rem[i] = rem[i];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  float f = j < n ? fit[j] : INFINITY;
  const bool live = f < INFINITY;
  if (live) {
    float oj[M];
#pragma unroll
    for (int k = 0; k < M; ++k) oj[k] = obj[(int64_t)j * M + k];
    for (int t = 0; t < count; ++t) {
      float ox[M];
#pragma unroll
      for (int k = 0; k < M; ++k) ox[k] = s_rem[t * (M + 1) + k];
      f += __expf(eps_ind<M>(ox, oj) * s_rem[t * (M + 1) + M]);
    }
    fit[j] = f;
  }
  const unsigned key = live ? member_key(f) : kDeadKey;
  const bool listed = guess != 0 && key < guess;
  const unsigned long long mask = __ballot(listed);
  if (mask != 0) {                                        // one atomic per wave
    const int lane = threadIdx.x & 63, leader = (int)__builtin_ctzll(mask);
    int base = 0;
    if (lane == leader) base = atomicAdd(&ctrl[kCandCount], (int)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (listed) {
      const int pos = base + (int)__popcll(mask & ((1ull << lane) - 1ull));
      if (pos < kCandCap) { cand[2 * pos] = (int)key; cand[2 * pos + 1] = j; }
    }
  }
}

__global__ __launch_bounds__(64) void ibea_lazy_reset_kernel(int* ctrl) {
  if (threadIdx.x < kCtrlWords) ctrl[threadIdx.x] = 0;
}

template <int M, int E>
int launch_rounds(const float* obj, int n, const float* C, float kappa, float* fit, int n_remove, int* ctrl, float* rem,
                  int* cand, int rounds, int flags, hipStream_t st) {
  const int blocks = (n + 255) / 256;
  for (int r = 0; r < rounds; ++r) {
    ibea_lazy_select_kernel<M, E><<<1, kSel, 0, st>>>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, flags);
    ibea_lazy_apply_kernel<M><<<blocks, 256, 0, st>>>(obj, n, fit, ctrl, rem, cand);
  }
  return last_launch_status();
}

template <int M>
int launch_rounds_any(const float* obj, int n, const float* C, float kappa, float* fit, int n_remove, int* ctrl, float* rem,
                      int* cand, int rounds, int flags, hipStream_t st) {
  if (n <= 4 * kSel) return launch_rounds<M, 4>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 8 * kSel) return launch_rounds<M, 8>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 12 * kSel) return launch_rounds<M, 12>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 16 * kSel) return launch_rounds<M, 16>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 20 * kSel) return launch_rounds<M, 20>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 24 * kSel) return launch_rounds<M, 24>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 32 * kSel) return launch_rounds<M, 32>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  if (n <= 64 * kSel) return launch_rounds<M, 64>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
  return (int)hipErrorInvalidValue;
}

}  // namespace

// Largest population of the round-wise removal.
EVOX_API int64_t evox_hip_ibea_lazy_max(int m) { return m >= 2 && m <= 8 ? (int64_t)64 * kSel : 0; }

// Floats of the `rem` scratch and int32 words of `ctrl`.
EVOX_API int64_t evox_hip_ibea_lazy_rem_floats(int m) { return (int64_t)kK * (m + 1); }
EVOX_API int64_t evox_hip_ibea_lazy_ctrl_words() { return kCtrlWords; }
EVOX_API int64_t evox_hip_ibea_lazy_cand_words() { return 2 * kCandCap; }

// reset != 0: start a removal (zeroes ctrl).  Then `rounds` rounds are issued; rounds after the n_remove-th removal do
// nothing.  fit: (n) working copy, removed members end at +inf.  ctrl[0] = members removed so far, ctrl[1] = rounds
// that did work, ctrl[3] = rounds that needed the tie order, ctrl[9] = of those, rounds with a sentinel.  flags: 1 = no
// sentinel rounds, 2 = no candidate lists.  cand: scratch of evox_hip_ibea_lazy_cand_words() int32.
EVOX_API int evox_hip_ibea_lazy_rounds(const float* obj, int n, int m, const float* C, float kappa, float* fit, int n_remove,
                                       int* ctrl, float* rem, int* cand, int reset, int rounds, int flags, void* stream) {
  if (n <= 0 || rounds <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (reset) ibea_lazy_reset_kernel<<<1, 64, 0, st>>>(ctrl);
  switch (m) {
    case 2: return launch_rounds_any<2>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 3: return launch_rounds_any<3>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 4: return launch_rounds_any<4>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 5: return launch_rounds_any<5>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 6: return launch_rounds_any<6>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 7: return launch_rounds_any<7>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    case 8: return launch_rounds_any<8>(obj, n, C, kappa, fit, n_remove, ctrl, rem, cand, rounds, flags, st);
    default: return (int)hipErrorInvalidValue;
  }
}
