// Differential-evolution variation in one pass: parent picks, mutation, binomial crossover, bounds.
//
//   picks   k distinct indices != row, drawn like evox_amd.core.random.choice_distinct:
//           r_j = mulhi(bits(key_sel, row * k + j), N - 1 - j), shifted past the already taken values
//           in ascending order (exact sampling without replacement, O(k^2) per row, k <= 9)
//   base    primary / secondary base vectors: rand (= pick 0), best, pbest (given row), current
//   mutant  p + F (s - p) + F * sum_t (x[pick 2t+1] - x[pick 2t+2])
//   trial   mutant where u(key_mask, row * D + c) < CR or c == jrand(key_jrand, row), else current;
//           clipped to [lb, ub]
//
// The random numbers are the same Philox elements the PyTorch formulation draws, so both produce the
// same trial vectors (up to floating-point association).  A row is handled by LPR lanes of a
// wavefront (coalesced row gathers); the picks are recomputed by each lane - a few dozen integer ops.
//
// Blocked populations (`blocks` > 1): a row of the n_pop x (blocks * dim) population is `blocks` independent
// sub-vectors, each evolving with its own picks, crossover mask, F and CR — the stacked members of a decomposition
// container.  The kernel then works on virtual rows v = row * blocks + block (the population viewed as
// (n_pop * blocks) x dim): picks are individuals in [0, n_pop), mapped to the virtual row of the same block; `rows`,
// F, CR, the pbest rows and the random elements are indexed by the virtual row, lb / ub by block * dim + column;
// best_per_block: best_index holds one individual per block (blocks with a fitness of their own).
#include "hip_common.h"
#include "philox.h"

using namespace evox;

namespace {

constexpr int kMaxPicks = 9;
enum Base { kRand = 0, kBest = 1, kPbest = 2, kCurrent = 3 };
enum Cross { kBin = 0, kExp = 1, kArith = 2 };            // kExp is not handled here (callers keep it on the tensor path)

__device__ __forceinline__ uint32_t bits_at(uint32_t k0, uint32_t k1, uint64_t e) {
  const u32x4 b = philox_block(k0, k1, e >> 2, kStreamDraw);
  switch ((int)(e & 3)) { case 0: return b.x; case 1: return b.y; case 2: return b.z; default: return b.w; }
}

template <int LPR>
__global__ __launch_bounds__(256) void de_trial_kernel(
    const float* __restrict__ pop, int64_t ldpop, int n_pop, int dim, const int64_t* __restrict__ rows, int b,
    const int64_t* __restrict__ best_index, const int64_t* __restrict__ pbest_rows, const float* __restrict__ F, int f_is_vec,
    const float* __restrict__ CR, int cr_is_vec, int prim, int sec, int num_diff, int cross,
    const int32_t* __restrict__ row_codes, int n_picks, const int32_t* __restrict__ key_sel, const int32_t* __restrict__ key_jrand, const int32_t* __restrict__ key_mask,
    const float* __restrict__ lb, const float* __restrict__ ub, float* __restrict__ out, int blocks, int best_per_block) {
  constexpr int RPB = 256 / LPR;                       // rows per block
  const int sub = threadIdx.x % LPR;
  const int r = blockIdx.x * RPB + threadIdx.x / LPR;
  if (r >= b) return;
  const int64_t self_v = rows ? rows[r] : r;           // virtual row: individual * blocks + block
  const int64_t self = self_v / blocks;
  const int64_t blk = self_v - self * blocks;
  lb += blk * dim;
  ub += blk * dim;
  if (row_codes) {                                     // per-row strategy: (primary, secondary, #differences, crossover)
    prim = row_codes[4 * r]; sec = row_codes[4 * r + 1]; num_diff = row_codes[4 * r + 2]; cross = row_codes[4 * r + 3];
  }

  // ---- the row's random draws, one per lane: draw j < n_picks is pick j, draw kMaxPicks the crossover column.  A
  // Philox block is a hundred instructions; every lane of the row computing all ten draws made the kernel bound by
  // them.  The lanes of a row are neighbours in the wave, so the draws are handed round with shuffles.
  const uint32_t s0 = (uint32_t)key_sel[0], s1 = (uint32_t)key_sel[1];
  const uint32_t j0 = (uint32_t)key_jrand[0], j1 = (uint32_t)key_jrand[1];
  constexpr int kDraws = kMaxPicks + 1;
  constexpr int ROUNDS = (kDraws + LPR - 1) / LPR;
  uint32_t mine[ROUNDS];
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    const int dno = q * LPR + sub;
    // one Philox evaluation for the wave: the lanes differ in key, element and range, not in code
    const bool pick = dno < n_picks, col = dno == kMaxPicks;
    const uint32_t ka = pick ? s0 : j0, kb = pick ? s1 : j1;
    const uint64_t e = pick ? (uint64_t)r * n_picks + dno : (uint64_t)r;
    const uint32_t range = pick ? (uint32_t)(n_pop - 1 - dno) : (uint32_t)dim;
    mine[q] = (pick || col) ? bits_to_range(bits_at(ka, kb, e), range) : 0u;
  }
  const int row_lane0 = (int)(threadIdx.x & 63) - sub;       // first lane of this row in its wave
  // ---- distinct picks (ascending list `taken` starts with the row itself) -------------------
  int picks[kMaxPicks];
  int taken[kMaxPicks + 1];
  taken[0] = (int)self;
#pragma unroll
  for (int j = 0; j < kMaxPicks; ++j) {
    picks[j] = 0;
    if (j < n_picks) {                                         // uniform
      int v = (int)__shfl(mine[j / LPR], row_lane0 + j % LPR, 64);
#pragma unroll
      for (int t = 0; t <= j; ++t) v += (v >= taken[t]) ? 1 : 0;      // `taken` holds j + 1 values, ascending
      picks[j] = v;
      // insert, keeping `taken` ascending
#pragma unroll
      for (int t = j + 1; t > 0; --t) {
        const bool up = taken[t - 1] > v;                      // everything above v moves one place up
        if (t == j + 1) taken[t] = up ? taken[t - 1] : v;
        else taken[t] = up ? taken[t - 1] : (taken[t] > v ? v : taken[t]);
      }
      if (taken[0] > v) taken[0] = v;
    }
  }
  const int jrand = (int)__shfl(mine[kMaxPicks / LPR], row_lane0 + kMaxPicks % LPR, 64);
  const uint32_t m0 = (uint32_t)key_mask[0], m1 = (uint32_t)key_mask[1];
  const float f = f_is_vec ? F[r] : F[0];
  const float cr = cr_is_vec ? CR[r] : CR[0];

  auto base_row = [&](int code) -> const float* {
    switch (code) {
      case kRand: return pop + ((int64_t)picks[0] * blocks + blk) * ldpop;
      case kBest: return pop + (best_index[best_per_block ? blk : 0] * blocks + blk) * ldpop;
      case kPbest: return pop + (pbest_rows[r] * blocks + blk) * ldpop;
      default: return pop + self_v * ldpop;
    }
  };
  const float* p_row = base_row(prim);
  const float* s_row = base_row(sec);
  const float* cur = pop + self_v * ldpop;
  const float* d1a = pop + ((int64_t)picks[1] * blocks + blk) * ldpop;
  const float* d1b = pop + ((int64_t)picks[2] * blocks + blk) * ldpop;
  const float* d2a = num_diff > 1 ? pop + ((int64_t)picks[3] * blocks + blk) * ldpop : cur;
  const float* d2b = num_diff > 1 ? pop + ((int64_t)picks[4] * blocks + blk) * ldpop : cur;

  for (int c = sub; c < dim; c += LPR) {
    const float p = p_row[c];
    float diff = d1a[c] - d1b[c];
    if (num_diff > 1) diff = diff + (d2a[c] - d2b[c]);
    const float mutant = p + f * (s_row[c] - p) + f * diff;
    float x;
    if (cross == kArith) {
      x = cur[c] + cr * (mutant - cur[c]);               // arithmetic recombination, CR plays K
    } else {
      const float u = bits_to_unit(bits_at(m0, m1, (uint64_t)r * dim + c));
      x = (u < cr || c == jrand) ? mutant : cur[c];
    }
    out[(int64_t)r * dim + c] = fminf(fmaxf(x, lb[c]), ub[c]);
  }
}

}  // namespace

// pop: (n_pop * blocks) virtual rows of `dim` columns, ldpop apart (blocks = 1: the plain population); rows: b virtual rows.
EVOX_API int evox_hip_de_trial(const float* pop, int64_t ldpop, int n_pop, int dim, const int64_t* rows, int b,
                               const int64_t* best_index, const int64_t* pbest_rows, const float* F, int f_is_vec,
                               const float* CR, int cr_is_vec, int prim, int sec, int num_diff, int cross,
                               const int32_t* row_codes, int n_picks, const int32_t* key_sel, const int32_t* key_jrand, const int32_t* key_mask, const float* lb,
                               const float* ub, float* out, int blocks, int best_per_block, void* stream) {
  if (b <= 0 || dim <= 0) return 0;
  if (blocks < 1) return (int)hipErrorInvalidValue;
  if (row_codes) num_diff = 2;                          // per-row values are 1 or 2 (checked by the caller)
  if (n_picks > kMaxPicks || n_picks < 2 * num_diff + 1 || num_diff < 1 || num_diff > 2 || n_pop <= n_picks)
    return (int)hipErrorInvalidValue;
  if (!row_codes && cross != kBin && cross != kArith) return (int)hipErrorInvalidValue;
  if ((row_codes || prim == kPbest || sec == kPbest) && pbest_rows == nullptr) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
#define EVOX_DE_LAUNCH(LPR)                                                                                            \
  de_trial_kernel<LPR><<<(b + 256 / LPR - 1) / (256 / LPR), 256, 0, s>>>(                                              \
      pop, ldpop, n_pop, dim, rows, b, best_index, pbest_rows, F, f_is_vec, CR, cr_is_vec, prim, sec, num_diff, cross,   \
      row_codes, n_picks,                                                                                              \
      key_sel, key_jrand, key_mask, lb, ub, out, blocks, best_per_block)
  if (dim <= 8) EVOX_DE_LAUNCH(8);
  else if (dim <= 16) EVOX_DE_LAUNCH(16);
  else if (dim <= 32) EVOX_DE_LAUNCH(32);
  else EVOX_DE_LAUNCH(64);
#undef EVOX_DE_LAUNCH
  return last_launch_status();
}
