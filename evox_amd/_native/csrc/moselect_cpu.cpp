// Host versions of the one-at-a-time selection loops of IBEA, SPEA2 and SRA (same arithmetic
// as moselect.hip; used on the CPU path and as the oracle of the GPU kernels).
#include <math.h>
#include <stdint.h>
#include <vector>

#include "sr_coins.h"

extern "C" {

// alive[j] = 1 for survivors.  fit is updated in place.
void evox_cpu_ibea_remove(float* fit, const float* I, const float* C, float kappa, int64_t n, int64_t n_remove,
                          uint8_t* alive) {
  for (int64_t j = 0; j < n; ++j) alive[j] = 1;
  for (int64_t r = 0; r < n_remove; ++r) {
    int64_t x = -1;
    float best = INFINITY;
    for (int64_t j = 0; j < n; ++j)
      if (alive[j] && (x < 0 || fit[j] < best)) { best = fit[j]; x = j; }
    if (x < 0) break;
    const float s = 1.0f / (C[x] * kappa);
    const float* row = I + x * n;
    for (int64_t j = 0; j < n; ++j) fit[j] += expf(-row[j] * s);
    alive[x] = 0;
  }
}

// dis: n x n distances with +inf where a pair does not take part (masked rows/cols, diagonal).
// keep[j] in: candidate mask, out: survivors.
void evox_cpu_spea2_truncate(float* dis, int64_t n, int64_t n_remove, uint8_t* keep) {
  std::vector<float> rowmin(n);
  std::vector<int64_t> nn(n);
  auto scan = [&](int64_t i) {
    float m = INFINITY; int64_t a = -1;
    const float* row = dis + i * n;
    for (int64_t j = 0; j < n; ++j) if (row[j] < m) { m = row[j]; a = j; }
    rowmin[i] = m; nn[i] = a;
  };
  for (int64_t i = 0; i < n; ++i) scan(i);
  for (int64_t r = 0; r < n_remove; ++r) {
    int64_t x = 0; float best = rowmin[0];
    for (int64_t i = 1; i < n; ++i) if (rowmin[i] < best) { best = rowmin[i]; x = i; }
    keep[x] = 0;
    for (int64_t j = 0; j < n; ++j) { dis[x * n + j] = INFINITY; dis[j * n + x] = INFINITY; }
    rowmin[x] = INFINITY; nn[x] = -1;
    for (int64_t i = 0; i < n; ++i) if (nn[i] == x) scan(i);
  }
}

// Stochastic bubble sort: at most ceil(n/2) sweeps; in sweep position j the pair is compared by
// I1 when u[j] < pc, else by I2; the larger value moves towards the front.  rank: out (n).
void evox_cpu_stochastic_ranking(const float* I1, const float* I2, const float* u, float pc, int64_t n, int64_t* rank) {
  for (int64_t i = 0; i < n; ++i) rank[i] = i;
  const int64_t sweeps = (n + 1) / 2;
  for (int64_t s = 0; s < sweeps; ++s) {
    bool swapped = false;
    for (int64_t j = 0; j + 1 < n; ++j) {
      const float* key = u[j] < pc ? I1 : I2;
      if (key[rank[j]] < key[rank[j + 1]]) {
        const int64_t t = rank[j]; rank[j] = rank[j + 1]; rank[j + 1] = t;
        swapped = true;
      }
    }
    if (!swapped) break;
  }
}

// The same sort with a fresh coin for every comparison (sr_coins.h): comparison (sweep s, position j) ranks by I1 when
// coin(seed, s, j), else by I2.
void evox_cpu_stochastic_ranking_fresh(const float* I1, const float* I2, const int32_t* seed_ptr, float pc, int64_t n,
                                       int64_t* rank) {
  const uint32_t seed = (uint32_t)*seed_ptr, pc7 = evox::sr_pc7(pc);
  for (int64_t i = 0; i < n; ++i) rank[i] = i;
  const int64_t sweeps = (n + 1) / 2;
  for (int64_t s = 0; s < sweeps; ++s) {
    bool swapped = false;
    for (int64_t j = 0; j + 1 < n; ++j) {
      const float* key = evox::sr_coin(seed, (uint32_t)s, (uint32_t)j, pc7) ? I1 : I2;
      if (key[rank[j]] < key[rank[j + 1]]) {
        const int64_t t = rank[j]; rank[j] = rank[j + 1]; rank[j + 1] = t;
        swapped = true;
      }
    }
    // no early exit: a sweep without swaps under one set of coins says nothing about the next set
    (void)swapped;
  }
}

}  // extern "C"
