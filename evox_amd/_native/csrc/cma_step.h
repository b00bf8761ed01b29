// Argument block of the native CMA-ES tell (cma_step.hip, cma_fast.hip); mirrored field by field by ops/cma_native.py.
#pragma once
#include <stdint.h>

struct CmaTell {
  // problem and workspace (constant for a given instance)
  int n, lam, mu, rb, ns_planes;
  int64_t ldn, ldmu, cov_ws_floats;
  float consts[8];           // cs, cc, mueff, cm, damps, chiN, c1, cmu
  const float* rowscale;     // sqrt of the recombination weights (mu)
  void* yt;                  // n x ldmu bf16
  float* colsum;             // rb x n
  float* cov_ws;
  int64_t* order;            // lam
  int* sort_scratch;         // evox_hip_rank_sort_scratch(lam) ints
  float* step;               // n
  float* whitened;           // n
  void* planes;              // Newton-Schulz planes
  float* ns_f32;             // 3 n^2
  float* vec4;               // 4 n
  float* scal;
  float* scal2;
  float* lib_work;           // 6 n^2 (fp32 library driver), or null
  const float* lib_eye;      // n^2 identity, or null
  // inputs of the call
  const float* fitness;
  const float* pop;
  int64_t ldpop;
  const float *mean, *ps, *pc, *sigma, *C, *B, *invsqrtC, *iter_dev;
  const float* ctrl_vec;     // 2 x n power vectors of the factorisation control
  const float* prev_rec;     // record of the previous call (7 floats): its condition estimates are carried over
  // outputs
  float *mean_o, *ps_o, *pc_o, *scalars_o;   // scalars: sigma', alpha, hsig, |ps|, then n floats of scratch
  float *C_o, *B_o, *inv_o;
  void* b16_o;               // n x ldn bf16 image of B_o (symmetric driver), or null
  float* ctrl_vec_o;         // 2 x n
  float* rec_o;              // 7
  // factorisation control
  int factor;                // 0: skip the factorisation (decomp_per_iter > 1)
  int ns_iters, ns_power, ns_safe, ns_symmetric, ns_cond, ns_first_poly4, ns_tune;
  float* rec_host;           // device-visible address of 7 pinned floats, or null
  // row-sharded generation: the selected rows were reduced on every rank and summed over the ranks beforehand; `step`
  // then is an INPUT (the summed mean shift) and `gram` the summed rank-mu term
  int sharded;
  const float* gram;         // n x n, or its tile-packed upper triangle (gram_packed)
  const float* poison;       // one float (> 0: rows were dropped somewhere), or null
  int gram_packed;           // gram holds evox_hip_packed_upper_floats(n) floats (64 x 64 upper tiles)
};
