// Non-dominated sorting of two or three objectives in O(N log^2 N) per front instead of O(N^2) pair tests.
//
// The caller sorts the points lexicographically by (f1, f2, f3), keeps one representative of every group of identical
// points, and pads the arrays to a power of two (pad entries: f2 = f3 = +inf, never alive).  Among distinct points in that
// order, j dominates i  <=>  j comes before i  and  f2_j <= f2_i  and  f3_j <= f3_i.  A point of the current front is
// therefore one whose
//
//     g(i) = min { f3_j : j alive, j < i, f2_j <= f2_i }
//
// exceeds f3_i.  g is a two-sided range minimum and decomposes over the levels of a merge sort by f2 of the index order:
// at level l the indices form segments of 2^l consecutive points, each held sorted by f2; a point i in a right-hand
// segment (odd segment number) sees its left sibling through ONE number — the minimum of f3 over the sibling's first
// pos_l(i) entries in f2 order, pos_l(i) = #{ j in sibling : f2_j <= f2_i }.  The orders and the positions depend on
// the points only, so they are built once (one merge pass per level); a peeling round recomputes the segmented prefix
// minima over the points still alive (one pass over L·N values), looks g up with at most L gathers per point, and marks
// the points that are not dominated.  Two objectives use the same code with f3 = 0.
#include "hip_common.h"

namespace {

constexpr int kChunk = 2048;      // elements per workgroup in the scan pass (256 threads x 8)
constexpr int kChunkLog = 11;
constexpr float kInf = __builtin_inff();

// ---- static structure --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_level0_kernel(const float* __restrict__ f2, int npad, int* __restrict__ ord,
                                                           float* __restrict__ key) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npad) return;
  ord[k] = k;
  key[k] = f2[k];
}

// Level l (segments of 2^l, sorted by f2) -> level l+1.  One thread per element: its place in the merged order is its
// place in its own segment plus the number of the sibling's keys in front of it (left before right on ties, so the merge
// is stable), found by binary search.  An element of a right-hand segment also records that number — pos_l.
__global__ __launch_bounds__(256) void sweep_merge_kernel(const int* __restrict__ ord_in, const float* __restrict__ key_in,
                                                          int npad, int level, int* __restrict__ ord_out,
                                                          float* __restrict__ key_out, int* __restrict__ pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npad) return;
  const int size = 1 << level;
  const int seg = k >> level, local = k & (size - 1);
  const float v = key_in[k];
  const int e = ord_in[k];
  const bool right = seg & 1;
  const float* sib = key_in + ((size_t)(seg ^ 1) << level);
  int lo = 0, hi = size;               // count of sibling keys < v (left element) or <= v (right element)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float s = sib[mid];
    const bool before = right ? (s <= v) : (s < v);
    if (before) lo = mid + 1; else hi = mid;
  }
  const int out = ((seg >> 1) << (level + 1)) + local + lo;
  ord_out[out] = e;
  key_out[out] = v;
  pos[e] = right ? lo : 0;
}

// ---- one peeling round ---------------------------------------------------------------------------------------------
// Segmented inclusive prefix minimum of f3 over the alive points, in the f2 order of level blockIdx.y.  Segments longer
// than a chunk: the chunk's own prefix minimum, plus its total in agg (combined by sweep_chunks_kernel).
__global__ __launch_bounds__(256) void sweep_scan_kernel(const int* __restrict__ ord, const float* __restrict__ f3,
                                                         const int* __restrict__ alive, int npad, float* __restrict__ pm,
                                                         float* __restrict__ agg, const int* __restrict__ state, int target) {
  // the sort is complete: later rounds of a batch do nothing (nothing writes state[0] while this pass runs)
  if (state[0] >= target) return;
  const int level = blockIdx.y;
  const size_t base = (size_t)level * npad + (size_t)blockIdx.x * kChunk;
  const int t = threadIdx.x;
  const int* o = ord + base + t * 8;
  float v[8];
  const int4 i0 = *reinterpret_cast<const int4*>(o), i1 = *reinterpret_cast<const int4*>(o + 4);
  const int idx[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    // keys must be totally ordered (no NaN: the caller replaces them) for every level to be a permutation; an index
    // out of range — a slot that no merge wrote — is treated as a dead point rather than dereferenced
    const bool ok = (unsigned)idx[q] < (unsigned)npad;
    v[q] = (ok && alive[ok ? idx[q] : 0]) ? f3[ok ? idx[q] : 0] : kInf;
  }
  const int size = 1 << level;
  if (size < 8) {
#pragma unroll
    for (int q = 1; q < 8; ++q)
      if (q & (size - 1)) v[q] = fminf(v[q], v[q - 1]);
  } else {
#pragma unroll
    for (int q = 1; q < 8; ++q) v[q] = fminf(v[q], v[q - 1]);
    // across the threads of one segment (size/8 of them, at most the whole block)
    __shared__ float tot[256];
    const int group = size >= kChunk ? 256 : (size >> 3);
    const int lane = t & (group - 1);
    float run = v[7];
    tot[t] = run;
    __syncthreads();
    for (int off = 1; off < group; off <<= 1) {
      const float other = lane >= off ? tot[t - off] : kInf;
      __syncthreads();
      run = fminf(run, other);
      tot[t] = run;
      __syncthreads();
    }
    const float before = lane > 0 ? tot[t - 1] : kInf;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = fminf(v[q], before);
    if (size > kChunk && t == 255) agg[(size_t)level * (npad >> kChunkLog) + blockIdx.x] = v[7];
  }
  float* p = pm + base + t * 8;
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// Exclusive prefix minimum of the chunk totals inside every segment longer than a chunk (levels > kChunkLog).
// Also copies the ranked count into state[1]: the query pass, whose blocks add to state[0] as they go, must take its
// "already complete" verdict from a value that no block of its own launch changes.
__global__ __launch_bounds__(256) void sweep_chunks_kernel(const float* __restrict__ agg, float* __restrict__ cpm, int npad,
                                                           int levels, int* __restrict__ state, int target) {
  const int ranked = state[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) state[1] = ranked;
  if (ranked >= target) return;
  const int chunks = npad >> kChunkLog;
  for (int level = kChunkLog + 1 + blockIdx.x; level < levels; level += gridDim.x) {
    const int per = 1 << (level - kChunkLog);           // chunks per segment
    for (int s = threadIdx.x; s < chunks / per; s += blockDim.x) {
      float run = kInf;
      for (int c = 0; c < per; ++c) {
        const size_t at = (size_t)level * chunks + (size_t)s * per + c;
        cpm[at] = run;
        run = fminf(run, agg[at]);
      }
    }
  }
}

// g(i) against f3_i for every alive point; the non-dominated ones form front `round`.
// state: [0] points ranked so far (identical points counted), [1] that count before this round, [2 + r] size of front r.
__global__ __launch_bounds__(256) void sweep_query_kernel(const int* __restrict__ pos, const float* __restrict__ pm,
                                                          const float* __restrict__ cpm, const float* __restrict__ f3,
                                                          int* __restrict__ alive, const int* __restrict__ gsize, int npad,
                                                          int levels, int round, int* __restrict__ rank,
                                                          int* __restrict__ state, int target) {
  if (state[1] >= target) return;                      // the count as it stood before this round (sweep_chunks_kernel)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int mine = 0;
  if (i < npad && alive[i]) {
    float g = kInf;
    const int chunks = npad >> kChunkLog;
    for (int level = 0; level < levels; ++level) {
      const int seg = i >> level;
      if (!(seg & 1)) continue;
      const int p = pos[(size_t)level * npad + i];
      if (p == 0) continue;
      const int at = ((seg - 1) << level) + p - 1;
      g = fminf(g, pm[(size_t)level * npad + at]);
      if (level > kChunkLog) g = fminf(g, cpm[(size_t)level * chunks + (at >> kChunkLog)]);
    }
    if (!(g <= f3[i])) {
      rank[i] = round;
      alive[i] = 0;
      mine = gsize[i];
    }
  }
  // one atomic per wavefront
  int total = mine;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
  if ((threadIdx.x & 63) == 0 && total > 0) {
    atomicAdd(&state[0], total);
    atomicAdd(&state[2 + round], total);
  }
}

}  // namespace

#ifndef EVOX_CPU_SHIM   // the launches below need the GPU tool chain; tools/cpu_shim drives the kernels directly
EVOX_API int evox_hip_nds_sweep_chunk() { return kChunk; }

// f2: npad floats (npad a power of two >= 2048, pad entries +inf).  ord / key: levels x npad (level l = segments of 2^l
// sorted by f2; levels = log2(npad)); pos: levels x npad.
EVOX_API int evox_hip_nds_sweep_build(const float* f2, int npad, int levels, int* ord, float* key, int* pos, void* stream) {
  if (npad < kChunk || (npad & (npad - 1)) || (1 << levels) != npad) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = npad / 256;
  sweep_level0_kernel<<<blocks, 256, 0, st>>>(f2, npad, ord, key);
  for (int level = 0; level < levels; ++level) {
    // the last pass only yields pos of the top level (its merged order is not needed): write it into scratch rows
    int* ord_out = level + 1 < levels ? ord + (size_t)(level + 1) * npad : ord + (size_t)levels * npad;
    float* key_out = level + 1 < levels ? key + (size_t)(level + 1) * npad : key + (size_t)levels * npad;
    sweep_merge_kernel<<<blocks, 256, 0, st>>>(ord + (size_t)level * npad, key + (size_t)level * npad, npad, level, ord_out,
                                               key_out, pos + (size_t)level * npad);
  }
  return evox::last_launch_status();
}

// `rounds` peeling rounds (fronts first_round ...).  state: 2 + max_rounds ints, zeroed before the first call:
// [0] ranked so far, [1] scratch, [2 + r] size of front r.  A round issued after `target` points are ranked changes
// nothing.  ord needs (levels + 1) x npad ints and key the same number of floats (see evox_hip_nds_sweep_build).
EVOX_API int evox_hip_nds_sweep_rounds(const int* ord, const int* pos, const float* f3, int* alive, const int* gsize,
                                       int npad, int levels, float* pm, float* agg, float* cpm, int* rank, int* state,
                                       int first_round, int rounds, int target, void* stream) {
  if (npad < kChunk || (npad & (npad - 1)) || (1 << levels) != npad || rounds < 1 || first_round < 0)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = npad >> kChunkLog;
  for (int r = first_round; r < first_round + rounds; ++r) {
    const int upper = levels - kChunkLog - 1;            // levels whose segments span several chunks
    sweep_scan_kernel<<<dim3(chunks, levels), 256, 0, st>>>(ord, f3, alive, npad, pm, agg, state, target);
    sweep_chunks_kernel<<<upper > 0 ? upper : 1, 256, 0, st>>>(agg, cpm, npad, levels, state, target);
    sweep_query_kernel<<<npad / 256, 256, 0, st>>>(pos, pm, cpm, f3, alive, gsize, npad, levels, r, rank, state, target);
  }
  return evox::last_launch_status();
}
#endif
