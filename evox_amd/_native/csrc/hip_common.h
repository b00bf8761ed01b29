// Small helpers shared by every gfx950 kernel file in this directory.
#pragma once
#ifndef EVOX_CPU_SHIM   // tools/cpu_shim/hip_common.h stands in for this file when a kernel is run on the CPU under sanitizers
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EVOX_API extern "C" __attribute__((visibility("default")))

namespace evox {

constexpr int kWave = 64;           // CDNA wavefront width
constexpr int kNumCU = 256;         // MI355X
constexpr int kNumXCD = 8;

inline int last_launch_status() { return (int)hipGetLastError(); }

// grid for a grid-stride elementwise pass: enough blocks to fill the chip, capped
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  const int64_t cap = (int64_t)kNumCU * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Squared Euclidean distance of two short vectors with a FIXED rounding sequence: the first square is rounded on its own,
// every further one is fused into the running sum.  Written out because `acc + d * d` leaves it to the compiler which
// of two products it fuses, and kernels that must agree bit for bit on a distance (a distance matrix here, a streaming
// nearest-neighbour scan there) would then differ in the last place now and then.
template <int M>
__device__ __forceinline__ float squared_distance(const float (&a)[M], const float (&b)[M]) {
  float d = a[0] - b[0];
  float acc = __fmul_rn(d, d);
#pragma unroll
  for (int k = 1; k < M; ++k) {
    d = a[k] - b[k];
    acc = fmaf(d, d, acc);
  }
  return acc;
}

// Remap a linear workgroup id so that each XCD (which owns a private L2) receives a
// contiguous chunk of the tile space.  Workgroups are dealt round-robin over the 8 XCDs,
// so ids congruent mod 8 share an L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg / kNumXCD, r = nwg % kNumXCD;
  const int xcd = orig % kNumXCD, idx = orig / kNumXCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace evox
#endif  // EVOX_CPU_SHIM
