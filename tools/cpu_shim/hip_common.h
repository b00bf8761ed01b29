// Host stand-in for the HIP kernel language: enough of it to run a simple kernel on the CPU, one workgroup at a time
// with one OS thread per work-item, so that address and undefined-behaviour sanitizers can watch it
// (g++ -fsanitize=address,undefined).  Covers: thread/block indices, __syncthreads, __shared__ arrays (function
// statics, valid because workgroups run one after the other), integer atomicAdd, wave64 __shfl_down, int4/float4.
// Kernels are called through shim::launch(grid, block, [&]{ kernel(args...); }).
#pragma once
#include <barrier>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <thread>
#include <vector>

#define EVOX_CPU_SHIM 1
#define EVOX_API extern "C"
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }

inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace shim {
inline std::barrier<>* bar = nullptr;
inline int xchg[1024];
template <class F>
void launch(dim3 grid, dim3 block, F body) {
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      std::barrier<> b((std::ptrdiff_t)block.x);
      bar = &b;
      std::vector<std::thread> ts;
      for (unsigned t = 0; t < block.x; ++t)
        ts.emplace_back([&, t] {
          threadIdx = dim3(t); blockIdx = dim3(bx, by); blockDim = block; gridDim = grid;
          body();
          bar->arrive_and_drop();
        });
      for (auto& th : ts) th.join();
    }
}
}  // namespace shim
inline void __syncthreads() { shim::bar->arrive_and_wait(); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int __shfl_down(int v, int off, int /*width*/) {
  const int t = threadIdx.x;
  shim::xchg[t] = v;
  __syncthreads();
  const int r = ((t & 63) + off < 64 && t + off < (int)blockDim.x) ? shim::xchg[t + off] : v;
  __syncthreads();
  return r;
}
namespace evox { inline int last_launch_status() { return 0; } }
