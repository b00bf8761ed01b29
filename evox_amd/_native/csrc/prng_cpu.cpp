// Host twin of the Philox draw kernels (bit-identical to prng.hip by construction:
// both include philox.h and evaluate the same expression per element).
#include "philox.h"
#include <stdint.h>

using namespace evox;

namespace {
template <typename F>
inline void for_each_block(int64_t n, int64_t offset, F&& f) {
  if (n <= 0) return;
  const uint64_t first = (uint64_t)offset >> 2, last = (uint64_t)(offset + n - 1) >> 2;
  const int64_t nblk = (int64_t)(last - first + 1);
#pragma omp parallel for schedule(static) if (nblk > 16384)
  for (int64_t t = 0; t < nblk; ++t) f(first + (uint64_t)t);
}
}  // namespace

extern "C" {

void evox_cpu_random_bits(const int32_t* key, uint32_t* out, int64_t n, int64_t offset) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  for_each_block(n, offset, [&](uint64_t blk) {
    u32x4 b = philox_block(k0, k1, blk, kStreamDraw);
    for (int l = 0; l < 4; ++l) {
      int64_t e = (int64_t)(blk * 4 + l) - offset;
      if (e >= 0 && e < n) out[e] = lane_of(b, l);
    }
  });
}

void evox_cpu_uniform(const int32_t* key, float* out, int64_t n, int64_t offset, float lo, float hi) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const float scale = hi - lo;
  for_each_block(n, offset, [&](uint64_t blk) {
    f32x4 v = unit4(philox_block(k0, k1, blk, kStreamDraw));
    for (int l = 0; l < 4; ++l) {
      int64_t e = (int64_t)(blk * 4 + l) - offset;
      if (e >= 0 && e < n) out[e] = fmaf(lane_of(v, l), scale, lo);
    }
  });
}

void evox_cpu_normal(const int32_t* key, float* out, int64_t n, int64_t offset, float mean, float stddev) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  for_each_block(n, offset, [&](uint64_t blk) {
    f32x4 v = normal4(philox_block(k0, k1, blk, kStreamDraw));
    for (int l = 0; l < 4; ++l) {
      int64_t e = (int64_t)(blk * 4 + l) - offset;
      if (e >= 0 && e < n) out[e] = fmaf(lane_of(v, l), stddev, mean);
    }
  });
}

void evox_cpu_randint(const int32_t* key, int64_t* out, int64_t n, int64_t offset, int64_t lo, int64_t hi) {
  const uint32_t k0 = (uint32_t)key[0], k1 = (uint32_t)key[1];
  const uint32_t range = (uint32_t)(hi - lo);
  for_each_block(n, offset, [&](uint64_t blk) {
    u32x4 b = philox_block(k0, k1, blk, kStreamDraw);
    for (int l = 0; l < 4; ++l) {
      int64_t e = (int64_t)(blk * 4 + l) - offset;
      if (e >= 0 && e < n) out[e] = lo + (int64_t)bits_to_range(lane_of(b, l), range);
    }
  });
}

// nkeys parent keys (nkeys x 2) -> nkeys x n x 2 children
void evox_cpu_split(const int32_t* keys, int32_t* out, int64_t nkeys, int64_t n) {
  for (int64_t q = 0; q < nkeys; ++q) {
    const uint32_t k0 = (uint32_t)keys[2 * q], k1 = (uint32_t)keys[2 * q + 1];
    for (int64_t i = 0; i < n; ++i) {
      u32x4 b = philox_block(k0, k1, (uint64_t)i, kStreamSplit);
      out[(q * n + i) * 2] = (int32_t)b.x;
      out[(q * n + i) * 2 + 1] = (int32_t)b.y;
    }
  }
}

void evox_cpu_fold_in(const int32_t* keys, int32_t* out, int64_t nkeys, int64_t data) {
  for (int64_t q = 0; q < nkeys; ++q) {
    u32x4 b = philox_block((uint32_t)keys[2 * q], (uint32_t)keys[2 * q + 1], (uint64_t)data, kStreamFold);
    out[2 * q] = (int32_t)b.x;
    out[2 * q + 1] = (int32_t)b.y;
  }
}

}  // extern "C"
