// Counter-based PRNG shared by the host (g++) and device (hipcc, gfx950) builds.
//
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") maps
// (key: 2x32 bit, counter: 4x32 bit) -> 4x32 random bits.  Every draw in the framework
// is a pure function of (key, element index): element e of a logical tensor lives in
// counter block e>>2, lane e&3.  That makes a draw independent of launch geometry and
// of how a population is sharded over GPUs, and lets a kernel regenerate noise instead
// of storing it.
//
// All float math below is written with explicit fmaf / IEEE add, mul, sqrt so that the
// host and the device produce bit-identical values (no libm calls, no contraction).
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define EVOX_HD __host__ __device__ __forceinline__
#else
#define EVOX_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace evox {

struct u32x4 { uint32_t x, y, z, w; };
struct f32x4 { float x, y, z, w; };

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u;
constexpr uint32_t kPhiloxW1 = 0xBB67AE85u;

// counter word 2 selects an independent stream for the same key
constexpr uint32_t kStreamDraw = 0u;
constexpr uint32_t kStreamSplit = 0x5EED5EEDu;
constexpr uint32_t kStreamFold = 0xF01DF01Du;

EVOX_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

EVOX_HD u32x4 philox4x32_10(uint32_t k0, uint32_t k1, u32x4 c) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = mulhi32(kPhiloxM0, c.x), lo0 = kPhiloxM0 * c.x;
    uint32_t hi1 = mulhi32(kPhiloxM1, c.z), lo1 = kPhiloxM1 * c.z;
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += kPhiloxW0;
    k1 += kPhiloxW1;
  }
  return c;
}

// block `blk` of the draw stream of `key`
EVOX_HD u32x4 philox_block(uint32_t k0, uint32_t k1, uint64_t blk, uint32_t stream) {
  u32x4 c;
  c.x = (uint32_t)blk;
  c.y = (uint32_t)(blk >> 32);
  c.z = stream;
  c.w = 0u;
  return philox4x32_10(k0, k1, c);
}

// [0,1): 24 random mantissa bits, exact in fp32
EVOX_HD float bits_to_unit(uint32_t b) { return (float)(b >> 8) * 5.9604644775390625e-08f; }
// (0,1]
EVOX_HD float bits_to_unit_open0(uint32_t b) {
  return (float)((b >> 8) + 1u) * 5.9604644775390625e-08f;
}

EVOX_HD uint32_t f2u(float f) {
  union { float f; uint32_t u; } v; v.f = f; return v.u;
}
EVOX_HD float u2f(uint32_t u) {
  union { float f; uint32_t u; } v; v.u = u; return v.f;
}

// natural log for x in (0, 1] (normal, non-denormal inputs: x >= 2^-24).
// Cephes-style: x = m * 2^e with m in [sqrt(1/2), sqrt(2)), ln m by a degree-8 polynomial.
EVOX_HD float det_logf(float x) {
  uint32_t ux = f2u(x);
  int e = (int)(ux >> 23) - 126;                       // x = m * 2^e, m in [0.5, 1)
  float m = u2f((ux & 0x007FFFFFu) | 0x3F000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = 7.0376836292e-2f;
  p = fmaf(p, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = p * m * z;
  float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

// sin(2*pi*u), cos(2*pi*u) for u in [0,1); exact quadrant reduction then minimax polynomials
EVOX_HD void det_sincos2pi(float u, float* s, float* c) {
  float a = u * 4.0f;                  // quarter turns, exact
  float kf = floorf(a + 0.5f);         // nearest quarter turn 0..4
  float f = a - kf;                    // [-0.5, 0.5], exact
  int k = (int)kf;
  float y = f * 1.57079632679489661923f;   // angle in [-pi/4, pi/4]
  float z = y * y;
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float sn = fmaf(ps * z, y, y);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float cs = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
  float rs, rc;
  switch (k & 3) {
    case 0: rs = sn; rc = cs; break;
    case 1: rs = cs; rc = -sn; break;
    case 2: rs = -sn; rc = -cs; break;
    default: rs = -cs; rc = sn; break;
  }
  *s = rs; *c = rc;
}

// sqrt for x in [0, ~40] from fused multiply-adds only: the hardware / libm square roots of the
// two sides do not round identically, a fixed fma sequence does.  Bit-trick seed for 1/sqrt(x),
// three Newton steps, then one residual correction of s = x * r (accurate to <= 1 ulp).
EVOX_HD float det_sqrtf(float x) {
  float r = u2f(0x5f3759dfu - (f2u(x) >> 1));
  const float hx = 0.5f * x;
  r = r * fmaf(-hx * r, r, 1.5f);
  r = r * fmaf(-hx * r, r, 1.5f);
  r = r * fmaf(-hx * r, r, 1.5f);
  float s = x * r;
  const float e = fmaf(-s, s, x);
  s = fmaf(e, 0.5f * r, s);
  return s;
}

EVOX_HD f32x4 unit4(u32x4 b) {
  f32x4 r; r.x = bits_to_unit(b.x); r.y = bits_to_unit(b.y);
  r.z = bits_to_unit(b.z); r.w = bits_to_unit(b.w); return r;
}

// four standard normals from one Philox block (two Box-Muller pairs)
EVOX_HD f32x4 normal4(u32x4 b) {
  f32x4 r;
  float r0 = det_sqrtf(-2.0f * det_logf(bits_to_unit_open0(b.x)));
  float r1 = det_sqrtf(-2.0f * det_logf(bits_to_unit_open0(b.z)));
  float s0, c0, s1, c1;
  det_sincos2pi(bits_to_unit(b.y), &s0, &c0);
  det_sincos2pi(bits_to_unit(b.w), &s1, &c1);
  r.x = r0 * c0; r.y = r0 * s0; r.z = r1 * c1; r.w = r1 * s1;
  return r;
}

EVOX_HD float lane_of(const f32x4& v, int l) { return l == 0 ? v.x : (l == 1 ? v.y : (l == 2 ? v.z : v.w)); }
EVOX_HD uint32_t lane_of(const u32x4& v, int l) { return l == 0 ? v.x : (l == 1 ? v.y : (l == 2 ? v.z : v.w)); }

// unbiased-enough integer in [0, range): 32-bit multiply-shift (bias < range / 2^32)
EVOX_HD uint32_t bits_to_range(uint32_t b, uint32_t range) { return mulhi32(b, range); }

}  // namespace evox
