// Coin flips of SRA's stochastic ranking when every comparison draws its own coin: a counter-based hash of
// (seed, sweep, position), identical on the host and on the device.
//
//   coin(s, p) = [ 7 bits of byte (p mod 4) of hash(seed, s, p div 4)  <  pc7 ],   pc7 = round(128 · pc)
//
// Four coins per hash; the probability is resolved to 1/128 (the algorithm draws pc itself at random from [0.4, 0.6]).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define EVOX_SR_HD __host__ __device__ __forceinline__
#else
#define EVOX_SR_HD inline
#endif

namespace evox {

EVOX_SR_HD uint32_t sr_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// the hash word that holds the coins of positions 4g .. 4g+3 of sweep s
EVOX_SR_HD uint32_t sr_coin_group(uint32_t seed, uint32_t s, uint32_t g) {
  return sr_hash((seed + s * 0x9E3779B1u) ^ (g * 0x85EBCA77u + 0x165667B1u));
}

// bit 8b+7 of the result = coin of position 4g + b (all four at once, no carries between the bytes: 7-bit values)
EVOX_SR_HD uint32_t sr_coin_mask(uint32_t group, uint32_t pc7) {
  const uint32_t z = (group & 0x7f7f7f7fu) + (0x80u - pc7) * 0x01010101u;     // byte >= pc7  <=>  its bit 7 set
  return ~z & 0x80808080u;
}

EVOX_SR_HD bool sr_coin(uint32_t seed, uint32_t s, uint32_t p, uint32_t pc7) {
  return (sr_coin_mask(sr_coin_group(seed, s, p >> 2), pc7) >> (8u * (p & 3u) + 7u)) & 1u;
}

EVOX_SR_HD uint32_t sr_pc7(float pc) {
  const float v = pc * 128.f + 0.5f;
  return v <= 0.f ? 0u : (v >= 128.f ? 128u : (uint32_t)v);
}

}  // namespace evox
