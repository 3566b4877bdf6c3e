// Built-in eps stream of the engine: Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.
// Host-side C++ twin of clonealign_amd/rng.py; both are pinned against the Random123
// known-answer vectors in tests/test_rng.py.  Replaces the TF-internal stream behind
// `qmu$sample(S, seed = get_next_seed())` (R/inference-tflow.R:49-51,269).
#pragma once
#include <cmath>
#include <cstdint>

namespace ca_philox {

inline void round1(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = 0xD2511F53ull * c[0];
  const uint64_t p1 = 0xCD9E8D57ull * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

inline void philox4x32_10(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  for (int r = 0; r < 10; ++r) {
    round1(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// n standard normals (float32) of draw `draw` in stream `seed`
inline void normal_draw(uint64_t seed, uint64_t draw, int64_t n, float* out) {
  const double two_pi = 6.283185307179586476925286766559;
  const int64_t nblk = (n + 3) / 4;
  for (int64_t b = 0; b < nblk; ++b) {
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)draw, (uint32_t)(draw >> 32)};
    uint32_t r[4];
    philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    double u[4];
    for (int i = 0; i < 4; ++i) u[i] = ((double)r[i] + 0.5) * (1.0 / 4294967296.0);
    const double rad0 = std::sqrt(-2.0 * std::log(u[0])), rad1 = std::sqrt(-2.0 * std::log(u[2]));
    const double a0 = two_pi * u[1], a1 = two_pi * u[3];
    const double z[4] = {rad0 * std::cos(a0), rad0 * std::sin(a0), rad1 * std::cos(a1), rad1 * std::sin(a1)};
    for (int i = 0; i < 4; ++i) {
      const int64_t j = 4 * b + i;
      if (j < n) out[j] = (float)z[i];
    }
  }
}

}  // namespace ca_philox
