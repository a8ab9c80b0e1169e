// rs_host.h -- host-only helpers: per-parameter-set prime, twiddle tables, exactness validation.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "rs_ntt.h"

namespace rs {

typedef unsigned __int128 u128_t;

struct PrimeSpec {
  uint64_t p;    // prime = 1 (mod 2048)
  uint64_t psi;  // primitive 2048-th root of unity mod p
};

// Smallest primes p = 1 (mod 2048) with p/2 > 1.02 * (k+1) l N (Bg/2) 2^31 (the largest possible
// |coefficient| of sum_p Dec_p(acc) * BK_row over the integers), found offline with sympy:
//   l=3,  Bgbit=7: bound 2^49.585 -> p = 1722626857492481 (2^50.61), generator 7
//   l=10, Bgbit=3: bound 2^47.322 -> p = 358880595312641  (2^48.35), generator 6
inline bool prime_for(int l, int bgbit, PrimeSpec* out) {
  if (l == 3 && bgbit == 7) { *out = {1722626857492481ull, 1197309855254028ull}; return true; }
  if (l == 10 && bgbit == 3) { *out = {358880595312641ull, 57304311327783ull}; return true; }
  return false;
}

inline uint64_t mulmod_u64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)((u128_t)a * b % p); }
inline uint64_t powmod_u64(uint64_t b, uint64_t e, uint64_t p) {
  uint64_t r = 1;
  while (e) { if (e & 1) r = mulmod_u64(r, b, p); b = mulmod_u64(b, b, p); e >>= 1; }
  return r;
}
inline double centered(uint64_t v, uint64_t p) { return v > p / 2 ? -(double)(p - v) : (double)v; }
inline uint32_t bitrev10(uint32_t x) {
  uint32_t r = 0;
  for (int i = 0; i < 10; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

struct Tables {
  Field f;
  std::vector<double> tw;   // [tw_pos(i)] psi^bitrev(i), [1024 + tw_pos(i)] psi^-bitrev(i), centered
  double ninv;              // N^-1 mod p, centered
};

inline Tables make_tables(const PrimeSpec& ps) {
  Tables t;
  t.f.p = (double)ps.p;
  t.f.pinv = 1.0 / (double)ps.p;
  t.tw.resize(2 * kN);
  const uint64_t psi_inv = powmod_u64(ps.psi, ps.p - 2, ps.p);
  for (uint32_t i = 0; i < (uint32_t)kN; ++i) {
    const uint32_t r = bitrev10(i);
    const int pos = i == 0 ? 0 : tw_pos((int)i);   // stage-transposed layout (rs_ntt.h)
    t.tw[pos] = centered(powmod_u64(ps.psi, r, ps.p), ps.p);
    t.tw[kN + pos] = centered(powmod_u64(psi_inv, r, ps.p), ps.p);
  }
  t.ninv = centered(powmod_u64((uint64_t)kN, ps.p - 2, ps.p), ps.p);
  return t;
}

// Re-derives the magnitude chain of rs_ntt.h ("Exactness") for a schedule; returns "" if every
// intermediate stays below 2^53 and the final lift is unambiguous, else a description.
inline std::string validate_schedule(double p, int l, int bgbit, unsigned fwd_mask, unsigned inv_mask) {
  const double lim = 9007199254740992.0;  // 2^53
  const double unit = p / lim;            // p * 2^-53
  auto V = [&](double c) { return 0.5 + 1.5 * c * unit + 2.0 / p; };  // mulmod output bound / p
  const double red = 0.5 + 2.0 / p;                                    // reduce() output bound / p
  char msg[256];
  // exact integer result must lift uniquely
  const double true_bound = 2.0 * l * kN * std::ldexp(1.0, bgbit - 1) * 2147483648.0;
  if (true_bound >= 0.5 * p) return "prime too small for the external-product bound";
  // forward: digits (|d| <= Bg/2) or key words (|x| <= 2^31)
  double a = 2147483648.0 / p;
  for (int s = 0; s < 10; ++s) {
    const double v = V(a);
    a = a + v;
    if (a * p >= lim) { snprintf(msg, sizeof msg, "forward stage %d reaches %.3f p", s, a); return msg; }
    if (fwd_mask & (1u << s)) a = red;
  }
  const double x_bound = a;
  // pointwise: 2 l products per column, key entries reduced
  const double acc = 2.0 * l * V(x_bound);
  if (acc * p >= lim) { snprintf(msg, sizeof msg, "pointwise sum reaches %.3f p", acc); return msg; }
  // inverse
  a = red;
  for (int s = 0; s < 10; ++s) {
    const double sum = 2.0 * a;
    if (sum * p >= lim) { snprintf(msg, sizeof msg, "inverse stage %d reaches %.3f p", s, sum); return msg; }
    const double prod = V(sum);
    a = sum > prod ? sum : prod;
    if (inv_mask & (1u << s)) a = red;
  }
  return "";
}

}  // namespace rs
