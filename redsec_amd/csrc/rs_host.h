// rs_host.h -- host-only helpers: per-parameter-set prime, twiddle tables, exactness validation.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "rs_fft.h"
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

// ---- slice exchange of a stage sharded over n contexts (rs_allgather_rows) ----
// Rank d owns the d-th of n balanced contiguous slices of `rows` (sizes differ by at most one, lower ranks first: the split of
// redsec_amd/sharding.py::shard_range and layers.cpp::shard_range). Every destination pulls every other slice exactly once;
// in round k (1 <= k < n) destination d pulls from (d + k) mod n, so the n copies of a round have n different sources and n
// different destinations: no link is asked for two transfers at a time (xGMI is point to point).
struct SliceCopy { int dst, src, round; size_t lo, hi; };
inline void slice_range(size_t total, int d, int D, size_t* lo, size_t* hi) {
  const size_t base = total / (size_t)D, rem = total % (size_t)D;
  *lo = (size_t)d * base + ((size_t)d < rem ? (size_t)d : rem);
  *hi = *lo + base + ((size_t)d < rem ? 1 : 0);
}
inline std::vector<SliceCopy> exchange_schedule(size_t rows, int n) {
  std::vector<SliceCopy> out;
  for (int d = 0; d < n; ++d)
    for (int k = 1; k < n; ++k) {
      const int e = (d + k) % n;
      SliceCopy c{d, e, k, 0, 0};
      slice_range(rows, e, n, &c.lo, &c.hi);
      if (c.hi > c.lo) out.push_back(c);
    }
  return out;
}

// ---- sliced keyswitch of small batches (rs_kernels.hip: keyswitch_tiled_kernel + keyswitch_reduce_kernel) ----
// A tiled keyswitch launch is (ceil(B / 256) ciphertext blocks) x (ceil(W / 32) word blocks) workgroups; a small batch cuts the N
// input coefficients into `slices` (a power of two, at most 64, at least 4 staging groups... coefficients each) until about 1,024
// workgroups exist. Each slice leaves its partial sums in a scratch of slices x W x B words, which the reduce kernel adds up.
// ---- keyswitch with combined digits (keyswitch_tiled_comb_kernel): the index arithmetic, shared with the host test --------
// A coefficient a-bar (the extracted word plus the rounding offset) carries t digits of basebit bits, digit 0 the most
// significant. Lookup group gq covers the digits [gq D, gq D + dl), dl = D except for a shorter last group.
//   ks_comb_index: the group's table row = its dl digits read as one number (digit gq D in the top bits)
//   ks_comb_digit: digit k of that row index, i.e. which base row (gq D + k, digit) the sum contains
#ifndef RS_HD
#ifdef __HIPCC__
#define RS_HD __host__ __device__ inline
#else
#define RS_HD inline
#endif
#endif
RS_HD constexpr uint32_t ks_digit(uint32_t aibar, int j, int basebit) { return (aibar >> (32 - (j + 1) * basebit)) & ((1u << basebit) - 1u); }
RS_HD constexpr uint32_t ks_comb_index(uint32_t aibar, int gq, int D, int dl, int basebit) {
  return (aibar >> (32 - (gq * D + dl) * basebit)) & ((1u << (basebit * dl)) - 1u);
}
RS_HD constexpr int ks_comb_digit(int comb, int k, int dl, int basebit) { return (comb >> (basebit * (dl - 1 - k))) & ((1 << basebit) - 1); }

inline unsigned keyswitch_slices(long B, int W, int N) {
  if (B <= 0) return 1;
  const unsigned gx = (unsigned)((B + 255) / 256), gy = (unsigned)((W + 31) / 32);
  unsigned split = 1;
  constexpr unsigned kWantWorkgroups = 1024;   // workgroups a sliced launch aims for (2048 and 4096 measured: no better)
  while (split < 64 && gx * gy * split < kWantWorkgroups && N / (int)(2 * split) >= 4) split *= 2;
  return split;
}
inline size_t keyswitch_scratch_words_for(long B, int W, int N) {
  const unsigned split = keyswitch_slices(B, W, N);
  return split > 1 ? (size_t)split * (size_t)W * (size_t)B : 0;
}

// How one slice travels from the context that computed it to a context that needs it. Peer access is asked for per device pair
// (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess); where it is refused the slice is staged through pinned host memory by
// this library itself -- source D2H once, every such destination H2D -- instead of relying on what the runtime does then.
enum CopyPath { kPathSameDevice = 0, kPathPeer = 1, kPathHostStaged = 2 };
inline CopyPath choose_copy_path(int dst_device, int src_device, bool peer_access, bool force_staged) {
  if (force_staged) return kPathHostStaged;
  if (dst_device == src_device) return kPathSameDevice;
  return peer_access ? kPathPeer : kPathHostStaged;
}
// The exchange as the literal list of operations rs_allgather_rows issues, in issue order. Every context has one copy stream and
// the events "slice" (recorded on its COMPUTE stream at the start of the call: everything queued there so far -- the kernels that
// wrote its slice and the kernels that still read the block the gathered batch lands in), "staged" (its slice sits in its pinned
// staging buffer) and "copied" (every copy INTO its buffer is done; recorded last on its copy stream).
//   kOpWaitSlice  ctx's copy stream waits for other's "slice"        kOpStageOut    ctx: D2H of its own slice, then record "staged"
//   kOpWaitCopied ctx's copy stream waits for other's "copied"       kOpWaitStaged  ctx's copy stream waits for other's "staged"
//   (of the PREVIOUS call: its staging buffer may still be read)     kOpCopy        ctx pulls rows [lo, hi) of other by `path`
//   kOpRecordCopied  record ctx's "copied"
// Invariants (checked on the CPU by tests/test_sharding_gloo.py through the emulator library):
//   * before its first copy a destination waits for its OWN "slice": its buffer may be a recycled block that kernels queued on
//     its compute stream are still reading (round-3 advisor finding: a fast source could overwrite it);
//   * every copy waits for the source's "slice" (direct paths) or "staged" (host path), and a source stages once per call,
//     behind its own "slice" and behind every context's previous "copied".
enum { kOpWaitSlice = 0, kOpWaitCopied = 1, kOpStageOut = 2, kOpWaitStaged = 3, kOpCopy = 4, kOpRecordCopied = 5 };
struct ExchangeOp { int kind, ctx, other, path; size_t lo, hi; };
// devices[i]: device of context i; peer[d * n + s] != 0: context d's device may read context s's device directly.
inline std::vector<ExchangeOp> exchange_plan(size_t rows, int n, const int* devices, const unsigned char* peer, bool force_staged) {
  std::vector<ExchangeOp> ops;
  const std::vector<SliceCopy> copies = exchange_schedule(rows, n);
  auto path_of = [&](const SliceCopy& c) { return choose_copy_path(devices[c.dst], devices[c.src], peer[c.dst * n + c.src] != 0, force_staged); };
  for (int s = 0; s < n; ++s) {
    bool staged = false;
    size_t lo = 0, hi = 0;
    for (const SliceCopy& c : copies) if (c.src == s && path_of(c) == kPathHostStaged) { staged = true; lo = c.lo; hi = c.hi; }
    if (!staged) continue;
    ops.push_back({kOpWaitSlice, s, s, 0, 0, 0});
    for (int e = 0; e < n; ++e) ops.push_back({kOpWaitCopied, s, e, 0, 0, 0});
    ops.push_back({kOpStageOut, s, s, kPathHostStaged, lo, hi});
  }
  for (int d = 0; d < n; ++d) {
    ops.push_back({kOpWaitSlice, d, d, 0, 0, 0});
    for (const SliceCopy& c : copies) {
      if (c.dst != d) continue;
      const CopyPath path = path_of(c);
      ops.push_back({path == kPathHostStaged ? kOpWaitStaged : kOpWaitSlice, d, c.src, 0, 0, 0});
      ops.push_back({kOpCopy, d, c.src, (int)path, c.lo, c.hi});
    }
    ops.push_back({kOpRecordCopied, d, d, 0, 0, 0});
  }
  return ops;
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
  std::vector<double> tw;   // kTwTotal doubles: forward, inverse (stage-transposed, tw_pos), fused-stage constants
  double ninv;              // N^-1 mod p, centered
};

inline Tables make_tables(const PrimeSpec& ps, int fuse) {
  Tables t;
  t.f.p = (double)ps.p;
  t.f.pinv = 1.0 / (double)ps.p;
  t.tw.assign(kTwTotal, 0.0);
  const uint64_t psi_inv = powmod_u64(ps.psi, ps.p - 2, ps.p);
  for (uint32_t i = 0; i < (uint32_t)kN; ++i) {
    const uint32_t r = bitrev10(i);
    const int pos = i == 0 ? 0 : tw_pos((int)i);   // stage-transposed layout (rs_ntt.h)
    t.tw[pos] = centered(powmod_u64(ps.psi, r, ps.p), ps.p);
    t.tw[kN + pos] = centered(powmod_u64(psi_inv, r, ps.p), ps.p);
  }
  t.ninv = centered(powmod_u64((uint64_t)kN, ps.p - 2, ps.p), ps.p);
  // constants of the fused forward stages 0-1 (rs_ntt.h fwd_F1_digits)
  const uint64_t I = powmod_u64(ps.psi, bitrev10(1), ps.p), w1 = powmod_u64(ps.psi, bitrev10(2), ps.p),
                 w2 = powmod_u64(ps.psi, bitrev10(3), ps.p);
  const uint64_t w1I = mulmod_u64(w1, I, ps.p), w2I = mulmod_u64(w2, I, ps.p);
  double* sm = t.tw.data() + 2 * kN;
  if (fuse == 2) {
    sm[0] = centered(w1I, ps.p);
    sm[1] = centered(w2I, ps.p);
  } else if (fuse == 1) {
    const uint64_t c[5] = {I, w1, w1I, w2, w2I};
    for (int k = 0; k < 5; ++k)
      for (int tt = 0; tt < 128; ++tt) {
        const int64_t dgt = tt - 64;
        const uint64_t du = dgt >= 0 ? (uint64_t)dgt : ps.p - (uint64_t)(-dgt);
        sm[k * 128 + tt] = centered(mulmod_u64(c[k], du, ps.p), ps.p);
      }
  }
  return t;
}

// Complex twiddles of the folded FFT (rs_fft.h): w_{m,i} = exp(2 pi i (1 + 4 bitrev_s(i)) / 2^(s+3)),
// m = 2^s, stored interleaved (re, im) at the stage-transposed position ftw_pos(m + i).
inline std::vector<double> make_fft_tables() {
  std::vector<double> t(kFftTwDoubles, 0.0);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (int s = 0; s < 9; ++s) {
    const int m = 1 << s;
    for (int i = 0; i < m; ++i) {
      // odd block index: theta + pi/2, stored as EXACTLY i times the even sibling (-im, re): the
      // device fetches only even entries and applies the i-variant butterflies (rs_fft.h)
      const int ie = i & ~1;
      int br = 0;
      for (int b = 0; b < s; ++b) br |= ((ie >> b) & 1) << (s - 1 - b);
      const long a = (long)(1 + 4 * br) << (9 - s);            // angle in units of 2 pi / 4096
      const long double ang = two_pi * (long double)a / 4096.0L;
      const int pos = ftw_pos(m + i);
      const double re = (double)cosl(ang), im = (double)sinl(ang);
      t[2 * pos] = (i & 1) ? -im : re;
      t[2 * pos + 1] = (i & 1) ? re : im;
    }
  }
  return t;
}

// Re-derives the magnitude chain of rs_ntt.h ("Exactness") for a schedule; returns "" if every
// intermediate stays below 2^53 and the final lift is unambiguous, else a description.
// Two forward entry points are checked: the generic one (key polynomials, |x| <= 2^31, all ten
// stages) and the fused-digit one (stages 0-1 evaluated exactly, starting bound 1.5 p / 6 p).
inline std::string validate_schedule(double p, int l, int bgbit, unsigned fwd_mask, unsigned inv_mask, int fuse, bool mid_reduce) {
  const double lim = 9007199254740992.0;  // 2^53
  const double unit = p / lim;            // p * 2^-53
  auto V = [&](double c) { return 0.5 + 1.5 * c * unit + 2.0 / p; };  // mulmod output bound / p
  const double red = 0.5 + 2.0 / p;                                    // reduce() output bound / p
  char msg[256];
  const double half_bg = std::ldexp(1.0, bgbit - 1);
  const double true_bound = 2.0 * l * kN * half_bg * 2147483648.0;
  if (true_bound >= 0.5 * p) return "prime too small for the external-product bound";
  auto forward = [&](double a, int first_stage, const char* what, double* out) -> std::string {
    for (int s = first_stage; s < 10; ++s) {
      a = a + V(a);
      if (a * p >= lim) { snprintf(msg, sizeof msg, "%s forward stage %d reaches %.3f p", what, s, a); return msg; }
      if (fwd_mask & (1u << s)) a = red;
    }
    *out = a;
    return "";
  };
  double x_key = 0, x_dig = 0;
  std::string why = forward(2147483648.0 / p, 0, "generic", &x_key);
  if (!why.empty()) return why;
  if (fuse == 1) {
    why = forward(1.5 + half_bg / p, 2, "table-fused", &x_dig);      // |d0| + 3 table entries of <= p/2
  } else if (fuse == 2) {
    if (half_bg * 0.5 * p * 3.0 + half_bg >= lim) return "fused FMA stage inexact";
    why = forward(3.0 * half_bg * 0.5 + half_bg / p, 2, "fma-fused", &x_dig);  // d0 + 3 * (p/2) * Bg/2
  } else {
    x_dig = x_key;
  }
  if (!why.empty()) return why;
  // pointwise: 2 l products per column, key entries reduced; optional reduction after l products
  const double prod = V(x_dig);
  double acc = l * prod;
  if (acc * p >= lim) { snprintf(msg, sizeof msg, "pointwise half-sum reaches %.3f p", acc); return msg; }
  if (mid_reduce) acc = red;
  acc += l * prod;
  if (acc * p >= lim) { snprintf(msg, sizeof msg, "pointwise sum reaches %.3f p", acc); return msg; }
  // inverse
  double a = red;
  for (int s = 0; s < 10; ++s) {
    const double sum = 2.0 * a;
    if (sum * p >= lim) { snprintf(msg, sizeof msg, "inverse stage %d reaches %.3f p", s, sum); return msg; }
    const double pr = V(sum);
    a = sum > pr ? sum : pr;
    if (inv_mask & (1u << s)) a = red;
  }
  return "";
}

}  // namespace rs
