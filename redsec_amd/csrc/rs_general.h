// rs_general.h -- the GENERAL ring path: any N = 2^LOGN in {1024, 2048, 4096, 8192}, any gadget (l, Bgbit) with
// l * Bgbit <= 32. Serves the parameter sets the specialised N = 1024 kernels do not cover
// (client/gen_secure_keyset.cpp:9-68: redsec_params_small l=3 Bgbit=10; redsec_params_medium N=4096;
// redsec_params_large N=8192) and, for every set, RS_MODE_FFT_SPLIT: a product that is exact by an A-PRIORI bound.
//
// Arithmetic: the same folded complex FP64 FFT as rs_fft.h (z_j = a_j + i a_{j+M}, M = N/2, twist merged into the
// twiddles, evaluation-tree form), but with the key split into two signed 16-bit halves K = Khi 2^16 + Klo:
//     sum_rows d_row * K_row  =  sum d * Klo  +  2^16 sum d * Khi          (mod 2^32)
// Each half product has coefficients below rows * N * (Bg/2) * 2^15 < 2^40, and the FFT's worst-case error for it is
// far below 1/2 (gen_error_bound below: Percival's bound with the 2-norms of the operands), so rounding to the
// nearest integer returns the exact integer product for EVERY input, not just with overwhelming probability.
// Cost against the unsplit FFT mode: twice the pointwise products and inverse transforms, forward transforms shared.
//
// Work decomposition: one workgroup of T = M/8 threads owns one ciphertext; thread t holds 8 complex values. A
// transform is P = ceil(log2(M)/3) passes of up to three radix-2 levels on those 8 registers, with an LDS exchange
// between passes. Pass p < P-1 works on the layout  idx = (blk << H) | (r << (H-3)) | low,  t = (blk << (H-3)) | low,
// H = log2(M) - 3p  (register r in [0,8) is the three index bits the pass resolves); the last pass works on
// idx = 8 t + r (H = 3) and runs only the log2(M) - 3(P-1) levels that are left. Local level e of a pass is global
// level s = log2(M) - H + e; its block index is idx >> (H - e) and its twiddle table entry 2^s + block.
#pragma once

#include <cmath>
#include <cstdint>

#include "rs_fft.h"

namespace rs {

constexpr int kGenMinLogN = 10, kGenMaxLogN = 13;

template <int LOGN>
struct Gen {
  static constexpr int N = 1 << LOGN, M = N / 2, LOGM = LOGN - 1, T = M / 8;
  static constexpr int P = (LOGM + 2) / 3;            // passes per transform
  static constexpr int K_LAST = LOGM - 3 * (P - 1);   // levels of the last pass, 1..3
  static constexpr int kPlane = M + M / 8;            // doubles per LDS plane, padding included
  static constexpr int H(int p) { return p == P - 1 ? 3 : LOGM - 3 * p; }
  static constexpr int first_level(int p) { return p == P - 1 ? 3 - K_LAST : 0; }
};

// logical index of register r of thread t in the layout of a pass with parameter H
constexpr RS_HD int gen_idx(int t, int H, int r) { return ((t >> (H - 3)) << H) | (r << (H - 3)) | (t & ((1 << (H - 3)) - 1)); }
// LDS position of logical index idx for the exchange whose READER (forward direction) has parameter Hr: blocks of
// 2^Hr values are spaced by 2^(Hr-3) extra slots, which makes the 8-byte accesses of both sides conflict-free
// (rs_fft.h's ppos_t1 / ppos_t2 are the Hr = 6 and Hr = 3 cases).
constexpr RS_HD int gen_phys(int idx, int Hr) { return Hr < 8 ? idx + ((idx >> Hr) << (Hr - 3)) : idx; }

// the (at most four) EVEN twiddles of one pass: level e of the pass uses blocks (blk << e) | g, g < 2^e; odd g is
// i times its even sibling and is applied by the _i butterflies (rs_fft.h)
struct GenPassTw { FftStageTw lv[3]; };
template <int LOGN, int PASS>
RS_HD void gen_pass_tw(GenPassTw& w, int t, const double* tw) {
  using G = Gen<LOGN>;
  constexpr int H = G::H(PASS), E0 = G::first_level(PASS), SB = G::LOGM - H;
  const int blk = t >> (H - 3);
#pragma unroll
  for (int e = E0; e < 3; ++e) {
    const int base = (1 << (SB + e)) + (blk << e);
#pragma unroll
    for (int g = 0; g < (1 << e); g += 2) {
      w.lv[e].wr[g >> 1] = tw[2 * (base + g)];
      w.lv[e].wi[g >> 1] = tw[2 * (base + g) + 1];
    }
  }
}
template <int LOGN, int PASS>
RS_HD void gen_pass_fwd(double (&x)[kRegs], const GenPassTw& w) {
  constexpr int E0 = Gen<LOGN>::first_level(PASS);
  if (E0 <= 0) fft_stage_fwd_tw<0>(x, w.lv[0]);
  if (E0 <= 1) fft_stage_fwd_tw<1>(x, w.lv[1]);
  fft_stage_fwd_tw<2>(x, w.lv[2]);
}
template <int LOGN, int PASS>
RS_HD void gen_pass_inv(double (&x)[kRegs], const GenPassTw& w) {
  constexpr int E0 = Gen<LOGN>::first_level(PASS);
  fft_stage_inv_tw<2>(x, w.lv[2]);
  if (E0 <= 1) fft_stage_inv_tw<1>(x, w.lv[1]);
  if (E0 <= 0) fft_stage_inv_tw<0>(x, w.lv[0]);
}

// registers <-> LDS planes in the layout of pass LAY, through the padding of the exchange between passes XP and XP+1.
// The position of register r is the position of register 0 plus a compile-time constant (gen_reg_offset; checked
// exhaustively by tests/test_emulator.py), so a thread computes one address per plane and the rest are instruction offsets.
template <int LOGN, int LAY, int XP>
constexpr int gen_reg_offset(int r) {
  using G = Gen<LOGN>;
  return gen_phys(gen_idx(0, G::H(LAY), r), G::H(XP + 1)) - gen_phys(gen_idx(0, G::H(LAY), 0), G::H(XP + 1));
}
template <int LOGN, int LAY, int XP>
RS_HD void gen_store(const double (&x)[kRegs], int t, double* pre, double* pim) {
  using G = Gen<LOGN>;
  const int base = gen_phys(gen_idx(t, G::H(LAY), 0), G::H(XP + 1));
  double* qre = pre + base;
  double* qim = pim + base;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    RS_PLANE_STORE(qre, (gen_reg_offset<LOGN, LAY, XP>(r)), x[r]);
    RS_PLANE_STORE(qim, (gen_reg_offset<LOGN, LAY, XP>(r)), x[r + 8]);
  }
}
template <int LOGN, int LAY, int XP>
RS_HD void gen_load(double (&x)[kRegs], int t, const double* pre, const double* pim) {
  using G = Gen<LOGN>;
  const int base = gen_phys(gen_idx(t, G::H(LAY), 0), G::H(XP + 1));
  const double* qre = pre + base;
  const double* qim = pim + base;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    x[r] = RS_PLANE_LOAD(qre, (gen_reg_offset<LOGN, LAY, XP>(r)));
    x[r + 8] = RS_PLANE_LOAD(qim, (gen_reg_offset<LOGN, LAY, XP>(r)));
  }
}

// One exchange = sync (the previous readers of the planes are done), store, sync, load. `sync` is a workgroup
// barrier (a wave-local fence when the workgroup is a single wavefront, N = 1024).
template <int LOGN, int XP, bool INV, class Sync>
RS_HD void gen_exchange(double (&x)[kRegs], int t, double* pre, double* pim, Sync sync) {
  sync();
  gen_store<LOGN, INV ? XP + 1 : XP, XP>(x, t, pre, pim);
  sync();
  gen_load<LOGN, INV ? XP : XP + 1, XP>(x, t, pre, pim);
}

// forward: x[r] + i x[r+8] = folded input value t + T r  ->  transform value 8 t + r (bit-reversed-order tree leaves)
template <int LOGN, class Sync>
RS_HD void gen_fft_fwd(double (&x)[kRegs], int t, const double* tw, double* pre, double* pim, Sync sync) {
  constexpr int P = Gen<LOGN>::P;
  GenPassTw w;
  gen_pass_tw<LOGN, 0>(w, t, tw);
  gen_pass_fwd<LOGN, 0>(x, w);
  if constexpr (P > 1) {
    gen_pass_tw<LOGN, 1>(w, t, tw);
    gen_exchange<LOGN, 0, false>(x, t, pre, pim, sync);
    gen_pass_fwd<LOGN, 1>(x, w);
  }
  if constexpr (P > 2) {
    gen_pass_tw<LOGN, 2>(w, t, tw);
    gen_exchange<LOGN, 1, false>(x, t, pre, pim, sync);
    gen_pass_fwd<LOGN, 2>(x, w);
  }
  if constexpr (P > 3) {
    gen_pass_tw<LOGN, 3>(w, t, tw);
    gen_exchange<LOGN, 2, false>(x, t, pre, pim, sync);
    gen_pass_fwd<LOGN, 3>(x, w);
  }
  static_assert(P <= 4, "at most four passes (N <= 8192)");
}
// inverse (unscaled: 1/M lives in the key): transform value 8 t + r -> folded value t + T r
template <int LOGN, class Sync>
RS_HD void gen_fft_inv(double (&x)[kRegs], int t, const double* tw, double* pre, double* pim, Sync sync) {
  constexpr int P = Gen<LOGN>::P;
  GenPassTw w;
  if constexpr (P > 3) {
    gen_pass_tw<LOGN, 3>(w, t, tw);
    gen_pass_inv<LOGN, 3>(x, w);
    gen_pass_tw<LOGN, 2>(w, t, tw);
    gen_exchange<LOGN, 2, true>(x, t, pre, pim, sync);
  } else if constexpr (P > 2) {
    gen_pass_tw<LOGN, 2>(w, t, tw);
  }
  if constexpr (P > 2) {
    gen_pass_inv<LOGN, 2>(x, w);
    gen_pass_tw<LOGN, 1>(w, t, tw);
    gen_exchange<LOGN, 1, true>(x, t, pre, pim, sync);
  } else if constexpr (P > 1) {
    gen_pass_tw<LOGN, 1>(w, t, tw);
  }
  if constexpr (P > 1) {
    gen_pass_inv<LOGN, 1>(x, w);
    gen_pass_tw<LOGN, 0>(w, t, tw);
    gen_exchange<LOGN, 0, true>(x, t, pre, pim, sync);
  } else {
    gen_pass_tw<LOGN, 0>(w, t, tw);
  }
  gen_pass_inv<LOGN, 0>(x, w);
}

// ---- CMUX pieces for a general ring ----
// modSwitchFromTorus32(a, 2N)
RS_HD int32_t gen_modswitch(int32_t a, int logn) {
  const int sh = 31 - logn;
  return (int32_t)(((uint32_t)a + (1u << (sh - 1))) >> sh);
}
// ((X^a - 1) * acc)_j, 0 < a < 2N
RS_HD int32_t gen_rotated_diff(const int32_t* acc, int j, int a, int logn) {
  const int n1 = (1 << logn) - 1;
  const int aa = a & n1, nb = (a >> logn) & 1;
  const int neg = (j < aa ? 1 : 0) ^ nb;
  const uint32_t v = (uint32_t)acc[(j - aa) & n1];
  return (int32_t)((neg ? (0u - v) : v) - (uint32_t)acc[j]);
}
RS_HD int32_t gen_rotated_const(int32_t mu, int j, int a, int logn) {
  const int aa = a & ((1 << logn) - 1), nb = (a >> logn) & 1;
  return (((j < aa) ? 1 : 0) ^ nb) ? (int32_t)(0u - (uint32_t)mu) : mu;
}
// tGswTorus32PolynomialDecompH with run-time (l, Bgbit): offset added and the top bit of every field flipped once per
// coefficient (rs_ntt.h gadget_prepare), then digit q is the field read as a signed Bgbit-bit number
RS_HD uint32_t gen_gadget_offset(int l, int bgbit) {
  uint32_t off = 0;
  for (int i = 1; i <= l; ++i) off += (1u << (bgbit - 1)) << (32 - i * bgbit);
  return off;
}
RS_HD int32_t gen_gadget_prepare(int32_t d, uint32_t off) { return (int32_t)(((uint32_t)d + off) ^ off); }
RS_HD int32_t gen_gadget_digit(int32_t dx, int q, int bgbit) { return (int32_t)((uint32_t)dx << (q * bgbit)) >> (32 - bgbit); }

// the two signed 16-bit halves of a key coefficient: K = hi * 65536 + lo, lo in [-32768, 32767], hi in [-32768, 32768]
RS_HD void gen_split_key(int32_t k, int32_t& lo, int32_t& hi) {
  lo = (int32_t)(int16_t)(uint16_t)((uint32_t)k & 0xffffu);
  hi = (int32_t)(((int64_t)k - (int64_t)lo) >> 16);
}

// ---- host side ----
// twiddle table: entry 2^s + i (s < log2 M, i < 2^s) = exp(i pi (1 + 4 bitrev_s(i)) / 2^(s+2)), interleaved (re, im);
// odd i stored as EXACTLY i times the even sibling, the value the _i butterflies apply
inline void gen_make_twiddles(int logn, double* tw /* 2 * M doubles */) {
  const int logm = logn - 1;
  const long double pi = 3.141592653589793238462643383279502884L;
  tw[0] = 1.0; tw[1] = 0.0;
  for (int s = 0; s < logm; ++s)
    for (int i = 0; i < (1 << s); ++i) {
      const int ie = i & ~1;
      long br = 0;
      for (int b = 0; b < s; ++b) br |= (long)((ie >> b) & 1) << (s - 1 - b);
      const long double ang = pi * (long double)(1 + 4 * br) / (long double)(1L << (s + 2));
      const double re = (double)cosl(ang), im = (double)sinl(ang);
      const int idx = (1 << s) + i;
      tw[2 * idx] = (i & 1) ? -im : re;
      tw[2 * idx + 1] = (i & 1) ? re : im;
    }
}

// A-priori bound on |computed - true| for one coefficient of  sum_{rows} d_row * Khalf_row  through the FP64 FFT
// (forward transforms of both operands, pointwise products, inverse), for ANY inputs with |d| <= Bg/2 and
// |Khalf| <= 2^15. Percival, "Rapid multiplication modulo the sum and difference of highly composite numbers",
// Math. Comp. 72 (2003), the error theorem for FFT-based convolution (Thm 5.1 there; restated in Brent & Zimmermann,
// Modern Computer Arithmetic, ch. 3) [recalled: no copy in this image]:
//     ||z' - z||_inf <= ||x||_2 ||y||_2 ((1+u)^(3n) (1+u sqrt5)^(3n+1) (1+beta)^(3n) - 1)
// for a length-2^n transform with unit roundoff u = 2^-53 and twiddles within beta of their true values. The merged
// twist makes every level's twiddle general, which is the case the theorem covers; n is taken one higher than
// log2(M) to cover the pre-scaled key and the accumulation over rows, and the result is doubled as slack. Products
// with FMAs round less often than the model assumes. RS_MODE_FFT_SPLIT is offered only when this is below 1/4.
// A self-contained, cruder bound -- Higham, Accuracy and Stability of Numerical Algorithms, Thm 24.2 (relative 2-norm
// error of a radix-2 FFT <= log2(M) eta, eta ~ 7u) carried through the product with |X_k| <= sqrt(M) ||x||_2:
// error <= rows sqrt(M) ||d||_2 ||Khalf||_2 (21 log2(M) + 3) u -- gives 0.006 (default-128), 0.0013 (REDsec shipped set),
// 0.05 (redsec_params_small), 0.49 (medium) and 1.5 (large): it certifies the three N = 1024 sets by itself; for the
// two large rings the claim rests on Percival's sharper analysis (and the largest rounding distance measured on
// operands of the largest norm is 4.3e-6, tests/test_gpu_general.py).
inline double gen_error_bound(int logn, int l, int bgbit) {
  const double N = std::ldexp(1.0, logn);
  const int n = logn;   // log2(M) + 1
  const double u = std::ldexp(1.0, -53), beta = u;
  const double growth = std::pow(1.0 + u, 3.0 * n) * std::pow(1.0 + u * std::sqrt(5.0), 3.0 * n + 1.0) * std::pow(1.0 + beta, 3.0 * n) - 1.0;
  const double norm_d = std::ldexp(1.0, bgbit - 1) * std::sqrt(N);   // |d| <= Bg/2 on N coefficients
  const double norm_k = 32768.0 * std::sqrt(N);
  const double rows = 2.0 * l;
  return 2.0 * rows * norm_d * norm_k * growth;
}

}  // namespace rs
