// rs_ntt.h -- exact negacyclic N=1024 transform for one 64-lane wavefront, FP64 registers.
//
// Why FP64: measured on MI355X (profiles/valu_rates_r01.jsonl) every FP64 VALU op (fma/mul/add/rndne)
// issues at ~4.25 cycles per wave64 per SIMD -- the same as v_mul_lo_u32 / v_mul_hi_u32 /
// v_mad_u64_u32 -- so one modular butterfly over a ~51-bit prime costs 8 FP64 ops (34 cycles) versus
// 74 cycles for a two-prime 32-bit RNS butterfly and 138 cycles for a 64-bit Goldilocks butterfly.
// All values are integers held in doubles; every operation below is exact (see "Exactness").
//
// What it replaces: the double-precision Lagrange FFT inside TFHE's tGswFFTExternMulToTLwe, reached
// from REDsec at /root/reference/lib/BinOps_enc.cpp:185,191 (tfhe_bootstrap_FFT) and the boots*
// gates (BinOps_enc.cpp:49-52,104-113,153-166,205).
//
// Structure (one wavefront = one polynomial, 16 coefficients per lane, three register layouts):
//   layout A: register r of lane L holds index j = L + 64 r      (index bits 9..6 in-lane)
//   layout B: register s of lane L holds j = 64 (L>>2) + 4 s + (L&3)   (bits 5..2 in-lane)
//   layout C: register u of lane L holds j = 16 L + u            (bits 3..0 in-lane)
// Forward (merged-twist Cooley-Tukey, natural -> bit-reversed): stages 0-3 in A, LDS transpose,
// stages 4-7 in B, LDS transpose, stages 8-9 in C. Inverse (Gentleman-Sande) mirrors it C -> B -> A.
// The 1/N factor is folded into the pre-transformed bootstrapping key.
//
// Exactness. Let f.p = p, |w| <= p/2 for every table entry, y an integer double with |y| = c p.
//   mulmod(y, w): h = fl(w y), l = w y - h (exact, FMA), q = rint(fl(h pinv)), r = h - q p (exact:
//   an integer below 2^53), result r + l == w y (mod p) with |result| <= p (0.5 + 1.5 c p 2^-53).
// Sums/differences of integers stay exact below 2^53, so a schedule of full reductions
// (reduce(x) = x - rint(x pinv) p, |result| <= p/2 + 1) is chosen per parameter set such that no
// intermediate exceeds 2^53; rs::validate_schedule() re-derives the chain at context creation.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define RS_HD __host__ __device__ __forceinline__
#else
#define RS_HD inline
#endif

namespace rs {

constexpr int kN = 1024;      // ring degree handled by one wavefront
constexpr int kLanes = 64;
constexpr int kRegs = 16;     // coefficients per lane
constexpr int kBufDoubles = 1152;  // LDS exchange buffer per wavefront (padded)

struct Field {
  double p;     // prime, p = 1 (mod 2048)
  double pinv;  // fl(1/p)
};

RS_HD double f_reduce(double x, const Field& f) {
  double q = __builtin_rint(x * f.pinv);
  return __builtin_fma(-q, f.p, x);
}

// w must satisfy |w| <= p/2 (a reduced table entry / key coefficient).
RS_HD double f_mulmod(double y, double w, const Field& f) {
  double h = w * y;
  double l = __builtin_fma(w, y, -h);
  double q = __builtin_rint(h * f.pinv);
  return __builtin_fma(-q, f.p, h) + l;
}

// Position of table entry idx (= m + i of the textbook psi_rev[m + i] indexing, m = 2^stage) in the
// stage-transposed tables the kernels read. Stages 0-3 are wave-uniform (unchanged). In stages 4-7
// entry m + b*E + e (b = lane>>2, E = m/16) moves to m + e*16 + b, and in stages 8-9 entry
// m + lane*E + e (E = m/64) moves to m + e*64 + lane, so that for a fixed register index e
// consecutive lanes read consecutive doubles: bank-conflict-free ds_read_b64 (the untransposed
// tables cost 24 % of all LDS cycles in conflicts, profiles/r01/pmc).
RS_HD int tw_pos(int idx) {
  int s = 0;
  while ((2 << s) <= idx) ++s;
  const int m = 1 << s, off = idx - m;
  if (s <= 3) return idx;
  if (s <= 7) { const int E = m >> 4; return m + (off % E) * 16 + off / E; }
  const int E = m >> 6;
  return m + (off % E) * 64 + off / E;
}

// LDS positions (in doubles) of coefficient j for the two transposes; the padding makes the
// ds_read_b64 of layout B and the 16-byte reads of layout C bank-conflict free.
RS_HD int pos_t1(int j) { return j + 4 * (j >> 6); }
RS_HD int pos_t2(int j) { return j + 2 * (j >> 4); }

// Reduction schedule: bit s of FWD_MASK = full reduction after forward stage s (0..9);
// bit s of INV_MASK = full reduction after inverse stage s. The inverse always reduces its input
// and its output. FUSE selects how forward stages 0-1 are evaluated for gadget-digit inputs
// (fwd_F1_digits): 1 = LDS product tables (7-bit digits: w*d does not fit 53 bits), 2 = exact FMAs
// (3-bit digits: |w*d| <= 2p < 2^53). MID_REDUCE = reduce the pointwise partial sums once, after
// the first accumulator component, which is what lets the forward outputs stay un-reduced.
template <int L_, int BGBIT_, unsigned FWD_MASK_, unsigned INV_MASK_, int FUSE_, bool MID_REDUCE_>
struct Cfg {
  static constexpr int L = L_;
  static constexpr int BGBIT = BGBIT_;
  static constexpr unsigned FWD_MASK = FWD_MASK_;
  static constexpr unsigned INV_MASK = INV_MASK_;
  static constexpr int FUSE = FUSE_;
  static constexpr bool MID_REDUCE = MID_REDUCE_;
};
// TFHE default-128: l=3, Bgbit=7 -> prime 2^50.61 (headroom 2^53/p = 5.2).
using CfgDefault128 = Cfg<3, 7, (1u << 3) | (1u << 7), (1u << 2) | (1u << 5) | (1u << 8), 1, true>;
// REDsec redsec_params_small_v2: l=10, Bgbit=3 -> prime 2^48.35 (headroom 25).
using CfgRedsecV2 = Cfg<10, 3, 0u, (1u << 4), 2, true>;
// redsec_params_small (client/gen_secure_keyset.cpp:47-68): l=3, Bgbit=10. Only the gadget shape is used (split-key
// workgroup kernel); its products exceed one FP64-carried prime, so there is no NTT schedule for it.
using CfgRedsecSmall = Cfg<3, 10, 0u, 0u, 0, false>;

// Table block handed to the kernels: [0,1024) forward twiddles, [1024,2048) inverse twiddles,
// [2048, 2048+kSmall) constants of the fused stages 0-1:
//   FUSE 2: sm[0] = w1*I, sm[1] = w2*I                      (I = tw[1], w1 = tw[2], w2 = tw[3])
//   FUSE 1: sm[k*128 + t] = c_k * (t - 64) mod p, c = (I, w1, w1*I, w2, w2*I), t = digit + 64
constexpr int kSmall = 5 * 128;
constexpr int kTwTotal = 2 * kN + kSmall;

// ---------------------------------------------------------------------------------------------
// Forward transform phases. tw = psi^bitrev(i) table (1024 centered doubles), buf = exchange buffer.
// ---------------------------------------------------------------------------------------------
template <class C>
RS_HD void fwd_stage_regs(double (&x)[kRegs], int s, int half, const double* tw, int tw_base, int shift, int stride, const Field& f) {
  // pairs (e, e+half), twiddle at tw_base + (e >> shift) * stride (stage-transposed table, see tw_pos)
#pragma unroll
  for (int e = 0; e < kRegs; ++e) {
    if (e & half) continue;
    const double w = tw[tw_base + (e >> shift) * stride];
    const double v = f_mulmod(x[e + half], w, f);
    const double u = x[e];
    x[e] = u + v;
    x[e + half] = u - v;
  }
  if (C::FWD_MASK & (1u << s)) {
#pragma unroll
    for (int e = 0; e < kRegs; ++e) x[e] = f_reduce(x[e], f);
  }
}

// A wavefront's LDS reads return in order: a twiddle read issued right before its butterflies is waited
// for with everything older still in flight drained, and at one or two waves per SIMD its own latency is
// exposed every time (the exact-NTT blind rotation had ~40 such waits per transform). The phases below
// therefore fetch the twiddles of stage s+1 BEFORE the butterflies of stage s (RS_LDS_FENCE pins the
// reads there; it is a compiler-only fence). Same operations on the same values: bit-identical.
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define RS_LDS_FENCE() ((void)0)
#endif
// the 16 >> SHIFT twiddles of a stage whose pairs are (e, e + half), half = 1 << (SHIFT - 1)
template <int SHIFT>
RS_HD void stage_tw_load(double (&w)[kRegs >> SHIFT], const double* tw, int tw_base, int stride) {
#pragma unroll
  for (int k = 0; k < (kRegs >> SHIFT); ++k) w[k] = tw[tw_base + k * stride];
}
template <class C, int SHIFT>
RS_HD void fwd_stage_tw(double (&x)[kRegs], int s, const double (&w)[kRegs >> SHIFT], const Field& f) {
  constexpr int half = 1 << (SHIFT - 1);
#pragma unroll
  for (int e = 0; e < kRegs; ++e) {
    if (e & half) continue;
    const double v = f_mulmod(x[e + half], w[e >> SHIFT], f);
    const double u = x[e];
    x[e] = u + v;
    x[e + half] = u - v;
  }
  if (C::FWD_MASK & (1u << s)) {
#pragma unroll
    for (int e = 0; e < kRegs; ++e) x[e] = f_reduce(x[e], f);
  }
}
template <class C, int SHIFT>
RS_HD void inv_stage_tw(double (&x)[kRegs], int s, const double (&w)[kRegs >> SHIFT], const Field& f) {
  constexpr int half = 1 << (SHIFT - 1);
#pragma unroll
  for (int e = 0; e < kRegs; ++e) {
    if (e & half) continue;
    const double u = x[e], v = x[e + half];
    x[e] = u + v;
    x[e + half] = f_mulmod(u - v, w[e >> SHIFT], f);
  }
  if (C::INV_MASK & (1u << s)) {
#pragma unroll
    for (int e = 0; e < kRegs; ++e) x[e] = f_reduce(x[e], f);
  }
}
// four consecutive stages with shifts 4, 3, 2, 1 (halves 8, 4, 2, 1): forward order / inverse order
template <class C>
RS_HD void fwd_four_stages(double (&x)[kRegs], int s0, const double* tw, int base0, int base1, int base2, int base3, int stride, const Field& f) {
  double w0[1], w1[2], w2[4], w3[8];
  stage_tw_load<4>(w0, tw, base0, stride); stage_tw_load<3>(w1, tw, base1, stride); RS_LDS_FENCE();
  fwd_stage_tw<C, 4>(x, s0, w0, f);
  stage_tw_load<2>(w2, tw, base2, stride); RS_LDS_FENCE();
  fwd_stage_tw<C, 3>(x, s0 + 1, w1, f);
  stage_tw_load<1>(w3, tw, base3, stride); RS_LDS_FENCE();
  fwd_stage_tw<C, 2>(x, s0 + 2, w2, f);
  fwd_stage_tw<C, 1>(x, s0 + 3, w3, f);
}
template <class C>
RS_HD void inv_four_stages(double (&x)[kRegs], int s0, const double* twi, int base0, int base1, int base2, int base3, int stride, const Field& f) {
  // halves 1, 2, 4, 8 = shifts 1, 2, 3, 4
  double w0[8], w1[4], w2[2], w3[1];
  stage_tw_load<1>(w0, twi, base0, stride); stage_tw_load<2>(w1, twi, base1, stride); RS_LDS_FENCE();
  inv_stage_tw<C, 1>(x, s0, w0, f);
  stage_tw_load<3>(w2, twi, base2, stride); stage_tw_load<4>(w3, twi, base3, stride); RS_LDS_FENCE();
  inv_stage_tw<C, 2>(x, s0 + 1, w1, f);
  inv_stage_tw<C, 3>(x, s0 + 2, w2, f);
  inv_stage_tw<C, 4>(x, s0 + 3, w3, f);
}

// F1: stages 0..3 on layout A, then store for transpose 1.
template <class C>
RS_HD void fwd_F1(int lane, double (&x)[kRegs], const double* tw, double* buf, const Field& f) {
  fwd_four_stages<C>(x, 0, tw, 1, 2, 4, 8, 1, f);
#pragma unroll
  for (int r = 0; r < kRegs; ++r) buf[lane + 68 * r] = x[r];
}
// gadget digit q (0-based) of coefficient d, biased by Bg/2: ((d + offset) >> decal) & (Bg-1)
// (tGswTorus32PolynomialDecompH; the signed digit is this minus Bg/2).
template <class C>
RS_HD int32_t gadget_digit_biased(int32_t d, int q, uint32_t offset) {
  const uint32_t u = (uint32_t)d + offset;
  const int decal = 32 - (q + 1) * C::BGBIT;
  return (int32_t)((u >> decal) & ((1u << C::BGBIT) - 1u));
}

// F1 for gadget digits: stages 0-1 fused over the register quadruples (g, g+4, g+8, g+12):
//   out[g]    = a + c,  out[g+4]  = a - c,   a = d0 + I d2,  c = w1 d1 + (w1 I) d3
//   out[g+8]  = b + e,  out[g+12] = b - e,   b = d0 - I d2,  e = w2 d1 - (w2 I) d3
// then stages 2-3 and the transpose-1 store. Bounds after stage 1: 1.5 p (tables) / 6 p (FMA).
template <class C>
RS_HD void fwd_F1_digits(int lane, double (&x)[kRegs], const int32_t (&d)[kRegs], int q, uint32_t offset, const double* tw,
                         double* buf, const Field& f) {
  const double* sm = tw + 2 * kN;
  constexpr int HALF = 1 << (C::BGBIT - 1);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int32_t t0 = gadget_digit_biased<C>(d[g], q, offset), t1 = gadget_digit_biased<C>(d[g + 4], q, offset);
    const int32_t t2 = gadget_digit_biased<C>(d[g + 8], q, offset), t3 = gadget_digit_biased<C>(d[g + 12], q, offset);
    double a, b, c, e;
    if (C::FUSE == 1) {
      const double d0 = (double)(t0 - HALF);
      const double id2 = sm[0 * 128 + t2];
      a = d0 + id2;
      b = d0 - id2;
      c = sm[1 * 128 + t1] + sm[2 * 128 + t3];
      e = sm[3 * 128 + t1] - sm[4 * 128 + t3];
    } else {
      const double d0 = (double)(t0 - HALF), d1 = (double)(t1 - HALF), d2 = (double)(t2 - HALF), d3 = (double)(t3 - HALF);
      const double I = tw[1], w1 = tw[2], w2 = tw[3], w1I = sm[0], w2I = sm[1];
      a = __builtin_fma(I, d2, d0);
      b = __builtin_fma(-I, d2, d0);
      c = __builtin_fma(w1I, d3, w1 * d1);
      e = __builtin_fma(-w2I, d3, w2 * d1);
    }
    x[g] = a + c;
    x[g + 4] = a - c;
    x[g + 8] = b + e;
    x[g + 12] = b - e;
  }
  {
    double w2[4], w3[8];
    stage_tw_load<2>(w2, tw, 4, 1); stage_tw_load<1>(w3, tw, 8, 1); RS_LDS_FENCE();
    fwd_stage_tw<C, 2>(x, 2, w2, f);
    fwd_stage_tw<C, 1>(x, 3, w3, f);
  }
#pragma unroll
  for (int r = 0; r < kRegs; ++r) buf[lane + 68 * r] = x[r];
}
// F2: load layout B, stages 4..7.
template <class C>
RS_HD void fwd_F2(int lane, double (&x)[kRegs], const double* tw, const double* buf, const Field& f) {
  const int b = lane >> 2, q = lane & 3;
#pragma unroll
  for (int s = 0; s < kRegs; ++s) x[s] = buf[68 * b + 4 * s + q];
  fwd_four_stages<C>(x, 4, tw, 16 + b, 32 + b, 64 + b, 128 + b, 16, f);
}
// F3: store for transpose 2.
RS_HD void fwd_F3(int lane, const double (&x)[kRegs], double* buf) {
  const int b = lane >> 2, q = lane & 3;
#pragma unroll
  for (int s = 0; s < kRegs; ++s) buf[72 * b + 4 * s + q + 2 * (s >> 2)] = x[s];
}
// F4: load layout C, stages 8..9. Output: x[u] = transform value at position 16*lane + u.
template <class C>
RS_HD void fwd_F4(int lane, double (&x)[kRegs], const double* tw, const double* buf, const Field& f) {
#pragma unroll
  for (int u = 0; u < kRegs; ++u) x[u] = buf[18 * lane + u];
  {
    double w8[4], w9[8];
    stage_tw_load<2>(w8, tw, 256 + lane, 64); stage_tw_load<1>(w9, tw, 512 + lane, 64); RS_LDS_FENCE();
    fwd_stage_tw<C, 2>(x, 8, w8, f);
    fwd_stage_tw<C, 1>(x, 9, w9, f);
  }
}

// ---------------------------------------------------------------------------------------------
// Inverse transform phases. twi = psi^-bitrev(i) table.
// ---------------------------------------------------------------------------------------------
template <class C>
RS_HD void inv_stage_regs(double (&x)[kRegs], int s, int half, const double* twi, int tw_base, int shift, int stride, const Field& f) {
#pragma unroll
  for (int e = 0; e < kRegs; ++e) {
    if (e & half) continue;
    const double w = twi[tw_base + (e >> shift) * stride];
    const double u = x[e], v = x[e + half];
    x[e] = u + v;
    x[e + half] = f_mulmod(u - v, w, f);
  }
  if (C::INV_MASK & (1u << s)) {
#pragma unroll
    for (int e = 0; e < kRegs; ++e) x[e] = f_reduce(x[e], f);
  }
}

// I1: reduce the pointwise sums, stages 0..1 on layout C, store (transpose 2 positions).
template <class C>
RS_HD void inv_I1(int lane, double (&x)[kRegs], const double* twi, double* buf, const Field& f) {
#pragma unroll
  for (int u = 0; u < kRegs; ++u) x[u] = f_reduce(x[u], f);
  {
    double w0[8], w1[4];
    stage_tw_load<1>(w0, twi, 512 + lane, 64); stage_tw_load<2>(w1, twi, 256 + lane, 64); RS_LDS_FENCE();
    inv_stage_tw<C, 1>(x, 0, w0, f);
    inv_stage_tw<C, 2>(x, 1, w1, f);
  }
#pragma unroll
  for (int u = 0; u < kRegs; ++u) buf[18 * lane + u] = x[u];
}
// I2: load layout B, stages 2..5.
template <class C>
RS_HD void inv_I2(int lane, double (&x)[kRegs], const double* twi, const double* buf, const Field& f) {
  const int b = lane >> 2, q = lane & 3;
#pragma unroll
  for (int s = 0; s < kRegs; ++s) x[s] = buf[72 * b + 4 * s + q + 2 * (s >> 2)];
  inv_four_stages<C>(x, 2, twi, 128 + b, 64 + b, 32 + b, 16 + b, 16, f);
}
// I3: store (transpose 1 positions).
RS_HD void inv_I3(int lane, const double (&x)[kRegs], double* buf) {
  const int b = lane >> 2, q = lane & 3;
#pragma unroll
  for (int s = 0; s < kRegs; ++s) buf[68 * b + 4 * s + q] = x[s];
}
// I4: load layout A, stages 6..9, final reduction to the centered residue.
template <class C>
RS_HD void inv_I4(int lane, double (&x)[kRegs], const double* twi, const double* buf, const Field& f) {
#pragma unroll
  for (int r = 0; r < kRegs; ++r) x[r] = buf[lane + 68 * r];
  inv_four_stages<C>(x, 6, twi, 8, 4, 2, 1, 1, f);
#pragma unroll
  for (int r = 0; r < kRegs; ++r) x[r] = f_reduce(x[r], f);
}

// Integer-valued double with |v| < 2^51 -> v mod 2^32 (two's complement), via the 1.5*2^52 shift.
RS_HD int32_t f_to_torus32(double v) {
  const double t = v + 6755399441055744.0;
  long long bits;
  __builtin_memcpy(&bits, &t, sizeof(bits));
  return (int32_t)(uint32_t)(unsigned long long)bits;
}

// ---------------------------------------------------------------------------------------------
// CMUX pieces shared by the kernel and the host emulator.
// ---------------------------------------------------------------------------------------------
// modSwitchFromTorus32(a, 2N) for N = 1024: ((a << 32) + 2^52) >> 53 with 64-bit wrap-around.
RS_HD int32_t modswitch_2N(int32_t a) { return (int32_t)(((uint32_t)a + (1u << 20)) >> 21); }

// d_j = ((X^a - 1) * acc)_j for 0 < a < 2N (torusPolynomialMulByXaiMinusOne).
RS_HD int32_t rotated_diff(const int32_t* acc, int j, int a) {
  const int aa = a & (kN - 1), nb = (a >> 10) & 1;
  const int idx = (j - aa) & (kN - 1);
  const int neg = (j < aa ? 1 : 0) ^ nb;
  const uint32_t v = (uint32_t)acc[idx];
  return (int32_t)((neg ? (0u - v) : v) - (uint32_t)acc[j]);
}
// j-th coefficient of X^a * (mu, mu, ..., mu) for 0 < a <= 2N (torusPolynomialMulByXai).
RS_HD int32_t rotated_const(int32_t mu, int j, int a) {
  const int aa = a & (kN - 1), nb = (a >> 10) & 1;
  const int neg = (j < aa ? 1 : 0) ^ nb;
  return neg ? (int32_t)(0u - (uint32_t)mu) : mu;
}
// Word k of a SYNTHETIC key (rs_load_synthetic_keys: benchmarks and tests of the large rings, whose real keys are gigabytes):
// the high half of splitmix64(seed + k), the same on the device, in the C++ host code and in numpy (redsec_amd/client.py).
RS_HD uint32_t synthetic_key_word(uint64_t seed, uint64_t k) {
  uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

// gadget digit q (0-based) of coefficient d (tGswTorus32PolynomialDecompH).
template <class C>
RS_HD int32_t gadget_digit(int32_t d, int q, uint32_t offset) {
  const uint32_t u = (uint32_t)d + offset;
  const int decal = 32 - (q + 1) * C::BGBIT;
  return (int32_t)((u >> decal) & ((1u << C::BGBIT) - 1u)) - (1 << (C::BGBIT - 1));
}
template <class C>
constexpr uint32_t gadget_offset() {
  uint32_t off = 0;
  for (int i = 1; i <= C::L; ++i) off += (1u << (C::BGBIT - 1)) << (32 - i * C::BGBIT);
  return off;
}

// The same digits with one operation less per digit: flipping the top bit of every Bgbit-wide field of
// u = d + offset (the bit pattern of that mask IS the offset) turns "field - Bg/2" into the field read
// as a SIGNED Bgbit-bit number, i.e. one v_bfe_i32 (or one arithmetic shift for q = 0):
//   signed(field ^ Bg/2) = ((field + Bg/2 + Bg/2) mod Bg) - Bg/2 = field - Bg/2.
// gadget_prepare runs once per coefficient, gadget_digit_prepared once per (coefficient, q).
template <class C>
RS_HD int32_t gadget_prepare(int32_t d) {
  constexpr uint32_t off = gadget_offset<C>();
  return (int32_t)(((uint32_t)d + off) ^ off);
}
template <class C>
RS_HD int32_t gadget_digit_prepared(int32_t dx, int q) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sbfe(dx, (unsigned)(32 - (q + 1) * C::BGBIT), (unsigned)C::BGBIT);   // v_bfe_i32, also for a run-time q
#else
  return (int32_t)((uint32_t)dx << (q * C::BGBIT)) >> (32 - C::BGBIT);
#endif
}

}  // namespace rs
