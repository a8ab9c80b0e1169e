// rs_fft.h -- negacyclic N=1024 product via a folded 512-point complex FP64 FFT, one wavefront per
// polynomial. This is the arithmetic CLASS TFHE itself uses (tGswFFTExternMulToTLwe: Lagrange
// half-complex FFT in double precision), restated with this backend's own butterfly order.
//
// Folding: for real a(X) of degree < N let z_j = a_j + i a_{j+N/2} (j < M = N/2) and zeta = exp(i pi / N).
// Then a(zeta^(4k+1)) = sum_j z_j zeta^j omega^(jk), omega = zeta^4 = exp(2 pi i / M): an M-point DFT of
// the twisted sequence; the other N/2 evaluation points are complex conjugates, so M complex values
// represent the polynomial and products are pointwise. With registers x[r] = coefficient L + 64 r
// (r < 16) the complex value j = L + 64 r (r < 8) is (x[r], x[r+8]): the folding costs nothing.
//
// The twist is merged into the twiddles (evaluation-tree form): block i of stage s (m = 2^s blocks)
// handles the roots of X^(M/m) = c_{m,i}, c_{1,0} = zeta^M = i; its twiddle is sqrt(c) =
// exp(i theta_{m,i} / 2) with theta_{1,0} = pi/2, theta_{2m,2i} = theta_{m,i}/2,
// theta_{2m,2i+1} = theta_{m,i}/2 + pi. Forward: Cooley-Tukey butterflies (x + w y, x - w y), natural ->
// bit-reversed; inverse: Gentleman-Sande with conj(w). 1/M is folded into the key (a power of two).
//
// Exactness: the true product coefficients are integers below 2^50; the FP64 FFT returns them with an
// error of standard deviation ~0.02 (default-128) or less, so rounding to the nearest integer
// reproduces the exact product (and hence the exact-NTT path and the CPU oracle) bit for bit with
// overwhelming probability. It is not a worst-case guarantee: the kernels track the largest distance
// to the nearest integer they ever round (the "certificate": values << 0.5 mean errors of +-1 are
// many tens of sigma away) and the exact NTT path remains available (RS_MODE_EXACT_NTT).
//
// Layouts (8 complex per lane): A' j = L + 64 r; B' j = 64 (L>>3) + 8 s + (L&7); C' j = 8 L + u.
// Stages 0-2 in A', LDS transpose, 3-5 in B', LDS transpose, 6-8 in C' (inverse mirrored).
#pragma once

#include <cstdint>

#include "rs_diag.h"
#include "rs_ntt.h"

namespace rs {

constexpr int kM = kN / 2;           // complex points
constexpr int kCRegs = 8;            // complex values per lane
constexpr int kFftTwDoubles = 2 * kM;  // complex twiddle table (interleaved re, im), stage-transposed

struct Cplx { double re, im; };

// Position (in complex units) of table entry m + i in the stage-transposed table (cf. tw_pos).
RS_HD int ftw_pos(int idx) {
  int s = 0;
  while ((2 << s) <= idx) ++s;
  const int m = 1 << s, off = idx - m;
  if (s <= 2) return idx;
  if (s <= 5) { const int E = m >> 3; return m + (off % E) * 8 + off / E; }     // stages 3-5: b = lane>>3 in [0,8)
  const int E = m >> 6;
  return m + (off % E) * 64 + off / E;                                            // stages 6-8: lane in [0,64)
}

// exchange-buffer positions (complex units); paddings give conflict-free 16-byte accesses
RS_HD int fpos_t1(int j) { return j + 8 * (j >> 7); }
RS_HD int fpos_t2(int j) { return j + (j >> 3); }

RS_HD void fft_bfly_fwd(double& xr, double& xi, double& yr, double& yi, double wr, double wi) {
  // 6 FP64 ops: x' = x + w*y as two chained FMAs per component, y' = 2x - x' as one more.
  const double sr = __builtin_fma(-wi, yi, __builtin_fma(wr, yr, xr));
  const double si = __builtin_fma(wi, yr, __builtin_fma(wr, yi, xi));
  yr = __builtin_fma(2.0, xr, -sr); yi = __builtin_fma(2.0, xi, -si);
  xr = sr; xi = si;
}
// The same butterfly with the twiddle i*w = (-wi, wr) given as w: bit-identical to fft_bfly_fwd(.., -wi, wr).
RS_HD void fft_bfly_fwd_i(double& xr, double& xi, double& yr, double& yi, double wr, double wi) {
  const double sr = __builtin_fma(-wr, yi, __builtin_fma(-wi, yr, xr));
  const double si = __builtin_fma(wr, yr, __builtin_fma(-wi, yi, xi));
  yr = __builtin_fma(2.0, xr, -sr); yi = __builtin_fma(2.0, xi, -si);
  xr = sr; xi = si;
}
RS_HD void fft_bfly_inv(double& xr, double& xi, double& yr, double& yi, double wr, double wi) {
  const double dr = xr - yr, di = xi - yi;
  xr = xr + yr; xi = xi + yi;
  yr = __builtin_fma(wi, di, wr * dr);      // conj(w) * d
  yi = __builtin_fma(-wi, dr, wr * di);
}
// twiddle i*w given as w: conj(i w) * d, bit-identical to fft_bfly_inv(.., -wi, wr)
RS_HD void fft_bfly_inv_i(double& xr, double& xi, double& yr, double& yi, double wr, double wi) {
  const double dr = xr - yr, di = xi - yi;
  xr = xr + yr; xi = xi + yi;
  yr = __builtin_fma(wr, di, -(wi * dr));
  yi = __builtin_fma(-wr, dr, -(wi * di));
}

// The twiddles depend only on (stage, lane, register index), never on the polynomial, so a wavefront
// loads them ONCE: 7 wave-uniform ones for stages 0-2 and 14 per-lane ones for stages 3-8
// (index of entry (stage, e'): 3 -> 0 ; 4 -> 1 + e' ; 5 -> 3 + e' ; 6 -> 7 ; 7 -> 8 + e' ; 8 -> 10 + e').
// The inverse transform uses the conjugates of the same values.
struct FftTw {
  double ur[7], ui[7];     // table entries 1..7 (stages 0, 1, 2)
  double lr[14], li[14];   // per-lane entries of stages 3..8
};

RS_HD void fft_tw_load(FftTw& t, int lane, const double* tw) {
#pragma unroll
  for (int k = 0; k < 7; ++k) { t.ur[k] = tw[2 * (k + 1)]; t.ui[k] = tw[2 * (k + 1) + 1]; }
  const int b = lane >> 3;
  int k = 0;
#pragma unroll
  for (int s = 3; s < 9; ++s) {
    const int per = 1 << ((s - 3) % 3);           // 1, 2, 4 entries per lane
#pragma unroll
    for (int e = 0; e < per; ++e) {
      const int pos = (s < 6) ? (1 << s) + e * 8 + b : (1 << s) + e * 64 + lane;   // stage-transposed table (ftw_pos)
      t.lr[k] = tw[2 * pos];
      t.li[k] = tw[2 * pos + 1];
      ++k;
    }
  }
}

template <int S>
RS_HD void fft_tw_get(const FftTw& t, int k, double& wr, double& wi) {
  constexpr int lbase = (S < 3) ? 0 : (S - 3 < 3 ? (S == 3 ? 0 : (S == 4 ? 1 : 3)) : (S == 6 ? 7 : (S == 7 ? 8 : 10)));
  wr = (S < 3) ? t.ur[(1 << S) - 1 + k] : t.lr[lbase + k];
  wi = (S < 3) ? t.ui[(1 << S) - 1 + k] : t.li[lbase + k];
}

// Stages 0-2 use wave-uniform twiddles (table entries 1, 2, 4, 6; the odd entries are i times their
// even sibling): literals, so they live in scalar registers instead of costing LDS reads. The values
// are those make_fft_tables() computes (tests/test_emulator.py checks the two against each other).
constexpr double kFftTwU[8] = {
    0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1,   // entry 1: exp(i * 2 pi * 512 / 4096)
    0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2,   // entry 2: exp(i * 2 pi * 256 / 4096)
    0x1.f6297cff75cbp-1, 0x1.8f8b83c69a60bp-3,    // entry 4: exp(i * 2 pi * 128 / 4096)
    0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1,   // entry 6: exp(i * 2 pi * 640 / 4096)
};
// literal of stage S < 3, EVEN twiddle index k (0 or 2)
template <int S>
RS_HD void fft_tw_uniform(int k, double& wr, double& wi) {
  const int e = (S == 0) ? 0 : (S == 1 ? 1 : 2 + (k >> 1));
  wr = kFftTwU[2 * e];
  wi = kFftTwU[2 * e + 1];
}

// Twiddle source reading the stage-transposed table (in LDS on the device) at every use: costs 21
// 16-byte LDS reads per transform but no registers. Same values, hence bit-identical results.
struct FftTwTable {
  const double* tw;
  int lane;
};
template <int S>
RS_HD void fft_tw_get(const FftTwTable& t, int k, double& wr, double& wi) {
  const int pos = (S < 3) ? (1 << S) + k : (S < 6 ? (1 << S) + k * 8 + (t.lane >> 3) : (1 << S) + k * 64 + t.lane);
  wr = t.tw[2 * pos];
  wi = t.tw[2 * pos + 1];
}

// Twiddle source that KEEPS the even per-lane entries of stages FIRST..8 in registers and reads the rest from the table: the
// lock-step workgroup kernel's pairs pass through the same entries again and again, and what an LDS read costs there is
// issue time of a SIMD whose LDS pipe is the busier unit. FIRST = 3: 8 complex values (32 registers),
// FIRST = 6: the 4 of the last group (16 registers).
template <int FIRST>
struct FftTwKept {
  FftTwTable tab;
  double wr[9 - FIRST][2], wi[9 - FIRST][2];
};
template <int S, int FIRST>
RS_HD void fft_tw_get(const FftTwKept<FIRST>& t, int k, double& wr, double& wi) {
  if (S >= FIRST) { wr = t.wr[S >= FIRST ? S - FIRST : 0][k >> 1]; wi = t.wi[S >= FIRST ? S - FIRST : 0][k >> 1]; }
  else fft_tw_get<S>(t.tab, k, wr, wi);
}
template <int S, int FIRST>
RS_HD void fft_kept_load_stage(FftTwKept<FIRST>& t) {
  if (S >= FIRST) {
    constexpr int ntw = kCRegs >> (3 - S % 3);
#pragma unroll
    for (int k = 0; k < ntw; k += 2) fft_tw_get<S>(t.tab, k, t.wr[S >= FIRST ? S - FIRST : 0][k >> 1], t.wi[S >= FIRST ? S - FIRST : 0][k >> 1]);
  }
}
template <int FIRST>
RS_HD void fft_kept_load(FftTwKept<FIRST>& t, const FftTwTable& tab) {
  t.tab = tab;
  fft_kept_load_stage<3>(t); fft_kept_load_stage<4>(t); fft_kept_load_stage<5>(t);
  fft_kept_load_stage<6>(t); fft_kept_load_stage<7>(t); fft_kept_load_stage<8>(t);
}

// x[e] = re, x[e+8] = im of the lane's e-th complex value. Stage s in [0,9): pairs (e, e+half),
// half = 4 >> (s % 3); twiddle index within the stage = e >> (3 - s % 3). Only the EVEN twiddle
// indices are fetched: index k+1 is i times index k (theta + pi/2), applied by the _i butterflies.
template <int S, class TW>
RS_HD void fft_tw_even(const TW& t, int k, double& wr, double& wi) {
  if (S < 3) fft_tw_uniform<S>(k, wr, wi); else fft_tw_get<S>(t, k, wr, wi);
}
// The (at most two) EVEN twiddles one lane needs for stage S, fetched ahead of their use. A wavefront's
// LDS operations return in order: a twiddle read issued AFTER an exchange's burst of plane loads/stores
// can only be waited for by draining that whole burst (s_waitcnt lgkmcnt(0) in front of every butterfly
// stage), whereas one issued BEFORE the burst is waited for with a counted lgkmcnt(N) and the burst
// stays in flight behind the butterflies.
struct FftStageTw { double wr[2], wi[2]; };
template <int S, class TW>
RS_HD void fft_stage_tw(const TW& t, FftStageTw& w) {
  constexpr int g = S % 3, shift = 3 - g, ntw = kCRegs >> shift;
#pragma unroll
  for (int k = 0; k < ntw; k += 2) fft_tw_even<S>(t, k, w.wr[k >> 1], w.wi[k >> 1]);
}
template <int S>
RS_HD void fft_stage_fwd_tw(double (&x)[kRegs], const FftStageTw& w) {
  constexpr int g = S % 3, half = 4 >> g, shift = 3 - g, ntw = kCRegs >> shift;
#pragma unroll
  for (int k = 0; k < ntw; k += 2) {
    const double wr = w.wr[k >> 1], wi = w.wi[k >> 1];
    const int e = k << shift;                              // elements [e, e + 2 half) use twiddle k
#pragma unroll
    for (int c = 0; c < half; ++c) fft_bfly_fwd(x[e + c], x[e + c + 8], x[e + c + half], x[e + c + half + 8], wr, wi);
    if (g != 0) {
      const int o = e + 2 * half;                          // ... and [o, o + 2 half) its sibling k + 1
#pragma unroll
      for (int c = 0; c < half; ++c) fft_bfly_fwd_i(x[o + c], x[o + c + 8], x[o + c + half], x[o + c + half + 8], wr, wi);
    }
  }
}
template <int S>
RS_HD void fft_stage_inv_tw(double (&x)[kRegs], const FftStageTw& w) {
  constexpr int g = S % 3, half = 4 >> g, shift = 3 - g, ntw = kCRegs >> shift;
#pragma unroll
  for (int k = 0; k < ntw; k += 2) {
    const double wr = w.wr[k >> 1], wi = w.wi[k >> 1];
    const int e = k << shift;
#pragma unroll
    for (int c = 0; c < half; ++c) fft_bfly_inv(x[e + c], x[e + c + 8], x[e + c + half], x[e + c + half + 8], wr, wi);
    if (g != 0) {
      const int o = e + 2 * half;
#pragma unroll
      for (int c = 0; c < half; ++c) fft_bfly_inv_i(x[o + c], x[o + c + 8], x[o + c + half], x[o + c + half + 8], wr, wi);
    }
  }
}
template <int S, class TW>
RS_HD void fft_stage_fwd(double (&x)[kRegs], const TW& t) {
  FftStageTw w;
  fft_stage_tw<S>(t, w);
  fft_stage_fwd_tw<S>(x, w);
}
template <int S, class TW>
RS_HD void fft_stage_inv(double (&x)[kRegs], const TW& t) {
  FftStageTw w;
  fft_stage_tw<S>(t, w);
  fft_stage_inv_tw<S>(x, w);
}

// three stages of group G with all their twiddle reads issued up front (no LDS burst to hide behind here:
// fetched stage by stage, each read's latency would be exposed in turn)
template <int G, class TW>
RS_HD void fft_fwd3_ahead(double (&x)[kRegs], const TW& t) {
  FftStageTw w0, w1, w2;
  fft_stage_tw<3 * G>(t, w0); fft_stage_tw<3 * G + 1>(t, w1); fft_stage_tw<3 * G + 2>(t, w2);
  fft_stage_fwd_tw<3 * G>(x, w0); fft_stage_fwd_tw<3 * G + 1>(x, w1); fft_stage_fwd_tw<3 * G + 2>(x, w2);
}
template <int G, class TW>
RS_HD void fft_inv3_ahead(double (&x)[kRegs], const TW& t) {
  FftStageTw w0, w1, w2;
  fft_stage_tw<3 * G + 2>(t, w0); fft_stage_tw<3 * G + 1>(t, w1); fft_stage_tw<3 * G>(t, w2);
  fft_stage_inv_tw<3 * G + 2>(x, w0); fft_stage_inv_tw<3 * G + 1>(x, w1); fft_stage_inv_tw<3 * G>(x, w2);
}
RS_HD void fbuf_store(double* buf, int pos, double re, double im) { buf[2 * pos] = re; buf[2 * pos + 1] = im; }
RS_HD void fbuf_load(const double* buf, int pos, double& re, double& im) { re = buf[2 * pos]; im = buf[2 * pos + 1]; }

// ---- forward: input x[r] = real coefficient L + 64 r (r < 16), i.e. already folded ----
template <class TW>
RS_HD void ffwd_F1(int lane, double (&x)[kRegs], const TW& t, double* buf) {
  fft_stage_fwd<0>(x, t); fft_stage_fwd<1>(x, t); fft_stage_fwd<2>(x, t);
#pragma unroll
  for (int r = 0; r < kCRegs; ++r) fbuf_store(buf, fpos_t1(lane + 64 * r), x[r], x[r + 8]);
}
template <class TW>
RS_HD void ffwd_F2(int lane, double (&x)[kRegs], const TW& t, const double* buf) {
  const int b = lane >> 3, q = lane & 7;
#pragma unroll
  for (int s = 0; s < kCRegs; ++s) fbuf_load(buf, fpos_t1(64 * b + 8 * s + q), x[s], x[s + 8]);
  fft_fwd3_ahead<1>(x, t);
}
RS_HD void ffwd_F3(int lane, const double (&x)[kRegs], double* buf) {
  const int b = lane >> 3, q = lane & 7;
#pragma unroll
  for (int s = 0; s < kCRegs; ++s) fbuf_store(buf, fpos_t2(64 * b + 8 * s + q), x[s], x[s + 8]);
}
// output: x[u] + i x[u+8] = transform value at position 8*lane + u
template <class TW>
RS_HD void ffwd_F4(int lane, double (&x)[kRegs], const TW& t, const double* buf) {
#pragma unroll
  for (int u = 0; u < kCRegs; ++u) fbuf_load(buf, fpos_t2(8 * lane + u), x[u], x[u + 8]);
  fft_fwd3_ahead<2>(x, t);
}

// ---- inverse (mirror) ----
template <class TW>
RS_HD void finv_I1(int lane, double (&x)[kRegs], const TW& t, double* buf) {
  fft_inv3_ahead<2>(x, t);
#pragma unroll
  for (int u = 0; u < kCRegs; ++u) fbuf_store(buf, fpos_t2(8 * lane + u), x[u], x[u + 8]);
}
template <class TW>
RS_HD void finv_I2(int lane, double (&x)[kRegs], const TW& t, const double* buf) {
  const int b = lane >> 3, q = lane & 7;
#pragma unroll
  for (int s = 0; s < kCRegs; ++s) fbuf_load(buf, fpos_t2(64 * b + 8 * s + q), x[s], x[s + 8]);
  fft_inv3_ahead<1>(x, t);
}
RS_HD void finv_I3(int lane, const double (&x)[kRegs], double* buf) {
  const int b = lane >> 3, q = lane & 7;
#pragma unroll
  for (int s = 0; s < kCRegs; ++s) fbuf_store(buf, fpos_t1(64 * b + 8 * s + q), x[s], x[s + 8]);
}
// output: x[r] = real coefficient L + 64 r of the product (r < 16), before rounding
template <class TW>
RS_HD void finv_I4(int lane, double (&x)[kRegs], const TW& t, const double* buf) {
#pragma unroll
  for (int r = 0; r < kCRegs; ++r) fbuf_load(buf, fpos_t1(lane + 64 * r), x[r], x[r + 8]);
  fft_stage_inv<2>(x, t); fft_stage_inv<1>(x, t); fft_stage_inv<0>(x, t);
}

// ---- planar exchange: the re plane and the im plane pass through the SAME 576-double buffer one after
// the other, halving the LDS a wavefront needs for its transposes (the workgroup kernel spends the
// difference on a shared key-row ring). Same data movement as the interleaved form, so the
// transform values are bit-identical. Paddings chosen for conflict-free 8-byte accesses.
constexpr int kPlaneDoubles = 576;
RS_HD int ppos_t1(int j) { return j + 8 * (j >> 6); }
RS_HD int ppos_t2(int j) { return j + (j >> 3); }
enum { kLayA = 0, kLayB = 1, kLayC = 2 };   // A' j = L + 64 k; B' j = 64 (L>>3) + 8 k + (L&7); C' j = 8 L + k
template <int LAY>
RS_HD int flay_index(int lane, int k) {
  return LAY == kLayA ? lane + 64 * k : (LAY == kLayB ? 64 * (lane >> 3) + 8 * k + (lane & 7) : 8 * lane + k);
}
// Device: volatile stores stay eight separate ds_write_b64 (3 source dwords = 6 cycles on the VGPR -> LDS
// path each); merged into ds_write2_b64 by the compiler a pair costs 13 (MI355X LDS table). The store
// path is what bounds the exchange phases (halving the stores in a timing build: +22 %).
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_PLANE_STORE(buf, pos, v) (*((volatile __attribute__((address_space(3))) double*)(buf) + (pos)) = (v))
#else
#define RS_PLANE_STORE(buf, pos, v) ((buf)[pos] = (v))
#endif
// ---- row form of the B' <-> C' exchange (T = 2), A/B switch RS_ADDTID ----
// The exchange between the layouts B' (lane = 8a + c, registers b) and C' (lane = 8a + b, registers c) moves value
// (a, b, c) from register b of lane 8a + c to register c of lane 8a + b -- in either direction: register k of the
// writer becomes the low lane bits of the reader. Stored as ROWS (row k = register k of all 64 lanes, lane-major,
// low and high dwords of the doubles in separate rows), the writer needs no address at all: ds_write_addtid_b32 puts
// lane L's dword at M0 + offset + 4 L, 2 cycles of the VGPR -> LDS path per dword against 6 per ds_write_b64 (address +
// 2 data dwords) = 4 instead of 6 per value; the reader's eight values sit in 8 consecutive dwords of row (lane & 7)
// (low halves) and of its twin (high halves): four 16-byte reads per plane, the cycles of eight ds_read_b64. Row bases
// (dwords) give every 16-lane group of a ds_read_b128 all 64 banks: residues {0, 4, 32, 36} by k mod 4.
RS_HD int frow_base(int k, int hi) { const int c = k & 3; return 260 * c + 24 * (c >> 1) + 64 * (k >> 2) + 128 * hi; }   // classes at 0, 260, 544, 804

// ---- 8-byte stores, 16-byte loads (the default form) ----
// What an LDS instruction costs is vector-issue time of its SIMD (tools/lds_issue_bench.hip, profiles/r03/a_lds_issue_costs.jsonl:
// beside an FP64 stream at two waves per SIMD a ds_read_b64 costs 1.1-1.5 v_fma_f64, a ds_read_b128 1.5, a ds_write_b64 1.8-2.0,
// a ds_write_b128 3.4), and the blind rotation's time IS the sum of those costs. Moving a plane back
// with FOUR 16-byte loads instead of eight 8-byte ones takes 4.4 FMA-equivalents off each of the 32 plane exchanges of a CMUX.
// The reader's registers (2m, 2m+1) must then be adjacent doubles. With the value index written j = 64 a + 8 b + c
// (A': lane 8b+c, register a; B': lane 8a+c, register b; C': lane 8a+b, register c) the positions (in doubles) are
//   A' -> B':  16 a + 2 c + 128 (b >> 1) + (b & 1)          B' -> A':  16 b + 2 c + 128 (a >> 1) + (a & 1)
//   B' -> C':  8 a + c + 72 b + 2 (b & 3)                   C' -> B':  8 a + b + 72 c + 2 (c & 3)
// Every store instruction (one register of 64 lanes) covers 16 consecutive doubles per 16-lane group, every 16-byte load
// instruction 16 different 16-byte slots per lane group of ds_read_b128: conflict-free both ways (checked exhaustively on the
// host against the lane groups of MI355X_MICROARCH.md, tests/test_emulator.py), 574 doubles at most (kPlaneDoubles = 576).
template <int LAY>
RS_HD void flay_abc(int lane, int k, int& a, int& b, int& c) {
  if (LAY == kLayA) { a = k; b = lane >> 3; c = lane & 7; }
  else if (LAY == kLayB) { a = lane >> 3; b = k; c = lane & 7; }
  else { a = lane >> 3; b = lane & 7; c = k; }
}
template <int FROM, int TO>
RS_HD int xpos(int a, int b, int c) {
  if (FROM == kLayA) return 16 * a + 2 * c + 128 * (b >> 1) + (b & 1);                      // A' -> B'
  if (FROM == kLayB && TO == kLayA) return 16 * b + 2 * c + 128 * (a >> 1) + (a & 1);       // B' -> A'
  if (FROM == kLayB) return 8 * a + c + 72 * b + 2 * (b & 3);                               // B' -> C'
  return 8 * a + b + 72 * c + 2 * (c & 3);                                                  // C' -> B'
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef double rs_d2 __attribute__((ext_vector_type(2)));
#endif

template <int LAY, int T, int H>
RS_HD void fpl_store(int lane, const double (&x)[kRegs], double* buf) {
  constexpr int TO = (T == 1) ? (LAY == kLayA ? kLayB : kLayA) : (LAY == kLayB ? kLayC : kLayB);
  if constexpr (diag::kHalfExchangeProbe && H == 1) return;   // diagnostic builds (rs_diag.h bit 32): the im plane stays where it is
#pragma unroll
  for (int k = 0; k < kCRegs; ++k) { int a, b, c; flay_abc<LAY>(lane, k, a, b, c); RS_PLANE_STORE(buf, (xpos<LAY, TO>(a, b, c)), x[k + 8 * H]); }
}
// Device: volatile LDS loads stay eight separate ds_read_b64 (2 LDS cycles each); merged into
// ds_read2_b64 by the compiler they cost 8 cycles per pair (MI355X LDS table), i.e. twice as much.
#if defined(__HIP_DEVICE_COMPILE__)
#define RS_PLANE_LOAD(buf, pos) (*((const volatile __attribute__((address_space(3))) double*)(buf) + (pos)))
#else
#define RS_PLANE_LOAD(buf, pos) ((buf)[pos])
#endif
template <int LAY, int T, int H>
RS_HD void fpl_load(int lane, double (&x)[kRegs], const double* buf) {
  constexpr int FROM = (T == 1) ? (LAY == kLayB ? kLayA : kLayB) : (LAY == kLayC ? kLayB : kLayC);
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (diag::kHalfExchangeProbe && H == 1) {   // diagnostic builds (rs_diag.h bit 32): no im plane; 16 cross-half swaps per transform
    if constexpr (T == 1 && diag::kHalfExchangeSwaps) {
#pragma unroll
      for (int k = 0; k < kCRegs; ++k) {
        unsigned long long u = (unsigned long long)__builtin_bit_cast(long long, x[k]), v = (unsigned long long)__builtin_bit_cast(long long, x[k + 8]);
        unsigned ul = (unsigned)u, uh = (unsigned)(u >> 32), vl = (unsigned)v, vh = (unsigned)(v >> 32);
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ul), "+v"(vl));
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(uh), "+v"(vh));
        x[k] = __builtin_bit_cast(double, (long long)(((unsigned long long)uh << 32) | ul));
        x[k + 8] = __builtin_bit_cast(double, (long long)(((unsigned long long)vh << 32) | vl));
      }
    }
    return;
  }
#endif
#pragma unroll
  for (int m = 0; m < kCRegs / 2; ++m) {
    int a, b, c;
    flay_abc<LAY>(lane, 2 * m, a, b, c);
    const int pos = xpos<FROM, LAY>(a, b, c);   // even, and register 2m + 1 sits at pos + 1
#if defined(__HIP_DEVICE_COMPILE__)
    const rs_d2 v = *((const volatile __attribute__((address_space(3))) rs_d2*)(buf + pos));   // one ds_read_b128
    x[2 * m + 8 * H] = v.x; x[2 * m + 1 + 8 * H] = v.y;
#else
    x[2 * m + 8 * H] = buf[pos]; x[2 * m + 1 + 8 * H] = buf[pos + 1];
#endif
  }
}
// One exchange FROM layout L0 TO layout L1 through padding T; `sync` orders the wavefront's LDS
// accesses (on the device a compiler-only fence: DS operations of one wavefront execute in order).
template <int L0, int L1, int T, class Sync>
RS_HD void fpl_exchange(int lane, double (&x)[kRegs], double* buf, Sync sync) {
  fpl_store<L0, T, 0>(lane, x, buf); sync();
  fpl_load<L1, T, 0>(lane, x, buf); sync();
  fpl_store<L0, T, 1>(lane, x, buf); sync();
  fpl_load<L1, T, 1>(lane, x, buf); sync();
}
// ---- two transforms in flight ----
// A wavefront's LDS operations execute in order, so a second transform may push its exchange through
// the SAME buffer as soon as the first one's loads have been ISSUED: its stores queue behind them.
// Interleaving two independent transforms phase by phase lets the butterflies of one cover the LDS
// round trip of the other inside a single wavefront (the kernels run only two wavefronts per SIMD).
template <int LAY, int T>
RS_HD void fil_store(int lane, const double (&x)[kRegs], double* buf) {
#pragma unroll
  for (int k = 0; k < kCRegs; ++k) { const int j = flay_index<LAY>(lane, k); fbuf_store(buf, T == 1 ? fpos_t1(j) : fpos_t2(j), x[k], x[k + 8]); }
}
template <int LAY, int T>
RS_HD void fil_load(int lane, double (&x)[kRegs], const double* buf) {
#pragma unroll
  for (int k = 0; k < kCRegs; ++k) { const int j = flay_index<LAY>(lane, k); fbuf_load(buf, T == 1 ? fpos_t1(j) : fpos_t2(j), x[k], x[k + 8]); }
}
template <int G, class TW>
RS_HD void fft_fwd3(double (&x)[kRegs], const TW& t) { fft_stage_fwd<3 * G>(x, t); fft_stage_fwd<3 * G + 1>(x, t); fft_stage_fwd<3 * G + 2>(x, t); }
template <int G, class TW>
RS_HD void fft_inv3(double (&x)[kRegs], const TW& t) { fft_stage_inv<3 * G + 2>(x, t); fft_stage_inv<3 * G + 1>(x, t); fft_stage_inv<3 * G>(x, t); }

// PLANAR: one 8-byte plane at a time through a kPlaneDoubles buffer; else (re, im) pairs through kBufDoubles.
// Exchange of x with `work(0..2)` -- three butterfly stages of the OTHER transform -- placed between
// its LDS phases, so that the vector ALU has independent work while the stores drain (a 16-byte-per-
// lane store occupies the LDS issue path for ~13 cycles) and the loads return.
// (A sched_barrier behind every burst, to force it out before the butterflies it overlaps with, was measured at -0.5 %:
// with the twiddles a burst ahead the scheduler's own placement is better.)
template <bool PLANAR, int L0, int L1, int T, class Sync, class Pre, class Run>
RS_HD void fft_exchange_with(int lane, double (&x)[kRegs], double* buf, Sync sync, Pre pre, Run run) {
  // pre(k) issues the twiddle reads of work chunk k, run(k) is its butterflies. Every pre() is issued a
  // whole LDS burst AHEAD of its run() (see FftStageTw): wherever the scheduler then puts the burst that
  // run(k) overlaps with, the twiddles of run(k) are older than it and are waited for with a counted lgkmcnt.
  if (PLANAR) {
    pre(0); pre(1); sync(); fpl_store<L0, T, 0>(lane, x, buf); run(0); sync();
    pre(2); sync(); fpl_load<L1, T, 0>(lane, x, buf); run(1); sync();
    fpl_store<L0, T, 1>(lane, x, buf);
    run(2); sync();
    fpl_load<L1, T, 1>(lane, x, buf);
    sync();
  } else {
    pre(0); pre(1); sync(); fil_store<L0, T>(lane, x, buf); run(0); pre(2); run(1); sync();
    fil_load<L1, T>(lane, x, buf); run(2); sync();
  }
}
// k-th stage (in execution order) of group G: twiddle fetch and butterflies
template <int G, bool INV, class TW>
RS_HD void fft_group_tw(const TW& t, int k, FftStageTw& w) {
  if (!INV) {
    if (k == 0) fft_stage_tw<3 * G>(t, w); else if (k == 1) fft_stage_tw<3 * G + 1>(t, w); else fft_stage_tw<3 * G + 2>(t, w);
  } else {
    if (k == 0) fft_stage_tw<3 * G + 2>(t, w); else if (k == 1) fft_stage_tw<3 * G + 1>(t, w); else fft_stage_tw<3 * G>(t, w);
  }
}
template <int G, bool INV>
RS_HD void fft_group_run(double (&x)[kRegs], const FftStageTw& w, int k) {
  if (!INV) {
    if (k == 0) fft_stage_fwd_tw<3 * G>(x, w); else if (k == 1) fft_stage_fwd_tw<3 * G + 1>(x, w); else fft_stage_fwd_tw<3 * G + 2>(x, w);
  } else {
    if (k == 0) fft_stage_inv_tw<3 * G + 2>(x, w); else if (k == 1) fft_stage_inv_tw<3 * G + 1>(x, w); else fft_stage_inv_tw<3 * G>(x, w);
  }
}
// one exchange of x (layout L0 -> L1) overlapped with the three stages of group G of the other transform y
template <bool PLANAR, int L0, int L1, int T, int G, bool INV, class TW, class Sync>
RS_HD void fft_exchange_over(int lane, double (&x)[kRegs], double (&y)[kRegs], const TW& t, double* buf, Sync sync) {
  FftStageTw w[3];
  fft_exchange_with<PLANAR, L0, L1, T>(lane, x, buf, sync, [&](int k) { fft_group_tw<G, INV>(t, k, w[k]); },
                                       [&](int k) { fft_group_run<G, INV>(y, w[k], k); });
}

// The same with the twiddles of group G kept by the caller: the two transforms of a pipelined pair pass through the same
// stages of the same lane half a segment apart, so the values fetched for the first one (FETCH) serve the second one too
// (24 registers across one exchange instead of 8 more 16-byte LDS reads per pair and group).
template <bool PLANAR, int L0, int L1, int T, int G, bool INV, bool FETCH, class TW, class Sync>
RS_HD void fft_exchange_over_keep(int lane, double (&x)[kRegs], double (&y)[kRegs], const TW& t, double* buf, Sync sync, FftStageTw (&w)[3]) {
  fft_exchange_with<PLANAR, L0, L1, T>(lane, x, buf, sync, [&](int k) { if (FETCH) fft_group_tw<G, INV>(t, k, w[k]); },
                                       [&](int k) { fft_group_run<G, INV>(y, w[k], k); });
}

// single inverse transform, planar exchanges (duo kernel): each group's twiddles fetched up front
template <class TW, class Sync>
RS_HD void finv_planar(int lane, double (&x)[kRegs], const TW& t, double* buf, Sync sync) {
  fft_inv3_ahead<2>(x, t);
  fpl_exchange<kLayC, kLayB, 2>(lane, x, buf, sync);
  fft_inv3_ahead<1>(x, t);
  fpl_exchange<kLayB, kLayA, 1>(lane, x, buf, sync);
  fft_stage_inv<2>(x, t); fft_stage_inv<1>(x, t); fft_stage_inv<0>(x, t);
}
// single forward transform, planar exchanges (split-key workgroup kernel)
template <class TW, class Sync>
RS_HD void ffwd_planar(int lane, double (&x)[kRegs], const TW& t, double* buf, Sync sync) {
  fft_fwd3_ahead<0>(x, t);
  fpl_exchange<kLayA, kLayB, 1>(lane, x, buf, sync);
  fft_fwd3_ahead<1>(x, t);
  fpl_exchange<kLayB, kLayC, 2>(lane, x, buf, sync);
  fft_fwd3_ahead<2>(x, t);
}
// Two forward transforms as one software pipeline: six segments, each an LDS phase of one transform's exchange paired with a
// butterfly stage group of the other.
template <bool PLANAR, class TW, class Sync>
RS_HD void ffwd_pair(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const TW& t, double* buf, Sync sync) {
  fft_fwd3<0>(xa, t);
  fft_exchange_over<PLANAR, kLayA, kLayB, 1, 0, false>(lane, xa, xb, t, buf, sync);
  FftStageTw w[3];
  fft_exchange_over_keep<PLANAR, kLayA, kLayB, 1, 1, false, true>(lane, xb, xa, t, buf, sync, w);
  fft_exchange_over_keep<PLANAR, kLayB, kLayC, 2, 1, false, false>(lane, xa, xb, t, buf, sync, w);
  fft_exchange_over_keep<PLANAR, kLayB, kLayC, 2, 2, false, true>(lane, xb, xa, t, buf, sync, w);
  fft_stage_fwd_tw<6>(xb, w[0]); fft_stage_fwd_tw<7>(xb, w[1]); fft_stage_fwd_tw<8>(xb, w[2]);
}
template <bool PLANAR, class TW, class Sync>
RS_HD void finv_pair(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const TW& t, double* buf, Sync sync) {
  FftStageTw wk[3];   // group 2 (stages 8, 7, 6) in execution order; xb reuses them in segment 1
  fft_group_tw<2, true>(t, 0, wk[0]); fft_group_tw<2, true>(t, 1, wk[1]); fft_group_tw<2, true>(t, 2, wk[2]);
  fft_group_run<2, true>(xa, wk[0], 0); fft_group_run<2, true>(xa, wk[1], 1); fft_group_run<2, true>(xa, wk[2], 2);
  fft_exchange_over_keep<PLANAR, kLayC, kLayB, 2, 2, true, false>(lane, xa, xb, t, buf, sync, wk);
  fft_exchange_over_keep<PLANAR, kLayC, kLayB, 2, 1, true, true>(lane, xb, xa, t, buf, sync, wk);
  fft_exchange_over_keep<PLANAR, kLayB, kLayA, 1, 1, true, false>(lane, xa, xb, t, buf, sync, wk);
  fft_exchange_over<PLANAR, kLayB, kLayA, 1, 0, true>(lane, xb, xa, t, buf, sync);
  fft_inv3<0>(xb, t);
}

// pointwise complex multiply-accumulate: (sr, si) += (xr, xi) * (wr, wi)
RS_HD void fft_cmac(double& sr, double& si, double xr, double xi, double wr, double wi) {
  sr = __builtin_fma(xr, wr, sr);
  sr = __builtin_fma(-xi, wi, sr);
  si = __builtin_fma(xr, wi, si);
  si = __builtin_fma(xi, wr, si);
}

// round an almost-integer double (|v| < 2^51) to torus32 and report its distance to that integer
RS_HD int32_t fft_round_torus32(double v, double& max_dev) {
  const double t = v + 6755399441055744.0;       // 1.5 * 2^52: t's low mantissa bits = rint(v)
  const double r = t - 6755399441055744.0;
  max_dev = __builtin_fmax(max_dev, __builtin_fabs(v - r));   // one v_max_f64 with an |abs| source modifier
  long long bits;
  __builtin_memcpy(&bits, &t, sizeof(bits));
  return (int32_t)(uint32_t)(unsigned long long)bits;
}

}  // namespace rs
