// rs_kernels.hip -- gfx950 kernels of the gate-bootstrapping hot path.
//
//   bk_transform_kernel    bootstrapping key -> transform domain (the bkFFT analogue; once per key)
//   blind_rotate_kernel    gate pre-combination + modswitch + n CMUX steps + sample extract
//                          (tfhe_bootstrap_woKS_FFT; REDsec: lib/BinOps_enc.cpp:185,191)
//   keyswitch_kernel       lweKeySwitch (N -> n)
//   polymul_kernel         debug/parity tap through the same transform path
//   lincomb / linear_fc / conv_ternary / sumpool   LWE word arithmetic of the layer linear stage
//
// One wavefront owns one ciphertext for the whole blind rotation: its TRLWE accumulator (2 x 1024
// int32) lives in LDS, each of the (k+1) l digit polynomials is transformed in registers with two
// LDS transposes (rs_ntt.h), multiplied against the coalesced-streamed key row and accumulated in
// registers, and two inverse transforms update the accumulator. Waves never synchronise with each
// other after the twiddle tables are staged, so the 2 waves per SIMD interleave freely.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rs_kernels.h"
#include "rs_ntt.h"

namespace rs {

// Same-wave LDS hand-off: DS operations of one wavefront execute in order, so only the compiler
// needs to be told not to move LDS accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
#if defined(RS_EXP_NOSYNC)  // timing experiment: let the compiler reorder/merge across LDS hand-offs
  return;
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Diagnostic build only (-DRS_STAMPS): per-phase cycle stamps, accumulated per wave and written to
// a debug buffer nothing else reads. Never enabled in the product build.
#if defined(RS_STAMPS)
#define RS_NSTAMP 12
struct Stamps { unsigned long long t, acc[RS_NSTAMP]; };
__device__ __forceinline__ void stamp_start(Stamps& s) {
  __builtin_amdgcn_sched_barrier(0);
  s.t = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void stamp(Stamps& s, int k) {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long now = __builtin_amdgcn_s_memtime();
  s.acc[k] += now - s.t;
  s.t = now;
  __builtin_amdgcn_sched_barrier(0);
}
#define RS_STAMP(k) stamp(st, k)
#else
#define RS_STAMP(k)
#endif

template <class C>
__device__ __forceinline__ void ntt_forward(int lane, double (&x)[kRegs], const double* tw, double* buf, const Field& f) {
  fwd_F1<C>(lane, x, tw, buf, f);
  wave_lds_sync();
  fwd_F2<C>(lane, x, tw, buf, f);
  wave_lds_sync();
  fwd_F3(lane, x, buf);
  wave_lds_sync();
  fwd_F4<C>(lane, x, tw, buf, f);
  wave_lds_sync();
}

// forward transform of gadget digit q of the coefficients d (fused stages 0-1, rs_ntt.h)
template <class C>
__device__ __forceinline__ void ntt_forward_digits(int lane, double (&x)[kRegs], const int32_t (&d)[kRegs], int q, uint32_t offset,
                                                   const double* tw, double* buf, const Field& f
#if defined(RS_STAMPS)
                                                   , Stamps& st
#endif
) {
  fwd_F1_digits<C>(lane, x, d, q, offset, tw, buf, f);
  wave_lds_sync();
  RS_STAMP(1);
  fwd_F2<C>(lane, x, tw, buf, f);
  wave_lds_sync();
  RS_STAMP(2);
  fwd_F3(lane, x, buf);
  wave_lds_sync();
  RS_STAMP(3);
  fwd_F4<C>(lane, x, tw, buf, f);
  wave_lds_sync();
  RS_STAMP(4);
}

template <class C>
__device__ __forceinline__ void ntt_inverse(int lane, double (&x)[kRegs], const double* twi, double* buf, const Field& f) {
  inv_I1<C>(lane, x, twi, buf, f);
  wave_lds_sync();
  inv_I2<C>(lane, x, twi, buf, f);
  wave_lds_sync();
  inv_I3(lane, x, buf);
  wave_lds_sync();
  inv_I4<C>(lane, x, twi, buf, f);
  wave_lds_sync();
}

__device__ __forceinline__ void stage_tables(double* s_tw, const double* tw_g, int nthreads, int count = 2 * kN) {
  for (int i = threadIdx.x; i < count; i += nthreads) s_tw[i] = tw_g[i];
  __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// Key transform: one wavefront per key polynomial. Output layout per polynomial: [v 0..7][lane][2]
// doubles = transform positions 16*lane + 2v, +1, so that the blind rotation reads each row with
// eight perfectly coalesced 16-byte-per-lane loads. Values are scaled by 1/N and fully reduced.
// -------------------------------------------------------------------------------------------------
template <class C, int WPB>
__global__ __launch_bounds__(64 * WPB) void bk_transform_kernel(const int32_t* __restrict__ bk, double* __restrict__ bk_ntt,
                                                                 const double* __restrict__ tw_g, Field f, double ninv,
                                                                 long n_polys) {
  __shared__ double s_tw[2 * kN];
  __shared__ double s_buf[WPB][kBufDoubles];
  stage_tables(s_tw, tw_g, 64 * WPB);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long poly = (long)blockIdx.x * WPB + wave;
  if (poly >= n_polys) return;
  double x[kRegs];
  const int32_t* src = bk + poly * kN;
#pragma unroll
  for (int r = 0; r < kRegs; ++r) x[r] = (double)src[lane + 64 * r];
  ntt_forward<C>(lane, x, s_tw, s_buf[wave], f);
  double2* dst = reinterpret_cast<double2*>(bk_ntt + poly * kN);
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    double a = f_reduce(f_mulmod(f_reduce(x[2 * v], f), ninv, f), f);
    double b = f_reduce(f_mulmod(f_reduce(x[2 * v + 1], f), ninv, f), f);
    dst[v * 64 + lane] = make_double2(a, b);
  }
}

// -------------------------------------------------------------------------------------------------
// Blind rotation + sample extract.
// -------------------------------------------------------------------------------------------------
template <class C, int WPB>
__global__ __launch_bounds__(64 * WPB) void blind_rotate_kernel(BlindRotateArgs a) {
  __shared__ double s_tw[kTwTotal];
  __shared__ double s_buf[WPB][kBufDoubles];
  __shared__ int32_t s_acc[WPB][2][kN];
  stage_tables(s_tw, a.tw, 64 * WPB, kTwTotal);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  // De-phase knobs (measured: no effect, kept for experiments): waves 4-7 start late / higher prio.
  if (WPB >= 8 && wave >= 4) {
    for (int k = 0; k < a.stagger; ++k) __builtin_amdgcn_s_sleep(8);
    if (a.prio) __builtin_amdgcn_s_setprio(1);
  }
  // Persistent waves: the first ciphertext is assigned statically, further ones are pulled from a
  // device counter (zeroed by the launcher on the same stream). A 152 KB-LDS workgroup cannot be
  // replaced until its LAST wave exits, and waves sharing a SIMD finish up to 20 % apart (issue
  // arbitration favours the older wave), which left 18 % of the wave slots idle with one
  // ciphertext per wave (profiles/r01: stamps build). Every wave leaves the loop as soon as the
  // counter passes B, so the grid always drains.
  long ct = (long)blockIdx.x * WPB + wave;
  const long first_dynamic = (long)gridDim.x * WPB;
  if (ct >= a.B) return;

  const Field f = a.f;
  double* buf = s_buf[wave];
  int32_t* acc0 = s_acc[wave][0];
  int32_t* acc1 = s_acc[wave][1];
  const double* tw = s_tw;
  const double* twi = s_tw + kN;
 for (;;) {
  const int32_t* row0 = a.in0 + ct * a.W;
  const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
  const int n = a.n;

  // gate pre-combination (0, bconst) + c0*in0 + c1*in1, evaluated word by word as it is consumed
  auto word = [&](int i) -> int32_t {
    uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
    if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
    return (int32_t)v;
  };

  {
    const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
    const int rot = 2 * kN - barb;  // in (0, 2N]
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      acc0[j] = 0;
      acc1[j] = rotated_const(a.mu, j, rot);
    }
  }
  wave_lds_sync();

  constexpr uint32_t offset = gadget_offset<C>();
  constexpr int KPL = 2 * C::L;
#if defined(RS_STAMPS)
  Stamps st;
  for (int k = 0; k < RS_NSTAMP; ++k) st.acc[k] = 0;
  const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
  stamp_start(st);
#endif

  for (int i = 0; i < n; ++i) {
    const int32_t bara = __builtin_amdgcn_readfirstlane(modswitch_2N(word(i)));
    if (bara == 0) continue;  // tfhe_blindRotate_FFT skips the identity CMUX
    if (WPB >= 8 && a.prio == 2) {   // experiment: alternate issue priority between the SIMD partners
      if (((wave >> 2) ^ i) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    double s0[kRegs], s1[kRegs];
#pragma unroll
    for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
    const double* bk_i = a.bk_ntt + (size_t)i * KPL * 2 * kN;

#pragma unroll 1
    for (int comp = 0; comp < 2; ++comp) {
      const int32_t* accc = comp ? acc1 : acc0;
      int32_t d[kRegs];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) d[r] = rotated_diff(accc, lane + 64 * r, bara);
      RS_STAMP(0);
#if defined(RS_EXP_UNROLLQ)
#pragma unroll
#else
#pragma unroll 1
#endif
      for (int q = 0; q < C::L; ++q) {
        const int row = comp * C::L + q;
        if (WPB >= 8 && a.prio == 3) {   // experiment: alternate priority per digit row
          if (((wave >> 2) ^ row) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)(row * 2) * kN);
        const double2* bp1 = bp0 + kN / 2;
        double2 w0[8], w1[8];
#if defined(RS_EXP_NOBK)   // timing experiment: no key-row loads
#pragma unroll
        for (int v = 0; v < 8; ++v) { w0[v] = make_double2(1.0 + lane + q, 2.0 + v); w1[v] = make_double2(3.0 + v, 5.0 + lane); }
#else
#pragma unroll
        for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
#endif
        double x[kRegs];
#if defined(RS_STAMPS)
        RS_STAMP(5);
        ntt_forward_digits<C>(lane, x, d, q, offset, tw, buf, f, st);
#else
        ntt_forward_digits<C>(lane, x, d, q, offset, tw, buf, f);
#endif
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          s0[2 * v] += f_mulmod(x[2 * v], w0[v].x, f);
          s0[2 * v + 1] += f_mulmod(x[2 * v + 1], w0[v].y, f);
          s1[2 * v] += f_mulmod(x[2 * v], w1[v].x, f);
          s1[2 * v + 1] += f_mulmod(x[2 * v + 1], w1[v].y, f);
        }
        RS_STAMP(6);
      }
      if (C::MID_REDUCE && comp == 0) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) { s0[u] = f_reduce(s0[u], f); s1[u] = f_reduce(s1[u], f); }
      }
    }

    RS_STAMP(7);
    ntt_inverse<C>(lane, s0, twi, buf, f);
    RS_STAMP(8);
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      acc0[j] = (int32_t)((uint32_t)acc0[j] + (uint32_t)f_to_torus32(s0[r]));
    }
    RS_STAMP(9);
    ntt_inverse<C>(lane, s1, twi, buf, f);
    RS_STAMP(10);
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      acc1[j] = (int32_t)((uint32_t)acc1[j] + (uint32_t)f_to_torus32(s1[r]));
    }
    wave_lds_sync();
  }

#if defined(RS_STAMPS)
  RS_STAMP(11);
  st.acc[11] = __builtin_amdgcn_s_memrealtime() - real0;   // 100 MHz ticks for the whole rotation
  if (lane == 0 && a.debug)
    for (int k = 0; k < RS_NSTAMP; ++k) a.debug[ct * RS_NSTAMP + k] = st.acc[k];
#endif
  // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
  int32_t* out = a.u_out + ct * (kN + 1);
#pragma unroll
  for (int r = 0; r < kRegs; ++r) {
    const int j = lane + 64 * r;
    out[j] = (j == 0) ? acc0[0] : (int32_t)(0u - (uint32_t)acc0[kN - j]);
  }
  if (lane == 0) out[kN] = acc1[0];

  if (!a.counter) break;
  unsigned int nxt = 0;
  if (lane == 0) nxt = atomicAdd(a.counter, 1u);
  nxt = (unsigned int)__builtin_amdgcn_readfirstlane((int)nxt);
  ct = first_dynamic + (long)nxt;
  if (ct >= a.B) break;
  wave_lds_sync();
 }
}

// -------------------------------------------------------------------------------------------------
// Cooperative blind rotation (latency form, B <= 2 x #CUs): G waves share ONE ciphertext.
// Wave g transforms the digit polynomials [g R, (g+1) R) (R = 2l / G, so each wave stays within one
// accumulator component) and accumulates its partial column sums; the partials meet in LDS, waves 0
// and 1 sum one column each, run the inverse transform and update the shared accumulator. Two
// workgroup barriers per CMUX step. A 196-neuron layer thus spreads over 196 CUs x 4 SIMDs
// instead of one wave per CU (MNIST layer 0: 18.4 ms -> see profiles/).
// -------------------------------------------------------------------------------------------------
#if !defined(RS_STAMPS)
template <class C, int G>
__global__ __launch_bounds__(64 * G) void blind_rotate_coop_kernel(BlindRotateArgs a) {
  constexpr int KPL = 2 * C::L;
  constexpr int R = KPL / G;
  static_assert(KPL % G == 0 && G % 2 == 0, "waves must split the digit rows evenly within a component");
  __shared__ double s_tw[kTwTotal];
  __shared__ double s_buf[G][kBufDoubles];
  __shared__ double s_part[G][2][kN];
  __shared__ int32_t s_acc[2][kN];
  stage_tables(s_tw, a.tw, 64 * G, kTwTotal);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long ct = blockIdx.x;
  const Field f = a.f;
  double* buf = s_buf[wave];
  const double* tw = s_tw;
  const double* twi = s_tw + kN;
  const int32_t* row0 = a.in0 + ct * a.W;
  const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
  const int n = a.n;
  const int comp = wave / (G / 2);
  const int row_begin = wave * R;
  auto word = [&](int i) -> int32_t {
    uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
    if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
    return (int32_t)v;
  };
  if (wave < 2) {
    const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
    const int rot = 2 * kN - barb;
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      s_acc[wave][j] = wave == 0 ? 0 : rotated_const(a.mu, j, rot);
    }
  }
  __syncthreads();
  constexpr uint32_t offset = gadget_offset<C>();
  for (int i = 0; i < n; ++i) {
    const int32_t bara = __builtin_amdgcn_readfirstlane(modswitch_2N(word(i)));
    if (bara == 0) continue;   // uniform over the workgroup: every wave works on the same ciphertext
    double s0[kRegs], s1[kRegs];
#pragma unroll
    for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
    const double* bk_i = a.bk_ntt + (size_t)i * KPL * 2 * kN;
    int32_t d[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; ++r) d[r] = rotated_diff(s_acc[comp], lane + 64 * r, bara);
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
      const int row = row_begin + rr;
      const int q = row - comp * C::L;
      const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)(row * 2) * kN);
      const double2* bp1 = bp0 + kN / 2;
      double2 w0[8], w1[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
      double x[kRegs];
      ntt_forward_digits<C>(lane, x, d, q, offset, tw, buf, f);
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        s0[2 * v] += f_mulmod(x[2 * v], w0[v].x, f);
        s0[2 * v + 1] += f_mulmod(x[2 * v + 1], w0[v].y, f);
        s1[2 * v] += f_mulmod(x[2 * v], w1[v].x, f);
        s1[2 * v + 1] += f_mulmod(x[2 * v + 1], w1[v].y, f);
      }
    }
    // partial sums (<= R products each) reduced, then exchanged: position u*64 + lane is conflict-free
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      s_part[wave][0][u * 64 + lane] = f_reduce(s0[u], f);
      s_part[wave][1][u * 64 + lane] = f_reduce(s1[u], f);
    }
    __syncthreads();   // partials visible; every wave has finished reading the accumulator
    if (wave < 2) {
      double x[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) {
        double t = s_part[0][wave][u * 64 + lane];
#pragma unroll
        for (int g = 1; g < G; ++g) t += s_part[g][wave][u * 64 + lane];
        x[u] = t;
      }
      ntt_inverse<C>(lane, x, twi, buf, f);
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        s_acc[wave][j] = (int32_t)((uint32_t)s_acc[wave][j] + (uint32_t)f_to_torus32(x[r]));
      }
    }
    __syncthreads();   // accumulator updated
  }
  int32_t* out = a.u_out + ct * (kN + 1);
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      out[j] = (j == 0) ? s_acc[0][0] : (int32_t)(0u - (uint32_t)s_acc[0][kN - j]);
    }
    if (lane == 0) out[kN] = s_acc[1][0];
  }
}
#endif

// -------------------------------------------------------------------------------------------------
// Keyswitch: one workgroup per ciphertext, threads over output words. The KSK (83-104 MB) stays in
// the Infinity Cache; rows are gathered by digit. u = u0 (+ u1) (+ bconst on the b word): the sum
// form serves bootsMUX.
// -------------------------------------------------------------------------------------------------
constexpr int KS_THREADS = 256;
constexpr int KS_MAXR = 4;  // output words per thread: W <= 1024

__global__ __launch_bounds__(KS_THREADS) void keyswitch_kernel(KeyswitchArgs a) {
  const long ct = blockIdx.x;
  const int tid = threadIdx.x;
  const int32_t* u0 = a.u0 + ct * (kN + 1);
  const int32_t* u1 = a.u1 ? a.u1 + ct * (kN + 1) : nullptr;
  uint32_t acc[KS_MAXR];
#pragma unroll
  for (int k = 0; k < KS_MAXR; ++k) acc[k] = 0;
  const int W = a.W, t = a.t, basebit = a.basebit;
  const uint32_t prec_offset = 1u << (32 - (1 + basebit * t));
  const uint32_t mask = (1u << basebit) - 1u;
  for (int i = 0; i < kN; ++i) {
    uint32_t ai = (uint32_t)u0[i];
    if (u1) ai += (uint32_t)u1[i];
    const uint32_t aibar = ai + prec_offset;
    for (int j = 0; j < t; ++j) {
      const uint32_t dgt = (aibar >> (32 - (j + 1) * basebit)) & mask;
      if (dgt == 0) continue;
      const int32_t* row = a.ksk + ((((size_t)i * t + j) << basebit) + dgt) * (size_t)W;
#pragma unroll
      for (int k = 0; k < KS_MAXR; ++k) {
        const int w = tid + k * KS_THREADS;
        if (w < W) acc[k] += (uint32_t)row[w];
      }
    }
  }
  uint32_t bw = (uint32_t)u0[kN];
  if (u1) bw += (uint32_t)u1[kN];
  bw += (uint32_t)a.bconst;
  int32_t* out = a.out + ct * W;
#pragma unroll
  for (int k = 0; k < KS_MAXR; ++k) {
    const int w = tid + k * KS_THREADS;
    if (w < W) out[w] = (int32_t)((w == W - 1 ? bw : 0u) - acc[k]);
  }
}

// -------------------------------------------------------------------------------------------------
// Tiled keyswitch (the production path for the shipped parameter sets).
//
// A workgroup of 4 waves owns 256 ciphertexts x KS_CH output words; lane = ciphertext. For a group
// of KS_IG input coefficients the KSK slice [KS_IG][t][base][KS_CH] is staged in LDS once (row v=0
// is a resident zero row), then every lane extracts its own digits and subtracts the LDS row they
// select: one ds_read_b128 + four v_sub per four output words. Each KSK byte fetched from L2 thus
// serves 256 ciphertexts instead of one, and the extracted samples are read once per word chunk.
// Rows are padded by 16 B so that the <= 8 distinct rows a wave touches sit in distinct bank quads.
// -------------------------------------------------------------------------------------------------
constexpr int KS_CH = 32;            // output words per workgroup
constexpr int KS_CHP = KS_CH + 4;    // padded row (words)
constexpr int KS_TILE_THREADS = 256;

template <int T, int BASEBIT, int KS_IG>  // KS_IG = input coefficients staged per round
__global__ __launch_bounds__(KS_TILE_THREADS) void keyswitch_tiled_kernel(KeyswitchArgs a) {
  constexpr int BASE = 1 << BASEBIT;
  constexpr int ROWS = KS_IG * T * BASE;
  __shared__ __attribute__((aligned(16))) int32_t s_ksk[2][ROWS * KS_CHP];
  const int tid = threadIdx.x;
  const long ct = (long)blockIdx.x * KS_TILE_THREADS + tid;
  const bool live = ct < a.B;
  const int w0 = blockIdx.y * KS_CH;
  const int W = a.W;
  const int32_t* u0 = a.u0 + (live ? ct : 0) * (kN + 1);
  const int32_t* u1 = a.u1 ? a.u1 + (live ? ct : 0) * (kN + 1) : nullptr;

  for (int e = tid; e < 2 * ROWS * KS_CHP; e += KS_TILE_THREADS) (&s_ksk[0][0])[e] = 0;

  uint32_t acc[KS_CH];
#pragma unroll
  for (int k = 0; k < KS_CH; ++k) acc[k] = 0;

  constexpr uint32_t prec_offset = 1u << (32 - (1 + BASEBIT * T));
  constexpr uint32_t mask = (1u << BASEBIT) - 1u;
  // staging: KS_IG * T * (BASE-1) row segments of KS_CH words; 8 threads x 16 B per segment
  constexpr int SEGS = KS_IG * T * (BASE - 1);
  auto stage = [&](int buf, int i0) {
    for (int sidx = tid; sidx < SEGS * 8; sidx += KS_TILE_THREADS) {
      const int seg = sidx >> 3, part = sidx & 7;
      const int v = seg % (BASE - 1) + 1;
      const int ij = seg / (BASE - 1);          // ii * T + j
      const int wq = w0 + part * 4;
      const int32_t* src = a.ksk + (((size_t)i0 * T + ij) * BASE + v) * (size_t)W + wq;
      int32_t* dst = &s_ksk[buf][(ij * BASE + v) * KS_CHP + part * 4];
      if (wq + 3 < W) {
        // rows are only 4-byte aligned in general (W odd): assemble from scalar loads
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = (wq + e < W) ? src[e] : 0;
      }
    }
  };

  // blockIdx.z selects a slice of the N input coefficients (latency form for small batches: the
  // slices add their partial sums into a zeroed output with integer atomics -- exact and
  // order-independent mod 2^32); gridDim.z == 1 is the plain-store throughput form.
  const int groups_per_split = (kN / KS_IG) / (int)gridDim.z;
  const int g_begin = (int)blockIdx.z * groups_per_split, g_end = g_begin + groups_per_split;
  __syncthreads();
  stage(g_begin & 1, g_begin * KS_IG);
  __syncthreads();
  for (int g = g_begin; g < g_end; ++g) {
    const int buf = g & 1;
    if (g + 1 < g_end) stage(buf ^ 1, (g + 1) * KS_IG);
    const int i0 = g * KS_IG;
    uint32_t ai[KS_IG];
#pragma unroll
    for (int ii = 0; ii < KS_IG; ++ii) {
      uint32_t v = live ? (uint32_t)u0[i0 + ii] : 0u;
      if (u1 && live) v += (uint32_t)u1[i0 + ii];
      ai[ii] = live ? v + prec_offset : 0u;
    }
#pragma unroll
    for (int ii = 0; ii < KS_IG; ++ii) {
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const uint32_t dgt = (ai[ii] >> (32 - (j + 1) * BASEBIT)) & mask;
        const int4* row = reinterpret_cast<const int4*>(&s_ksk[buf][((ii * T + j) * BASE + (int)dgt) * KS_CHP]);
#pragma unroll
        for (int q = 0; q < KS_CH / 4; ++q) {
          const int4 r = row[q];
          acc[4 * q + 0] += (uint32_t)r.x;
          acc[4 * q + 1] += (uint32_t)r.y;
          acc[4 * q + 2] += (uint32_t)r.z;
          acc[4 * q + 3] += (uint32_t)r.w;
        }
      }
    }
    __syncthreads();
  }
  if (!live) return;
  uint32_t bw = 0;
  if (blockIdx.z == 0) {
    bw = (uint32_t)u0[kN];
    if (u1) bw += (uint32_t)u1[kN];
    bw += (uint32_t)a.bconst;
  }
  int32_t* out = a.out + ct * W + w0;
  if (gridDim.z == 1) {
#pragma unroll
    for (int k = 0; k < KS_CH; ++k) {
      const int w = w0 + k;
      if (w < W) out[k] = (int32_t)((w == W - 1 ? bw : 0u) - acc[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < KS_CH; ++k) {
      const int w = w0 + k;
      if (w < W) atomicAdd(reinterpret_cast<unsigned int*>(out + k), (w == W - 1 ? bw : 0u) - acc[k]);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Debug tap: out = a_small * b_torus (negacyclic, mod 2^32) through forward/pointwise/inverse.
// -------------------------------------------------------------------------------------------------
template <class C, int WPB>
__global__ __launch_bounds__(64 * WPB) void polymul_kernel(const int32_t* __restrict__ a_small, const int32_t* __restrict__ b_torus,
                                                            int32_t* __restrict__ out, const double* __restrict__ tw_g, Field f,
                                                            double ninv, long count) {
  __shared__ double s_tw[2 * kN];
  __shared__ double s_buf[WPB][kBufDoubles];
  stage_tables(s_tw, tw_g, 64 * WPB);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * WPB + wave;
  if (idx >= count) return;
  double* buf = s_buf[wave];
  double xa[kRegs], xb[kRegs];
#pragma unroll
  for (int r = 0; r < kRegs; ++r) {
    xa[r] = (double)a_small[idx * kN + lane + 64 * r];
    xb[r] = (double)b_torus[idx * kN + lane + 64 * r];
  }
  ntt_forward<C>(lane, xb, s_tw, buf, f);
#pragma unroll
  for (int u = 0; u < kRegs; ++u) xb[u] = f_reduce(f_mulmod(f_reduce(xb[u], f), ninv, f), f);
  ntt_forward<C>(lane, xa, s_tw, buf, f);
#pragma unroll
  for (int u = 0; u < kRegs; ++u) xa[u] = f_mulmod(xa[u], xb[u], f);
  ntt_inverse<C>(lane, xa, s_tw + kN, buf, f);
#pragma unroll
  for (int r = 0; r < kRegs; ++r) out[idx * kN + lane + 64 * r] = f_to_torus32(xa[r]);
}

// -------------------------------------------------------------------------------------------------
// LWE word arithmetic
// -------------------------------------------------------------------------------------------------
__global__ void lincomb_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ x, int32_t cx, const int32_t* __restrict__ y,
                               int32_t cy, int32_t bconst, int W, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    uint32_t v = (uint32_t)cx * (uint32_t)x[e];
    if (y) v += (uint32_t)cy * (uint32_t)y[e];
    if ((int)(e % W) == W - 1) v += (uint32_t)bconst;
    out[e] = (int32_t)v;
  }
}

__global__ void gather_rows_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in, const int32_t* __restrict__ idx, int W,
                                   long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long row = e / W;
    const int w = (int)(e - row * W);
    const int src = idx[row];
    out[e] = src < 0 ? 0 : in[(size_t)src * W + w];
  }
}

// out[m][w] = sum_k s(k,m) in[k][w] (+ constants on the b word). grid (ceil(W/256), M).
__global__ __launch_bounds__(256) void linear_fc_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                        const uint8_t* __restrict__ sign, const uint8_t* __restrict__ zero, int K,
                                                        int M, int W, int32_t zero_tap_b, const int32_t* __restrict__ bias_b,
                                                        int bias_depth) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  if (w >= W) return;
  uint32_t acc = 0;
  uint32_t nzero = 0;
  for (int k = 0; k < K; ++k) {
    const size_t fi = (size_t)k * M + m;
    if (zero && zero[fi]) { ++nzero; continue; }
    const uint32_t v = (uint32_t)in[(size_t)k * W + w];
    acc += sign[fi] ? v : (0u - v);
  }
  if (w == W - 1) {
    acc += nzero * (uint32_t)zero_tap_b;
    if (bias_b) acc += (uint32_t)bias_b[m % bias_depth];
  }
  out[(size_t)m * W + w] = (int32_t)acc;
}

// out[oh][ow][od][w]; grid (ceil(W/256), Cout, Ho*Wo)
__global__ __launch_bounds__(256) void conv_ternary_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                           const uint8_t* __restrict__ sign, const uint8_t* __restrict__ zero,
                                                           ConvShape s, int W, int32_t zero_tap_b, int32_t pad_tap_b,
                                                           const int32_t* __restrict__ bias_b, int bias_depth) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int od = blockIdx.y;
  const int pix = blockIdx.z;
  if (w >= W) return;
  const int oh = pix / s.Wo, ow = pix % s.Wo;
  uint32_t acc = 0, nzero = 0, npad = 0;
  for (int fh = 0; fh < s.fh; ++fh) {
    const int ih = fh + oh * s.stride_h - s.off_h;
    for (int fw = 0; fw < s.fw; ++fw) {
      const int iw = fw + ow * s.stride_w - s.off_w;
      const bool oob = (unsigned)ih >= (unsigned)s.H || (unsigned)iw >= (unsigned)s.Wd;
      for (int di = 0; di < s.Cin; ++di) {
        const size_t fi = (((size_t)fh * s.fw + fw) * s.Cin + di) * s.Cout + od;
        if (!oob && !(zero && zero[fi])) {
          const uint32_t v = (uint32_t)in[(((size_t)ih * s.Wd + iw) * s.Cin + di) * W + w];
          acc += sign[fi] ? v : (0u - v);
        } else if (zero && zero[fi]) {
          ++nzero;  // reference order: ternary-zero test precedes the padding branch
        } else {
          ++npad;
        }
      }
    }
  }
  if (w == W - 1) {
    acc += nzero * (uint32_t)zero_tap_b + npad * (uint32_t)pad_tap_b;
    if (bias_b) acc += (uint32_t)bias_b[od % bias_depth];
  }
  out[(((size_t)oh * s.Wo + ow) * s.Cout + od) * W + w] = (int32_t)acc;
}

// grid (ceil(W/256), C, Ho*Wo)
__global__ __launch_bounds__(256) void sumpool_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in, PoolShape s, int W,
                                                      const int32_t* __restrict__ bias_b, int bias_depth) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  const int pix = blockIdx.z;
  if (w >= W) return;
  const int oh = pix / s.Wo, ow = pix % s.Wo;
  uint32_t acc = 0;
  for (int fh = 0; fh < s.win_h; ++fh) {
    const int ih = oh * s.stride_h - s.off_h + fh;
    if (ih < 0 || ih >= s.H) continue;
    for (int fw = 0; fw < s.win_w; ++fw) {
      const int iw = ow * s.stride_w - s.off_w + fw;
      if (iw < 0 || iw >= s.Wd) continue;
      acc += (uint32_t)in[(((size_t)ih * s.Wd + iw) * s.C + c) * W + w];
    }
  }
  if (w == W - 1 && bias_b) acc += (uint32_t)bias_b[c % bias_depth];
  out[(((size_t)oh * s.Wo + ow) * s.C + c) * W + w] = (int32_t)acc;
}

// -------------------------------------------------------------------------------------------------
// Launchers
// -------------------------------------------------------------------------------------------------
template <class C, int WPB>
static hipError_t launch_br(const BlindRotateArgs& a, long max_blocks, hipStream_t st) {
  long blocks = (a.B + WPB - 1) / WPB;
  BlindRotateArgs args = a;
  if (a.counter && blocks > max_blocks) {
    blocks = max_blocks;                       // persistent: one workgroup per CU, waves pull work
    hipError_t e = hipMemsetAsync(a.counter, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
  } else {
    args.counter = nullptr;                    // every wave has exactly one ciphertext
  }
  hipLaunchKernelGGL((blind_rotate_kernel<C, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, args);
  return hipGetLastError();
}

template <class C>
static hipError_t launch_br_cfg(const BlindRotateArgs& a, int wpb, long num_cus, hipStream_t st) {
  switch (wpb) {
    case 1: return launch_br<C, 1>(a, 1L << 40, st);
    case 2: return launch_br<C, 2>(a, 1L << 40, st);
    case 4: return launch_br<C, 4>(a, 1L << 40, st);
    default: return launch_br<C, 8>(a, num_cus, st);   // 152 KB LDS: exactly one workgroup per CU
  }
}

hipError_t launch_blind_rotate(int cfg, const BlindRotateArgs& a, int wpb, int num_cus, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
#if !defined(RS_STAMPS)
  // latency form: several waves per ciphertext while the batch cannot fill the chip by itself
  if (!getenv("RS_NO_COOP")) {
    if (cfg == 1 && a.B <= num_cus) {
      hipLaunchKernelGGL((blind_rotate_coop_kernel<CfgRedsecV2, 4>), dim3((unsigned)a.B), dim3(256), 0, st, a);
      return hipGetLastError();
    }
    if (a.B <= 2L * num_cus) {
      if (cfg == 0) hipLaunchKernelGGL((blind_rotate_coop_kernel<CfgDefault128, 2>), dim3((unsigned)a.B), dim3(128), 0, st, a);
      else hipLaunchKernelGGL((blind_rotate_coop_kernel<CfgRedsecV2, 2>), dim3((unsigned)a.B), dim3(128), 0, st, a);
      return hipGetLastError();
    }
  }
#endif
  return cfg == 0 ? launch_br_cfg<CfgDefault128>(a, wpb, num_cus, st) : launch_br_cfg<CfgRedsecV2>(a, wpb, num_cus, st);
}

hipError_t launch_bk_transform(int cfg, const int32_t* bk, double* bk_ntt, const double* tw, Field f, double ninv, long n_polys,
                               hipStream_t st) {
  constexpr int WPB = 4;
  const long blocks = (n_polys + WPB - 1) / WPB;
  if (cfg == 0)
    hipLaunchKernelGGL((bk_transform_kernel<CfgDefault128, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, bk, bk_ntt, tw, f, ninv, n_polys);
  else
    hipLaunchKernelGGL((bk_transform_kernel<CfgRedsecV2, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, bk, bk_ntt, tw, f, ninv, n_polys);
  return hipGetLastError();
}

hipError_t launch_keyswitch(const KeyswitchArgs& a, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  dim3 grid((unsigned)((a.B + KS_TILE_THREADS - 1) / KS_TILE_THREADS), (unsigned)((a.W + KS_CH - 1) / KS_CH), 1);
  const bool tiled = (a.t == 8 && a.basebit == 2) || (a.t == 9 && a.basebit == 3);
  if (tiled) {
    // small batches: slice the input coefficients until ~1024 workgroups exist (latency form)
    unsigned split = 1;
    while (split < 64 && grid.x * grid.y * split < 1024) split *= 2;
    if (split > 1) {
      hipError_t e = hipMemsetAsync(a.out, 0, (size_t)a.B * a.W * sizeof(int32_t), st);
      if (e != hipSuccess) return e;
      grid.z = split;
    }
  }
  if (a.t == 8 && a.basebit == 2) {
    hipLaunchKernelGGL((keyswitch_tiled_kernel<8, 2, 4>), grid, dim3(KS_TILE_THREADS), 0, st, a);
  } else if (a.t == 9 && a.basebit == 3) {
    hipLaunchKernelGGL((keyswitch_tiled_kernel<9, 3, 2>), grid, dim3(KS_TILE_THREADS), 0, st, a);
  } else {
    hipLaunchKernelGGL(keyswitch_kernel, dim3((unsigned)a.B), dim3(KS_THREADS), 0, st, a);  // generic gather form
  }
  return hipGetLastError();
}

hipError_t launch_polymul(int cfg, const int32_t* a_small, const int32_t* b_torus, int32_t* out, const double* tw, Field f,
                          double ninv, long count, hipStream_t st) {
  constexpr int WPB = 4;
  const long blocks = (count + WPB - 1) / WPB;
  if (cfg == 0)
    hipLaunchKernelGGL((polymul_kernel<CfgDefault128, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, a_small, b_torus, out, tw, f, ninv, count);
  else
    hipLaunchKernelGGL((polymul_kernel<CfgRedsecV2, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, a_small, b_torus, out, tw, f, ninv, count);
  return hipGetLastError();
}

hipError_t launch_lincomb(int32_t* out, const int32_t* x, int32_t cx, const int32_t* y, int32_t cy, int32_t bconst, int W, long B,
                          hipStream_t st) {
  const long total = B * W;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, x, cx, y, cy, bconst, W, total);
  return hipGetLastError();
}

hipError_t launch_gather_rows(int32_t* out, const int32_t* in, const int32_t* idx, int W, long B, hipStream_t st) {
  const long total = B * W;
  if (total <= 0) return hipSuccess;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, in, idx, W, total);
  return hipGetLastError();
}

hipError_t launch_linear_fc(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, int K, int M, int W,
                            int32_t zero_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st) {
  hipLaunchKernelGGL(linear_fc_kernel, dim3((W + 255) / 256, M), dim3(256), 0, st, out, in, sign, zero, K, M, W, zero_tap_b, bias_b,
                     bias_depth);
  return hipGetLastError();
}

hipError_t launch_conv_ternary(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, const ConvShape& s, int W,
                               int32_t zero_tap_b, int32_t pad_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st) {
  hipLaunchKernelGGL(conv_ternary_kernel, dim3((W + 255) / 256, s.Cout, s.Ho * s.Wo), dim3(256), 0, st, out, in, sign, zero, s, W,
                     zero_tap_b, pad_tap_b, bias_b, bias_depth);
  return hipGetLastError();
}

hipError_t launch_sumpool(int32_t* out, const int32_t* in, const PoolShape& s, int W, const int32_t* bias_b, int bias_depth,
                          hipStream_t st) {
  hipLaunchKernelGGL(sumpool_kernel, dim3((W + 255) / 256, s.C, s.Ho * s.Wo), dim3(256), 0, st, out, in, s, W, bias_b, bias_depth);
  return hipGetLastError();
}

}  // namespace rs
