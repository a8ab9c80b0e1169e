// rs_bootstrap.hip -- the transform-based kernels of the gate bootstrap, written once over a
// "transform policy":
//   XfNtt<Cfg>  exact negacyclic NTT over a 51-bit prime carried in FP64 (rs_ntt.h)  -- guaranteed exact
//   XfFft<Cfg>  folded 512-point complex FP64 FFT (rs_fft.h), TFHE's own arithmetic class -- 2.5x fewer
//               FP64 ops; exact after rounding with overwhelming probability, with a run-time certificate
//
//   bk_transform_kernel       bootstrapping key -> transform domain (the bkFFT analogue; once per key)
//   blind_rotate_kernel       gate pre-combination + modswitch + n CMUX steps + sample extract
//                             (tfhe_bootstrap_woKS_FFT; REDsec: lib/BinOps_enc.cpp:185,191), throughput form
//   blind_rotate_coop_kernel  the same with G waves per ciphertext (latency form for small batches)
//   polymul_kernel            debug/parity tap through the same transform path
//
// One wavefront owns one ciphertext for the whole blind rotation: its TRLWE accumulator (2 x 1024
// int32) lives in LDS, each of the (k+1) l digit polynomials is transformed in registers with two
// LDS transposes, multiplied against the coalesced-streamed key row and accumulated in registers,
// and two inverse transforms update the accumulator. Waves never synchronise with each other after
// the twiddle tables are staged.
//
// Compile-time switches of this file: RS_BS_PART (below: which launchers an object holds) and RS_DIAG (rs_diag.h: phase stamps
// and the no-key timing probe of diagnostic builds). Every experiment of rounds 1-4 that was measured and not adopted has its
// verdict in MEASUREMENTS.md and no code path here.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "rs_fft.h"
#include "rs_cohort.h"
#include "rs_diag.h"
#include "rs_kernels.h"
#include "rs_lds_plan.h"
#include "rs_ntt.h"

// RS_BS_PART: redsec_amd/build.py compiles this file THREE times, because its kernels want different code-generation flags
// (profiles/r03/y_ab_compiler_scheduling_*.txt). Bit 1 = every launcher except the one of bit 2 (built with LLVM's post-RA
// scheduler off: the FFT / exact-NTT kernels and the split duo form gain 1-3 % from it); bit 2 = launch_blind_rotate_split_wg
// with the split cooperative and split lock-step kernels (built with the default pipeline: they lose 6 % / 0.7 % without that
// pass); bit 4 = launch_coop8_listed with blind_rotate_coop8_listed_kernel (round 6; part 1's flags; apart so that part 1's
// device code stays what it was, see there). Kernels are templates, so each object holds only what its launchers name.
// Default 7: one object with everything.
#ifndef RS_BS_PART
#define RS_BS_PART 7
#endif

namespace rs {

// Same-wave LDS hand-off: DS operations of one wavefront execute in order, so only the compiler
// needs to be told not to move LDS accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void stage_tables(double* s_tw, const double* tw_g, int nthreads, int count) {
  for (int i = threadIdx.x; i < count; i += nthreads) s_tw[i] = tw_g[i];
  __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// Transform policies
// -------------------------------------------------------------------------------------------------
template <class C>
struct XfNtt {
  using Cfg = C;
  static constexpr int kTableDoubles = kTwTotal;   // staged in LDS
  static constexpr bool kCertificate = false;
  static constexpr bool kSplitKeyLoads = false;   // whole key row prefetched across the transform
  static constexpr bool kWorkgroupForm = false;
  static constexpr bool kPreparedDigits = false;  // digits extracted from the raw rotated difference
  struct State { const double* tw; };
  __device__ static __forceinline__ void init(State& st, int, const double* tw_lds, const double*) { st.tw = tw_lds; }
  using LatencyState = State;   // the cooperative kernel's twiddle source (XfFft keeps its per-lane twiddles in registers there)
  __device__ static __forceinline__ void init_latency(LatencyState& st, int lane, const double* tw_lds, const double* tw_g) { init(st, lane, tw_lds, tw_g); }

  __device__ static __forceinline__ void fwd_digits(int lane, double (&x)[kRegs], const int32_t (&d)[kRegs], int q, uint32_t offset,
                                                    const State& st, double* buf, const Field& f) {
    const double* tw = st.tw;
    fwd_F1_digits<C>(lane, x, d, q, offset, tw, buf, f);
    wave_lds_sync();
    fwd_F2<C>(lane, x, tw, buf, f);
    wave_lds_sync();
    fwd_F3(lane, x, buf);
    wave_lds_sync();
    fwd_F4<C>(lane, x, tw, buf, f);
    wave_lds_sync();
  }
  __device__ static __forceinline__ void fwd_generic(int lane, double (&x)[kRegs], const State& st, double* buf, const Field& f) {
    const double* tw = st.tw;
    fwd_F1<C>(lane, x, tw, buf, f);
    wave_lds_sync();
    fwd_F2<C>(lane, x, tw, buf, f);
    wave_lds_sync();
    fwd_F3(lane, x, buf);
    wave_lds_sync();
    fwd_F4<C>(lane, x, tw, buf, f);
    wave_lds_sync();
  }
  // key values: scaled by 1/N and fully reduced; stored as pairs (positions 16 lane + 2v, +1)
  __device__ static __forceinline__ void key_store(double2* dst, int lane, const double (&x)[kRegs], double scale, const Field& f) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const double a = f_reduce(f_mulmod(f_reduce(x[2 * v], f), scale, f), f);
      const double b = f_reduce(f_mulmod(f_reduce(x[2 * v + 1], f), scale, f), f);
      dst[v * 64 + lane] = make_double2(a, b);
    }
  }
  // multiply-accumulate against key entries v0 .. v0+3 of both columns
  __device__ static __forceinline__ void mac(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs],
                                             const double2 (&w0)[4], const double2 (&w1)[4], int v0, const Field& f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = v0 + k;
      s0[2 * v] += f_mulmod(x[2 * v], w0[k].x, f);
      s0[2 * v + 1] += f_mulmod(x[2 * v + 1], w0[k].y, f);
      s1[2 * v] += f_mulmod(x[2 * v], w1[k].x, f);
      s1[2 * v + 1] += f_mulmod(x[2 * v + 1], w1[k].y, f);
    }
  }
  __device__ static __forceinline__ void mac8(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs],
                                              const double2 (&w0)[8], const double2 (&w1)[8], const Field& f) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      s0[2 * v] += f_mulmod(x[2 * v], w0[v].x, f);
      s0[2 * v + 1] += f_mulmod(x[2 * v + 1], w0[v].y, f);
      s1[2 * v] += f_mulmod(x[2 * v], w1[v].x, f);
      s1[2 * v + 1] += f_mulmod(x[2 * v + 1], w1[v].y, f);
    }
  }
  __device__ static __forceinline__ void mid(double (&s0)[kRegs], double (&s1)[kRegs], const Field& f) {
    if (C::MID_REDUCE) {
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { s0[u] = f_reduce(s0[u], f); s1[u] = f_reduce(s1[u], f); }
    }
  }
  __device__ static __forceinline__ double partial(double v, const Field& f) { return f_reduce(v, f); }
  __device__ static __forceinline__ void inverse(int lane, double (&x)[kRegs], const State& st, double* buf, const Field& f) {
    const double* twi = st.tw + kN;
    inv_I1<C>(lane, x, twi, buf, f);
    wave_lds_sync();
    inv_I2<C>(lane, x, twi, buf, f);
    wave_lds_sync();
    inv_I3(lane, x, buf);
    wave_lds_sync();
    inv_I4<C>(lane, x, twi, buf, f);
    wave_lds_sync();
  }
  __device__ static __forceinline__ void inverse2(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const State& st, double* buf, const Field& f) {
    inverse(lane, xa, st, buf, f);
    inverse(lane, xb, st, buf, f);
  }
  __device__ static __forceinline__ int32_t to_torus(double v, double&) { return f_to_torus32(v); }
};

template <class C>
struct XfFft {
  using Cfg = C;
  static constexpr int kTableDoubles = kFftTwDoubles;   // stage-transposed complex table staged in LDS (8 KB)
  static constexpr bool kCertificate = true;
  static constexpr bool kSplitKeyLoads = true;    // second half of the key row fetched after the transform
  static constexpr bool kWorkgroupForm = true;    // blind_rotate_wg_kernel available
  static constexpr bool kPreparedDigits = true;   // d[] = gadget_prepare(rotated difference): one v_bfe_i32 per digit
  // Twiddles are read from the LDS table at every use: keeping the 21 complex values of a lane in
  // registers (FftTw) spilled 250 B/lane to scratch at the 256-VGPR budget and cost 40 % (scratch
  // reloads share vmcnt with the in-flight key-row loads).
  using State = FftTwTable;
  __device__ static __forceinline__ void init(State& st, int lane, const double* tw_lds, const double*) { st.tw = tw_lds; st.lane = lane; }

  // One wave per SIMD in the cooperative kernel (512 registers): all eight per-lane twiddles stay in registers
  using LatencyState = FftTwKept<3>;
  __device__ static __forceinline__ void init_latency(LatencyState& st, int lane, const double* tw_lds, const double* tw_g) {
    State t;
    init(t, lane, tw_lds, tw_g);
    fft_kept_load(st, t);
  }
  template <class TWS>
  __device__ static __forceinline__ void fwd_generic(int lane, double (&x)[kRegs], const TWS& st, double* buf, const Field&) {
    ffwd_F1(lane, x, st, buf);
    wave_lds_sync();
    ffwd_F2(lane, x, st, buf);
    wave_lds_sync();
    ffwd_F3(lane, x, buf);
    wave_lds_sync();
    ffwd_F4(lane, x, st, buf);
    wave_lds_sync();
  }
  template <class TWS>
  __device__ static __forceinline__ void fwd_digits(int lane, double (&x)[kRegs], const int32_t (&d)[kRegs], int q, uint32_t offset,
                                                    const TWS& st, double* buf, const Field& f) {
#pragma unroll
    for (int r = 0; r < kRegs; ++r) x[r] = (double)gadget_digit_prepared<C>(d[r], q);
    fwd_generic(lane, x, st, buf, f);
  }
  // key values scaled by 1/M (exact power of two); stored as (re, im) of position 8 lane + v
  __device__ static __forceinline__ void key_store(double2* dst, int lane, const double (&x)[kRegs], double, const Field&) {
#pragma unroll
    for (int v = 0; v < 8; ++v) dst[v * 64 + lane] = make_double2(x[v] * (1.0 / kM), x[v + 8] * (1.0 / kM));
  }
  __device__ static __forceinline__ void mac(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs],
                                             const double2 (&w0)[4], const double2 (&w1)[4], int v0, const Field&) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = v0 + k;
      fft_cmac(s0[v], s0[v + 8], x[v], x[v + 8], w0[k].x, w0[k].y);
      fft_cmac(s1[v], s1[v + 8], x[v], x[v + 8], w1[k].x, w1[k].y);
    }
  }
  __device__ static __forceinline__ void mac8(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs],
                                              const double2 (&w0)[8], const double2 (&w1)[8], const Field&) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      fft_cmac(s0[v], s0[v + 8], x[v], x[v + 8], w0[v].x, w0[v].y);
      fft_cmac(s1[v], s1[v + 8], x[v], x[v + 8], w1[v].x, w1[v].y);
    }
  }
  __device__ static __forceinline__ void mid(double (&)[kRegs], double (&)[kRegs], const Field&) {}
  __device__ static __forceinline__ double partial(double v, const Field&) { return v; }
  template <class TWS>
  __device__ static __forceinline__ void inverse(int lane, double (&x)[kRegs], const TWS& st, double* buf, const Field&) {
    finv_I1(lane, x, st, buf);
    wave_lds_sync();
    finv_I2(lane, x, st, buf);
    wave_lds_sync();
    finv_I3(lane, x, buf);
    wave_lds_sync();
    finv_I4(lane, x, st, buf);
    wave_lds_sync();
  }
  // both accumulator columns at once: the two inverse transforms interleaved phase by phase
  __device__ static __forceinline__ void inverse2(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const State& st, double* buf, const Field&) {
    finv_pair<false>(lane, xa, xb, st, buf, [] { wave_lds_sync(); });
  }
  __device__ static __forceinline__ int32_t to_torus(double v, double& dev) { return fft_round_torus32(v, dev); }

  // workgroup kernel: planar exchanges through a half-size per-wave buffer
  static constexpr int kWgBufDoubles = kPlaneDoubles;
  __device__ static __forceinline__ void inverse_wg(int lane, double (&x)[kRegs], const State& st, double* buf, const Field&) {
    finv_planar(lane, x, st, buf, [] { wave_lds_sync(); });
  }
  __device__ static __forceinline__ void digits(double (&x)[kRegs], const int32_t (&d)[kRegs], int q) {
#pragma unroll
    for (int r = 0; r < kRegs; ++r) x[r] = (double)gadget_digit_prepared<C>(d[r], q);
  }
  // the lock-step workgroup kernel's pairs: per-lane twiddles from registers (FftTwKept) or the LDS tables (State)
  template <class TWS>
  __device__ static __forceinline__ void fwd_pair_wg(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const TWS& st, double* buf) {
    ffwd_pair<true>(lane, xa, xb, st, buf, [] { wave_lds_sync(); });
  }
  template <class TWS>
  __device__ static __forceinline__ void inverse_pair_wg(int lane, double (&xa)[kRegs], double (&xb)[kRegs], const TWS& st, double* buf) {
    finv_pair<true>(lane, xa, xb, st, buf, [] { wave_lds_sync(); });
  }
  static constexpr int kWgTableDoubles = kFftTwDoubles;
};

// largest rounding distance of the wave -> device flag (positive doubles order like their bit patterns)
__device__ __forceinline__ void publish_certificate(double dev, unsigned long long* flag, int lane) {
  if (!flag) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(dev, off, 64);
    dev = o > dev ? o : dev;
  }
  if (lane == 0) atomicMax(flag, (unsigned long long)__double_as_longlong(dev));
}

// Initial accumulator body: (test polynomial * X^rot)_j. Constant test vector (tfhe_bootstrap_woKS_FFT) or, in
// the programmable form, the ciphertext's own polynomial lut[ct % lut_count] (tfhe_blindRotateAndExtract_FFT).
__device__ __forceinline__ int32_t test_vector(const BlindRotateArgs& a, long ct, int j, int rot) {
  if (!a.lut) return rotated_const(a.mu, j, rot);
  const int32_t* v = a.lut + (size_t)((ct + a.lut_first) % a.lut_count) * kN;
  const int aa = rot & (kN - 1), nb = (rot >> 10) & 1;
  const uint32_t x = (uint32_t)v[(j - aa) & (kN - 1)];
  return (int32_t)((((j < aa) ? 1 : 0) ^ nb) ? 0u - x : x);
}

// Exact recomputation gate (see BlindRotateArgs::gate_flag). Uniform over the grid: every thread reads the
// same word, so whole workgroups leave before their first barrier. Returns true when the launch has nothing to do.
__device__ __forceinline__ bool recompute_not_needed(const BlindRotateArgs& a) {
  if (!a.gate_flag) return false;
  const unsigned long long bits = *(const volatile unsigned long long*)a.gate_flag;   // positive doubles order like their bit patterns
  const bool needed = bits >= a.gate_limit_bits;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.running_flag) atomicMax(a.running_flag, bits);
    if (needed && a.fallback_count) atomicAdd(a.fallback_count, 1ull);
  }
  return !needed;
}

// -------------------------------------------------------------------------------------------------
// Key transform: one wavefront per key polynomial. Output layout per polynomial: [v 0..7][lane][2]
// doubles in the exact register order of the consuming wavefront, so the blind rotation reads each
// row with eight perfectly coalesced 16-byte-per-lane loads.
// -------------------------------------------------------------------------------------------------
template <class Xf, int WPB>
__global__ __launch_bounds__(64 * WPB) void bk_transform_kernel(const int32_t* __restrict__ bk, double* __restrict__ bk_x,
                                                                 const double* __restrict__ tw_g, Field f, double scale, long n_polys) {
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[WPB][kBufDoubles];
  stage_tables(s_tw, tw_g, 64 * WPB, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long poly = (long)blockIdx.x * WPB + wave;
  if (poly >= n_polys) return;
  typename Xf::State st;
  Xf::init(st, lane, s_tw, tw_g);
  double x[kRegs];
  const int32_t* src = bk + poly * kN;
#pragma unroll
  for (int r = 0; r < kRegs; ++r) x[r] = (double)src[lane + 64 * r];
  Xf::fwd_generic(lane, x, st, s_buf[wave], f);
  Xf::key_store(reinterpret_cast<double2*>(bk_x + poly * kN), lane, x, scale, f);
}

// -------------------------------------------------------------------------------------------------
// Blind rotation + sample extract, throughput form (persistent waves).
// -------------------------------------------------------------------------------------------------
template <class Xf, int WPB>
__global__ __launch_bounds__(64 * WPB) void blind_rotate_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  if (recompute_not_needed(a)) return;
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[WPB][kBufDoubles];
  __shared__ int32_t s_acc[WPB][2][kN];
  stage_tables(s_tw, a.tw, 64 * WPB, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  // Persistent waves: the first ciphertext is assigned statically, further ones are pulled from a
  // device counter (zeroed by the launcher on the same stream). A 150 KB-LDS workgroup cannot be
  // replaced until its LAST wave exits, and waves sharing a SIMD finish up to 20 % apart (issue
  // arbitration favours the older wave), which left 18 % of the wave slots idle with one ciphertext
  // per wave. Every wave leaves the loop as soon as the counter passes B, so the grid always drains.
  long ct = (long)blockIdx.x * WPB + wave;
  const long first_dynamic = (long)gridDim.x * WPB;
  if (ct >= a.B) return;

  const Field f = a.f;
  double* buf = s_buf[wave];
  int32_t* acc0 = s_acc[wave][0];
  int32_t* acc1 = s_acc[wave][1];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  const int n = a.n;
  constexpr uint32_t offset = gadget_offset<C>();
  constexpr int KPL = 2 * C::L;
  double dev = 0.0;

  for (;;) {
    const int32_t* row0 = a.in0 + ct * a.W;
    const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
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
        acc1[j] = test_vector(a, ct, j, rot);
      }
    }
    wave_lds_sync();

    for (int i = 0; i < n; ++i) {
      const int32_t bara = __builtin_amdgcn_readfirstlane(modswitch_2N(word(i)));
      if (bara == 0) continue;  // tfhe_blindRotate_FFT skips the identity CMUX
      double s0[kRegs], s1[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
      const double* bk_i = a.bk_x + (size_t)i * KPL * 2 * kN;

#pragma unroll 1
      for (int comp = 0; comp < 2; ++comp) {
        const int32_t* accc = comp ? acc1 : acc0;
        int32_t d[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          d[r] = rotated_diff(accc, lane + 64 * r, bara);
          if (Xf::kPreparedDigits) d[r] = gadget_prepare<C>(d[r]);
        }
#pragma unroll 1
        for (int q = 0; q < C::L; ++q) {
          const int row = comp * C::L + q;
          const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)(row * 2) * kN);
          const double2* bp1 = bp0 + kN / 2;
          double x[kRegs];
          if constexpr (Xf::kSplitKeyLoads) {
            // first half of the key row prefetched across the transform, second half fetched after it
            double2 wa0[4], wa1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { wa0[v] = bp0[v * 64 + lane]; wa1[v] = bp1[v * 64 + lane]; }
            Xf::fwd_digits(lane, x, d, q, offset, tw, buf, f);
            double2 wb0[4], wb1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { wb0[v] = bp0[(v + 4) * 64 + lane]; wb1[v] = bp1[(v + 4) * 64 + lane]; }
            Xf::mac(s0, s1, x, wa0, wa1, 0, f);
            Xf::mac(s0, s1, x, wb0, wb1, 4, f);
          } else {
            double2 w0[8], w1[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
            Xf::fwd_digits(lane, x, d, q, offset, tw, buf, f);
            Xf::mac8(s0, s1, x, w0, w1, f);
          }
        }
        if (comp == 0) Xf::mid(s0, s1, f);
      }

      Xf::inverse2(lane, s0, s1, tw, buf, f);
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc0[j] = (int32_t)((uint32_t)acc0[j] + (uint32_t)Xf::to_torus(s0[r], dev));
        acc1[j] = (int32_t)((uint32_t)acc1[j] + (uint32_t)Xf::to_torus(s1[r], dev));
      }
      wave_lds_sync();
    }

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
  if (Xf::kCertificate) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Blind rotation, workgroup lock-step form (large batches). The per-wave kernel above makes EVERY
// wavefront stream the whole transformed key by itself: at 150k bootstraps/s that is ~9.6 TB/s of
// L2 -> CU reads with a 64 % L2 miss rate (waves drift apart, so a key row is rarely still in L2 for
// the next wave), and both parameter sets stall at that same byte rate. Here the WPB waves of a
// workgroup walk the CMUX chain of WPB different ciphertexts in lock step and share each key row
// through LDS: every wave fetches 1/WPB of the row with direct global->LDS loads (no VGPRs), one
// workgroup barrier per row publishes it, and the multiply-accumulate reads it with 16-byte LDS
// reads. Rows go in pairs through a two-slot ring (see the loop) -- two barriers per pair.
// Every wave executes every barrier: inactive waves (ragged last group) and identity CMUX steps
// (bara == 0) only skip the arithmetic. Groups are assigned round-robin: all groups take the same
// number of steps, so there is nothing to balance dynamically.
// -------------------------------------------------------------------------------------------------
// Direct global -> LDS loads, 16 bytes per lane = 1 KB per wave-instruction, NCHUNK consecutive KB:
// global address = wave-uniform base (SGPR pair) + lane_off (one VGPR, lane * 16) + k KB; LDS address =
// M0 + k KB + lane * 16 (the instruction offset advances both sides). Written as asm because (a) hipcc
// puts a vmcnt(0) in front of the next LDS read whenever it knows of a pending LDS-DMA, which would
// serialise the prefetch -- the kernels' own `s_waitcnt vmcnt(0)` + barrier orders the data instead;
// (b) with per-lane 64-bit source pointers the compiler spilled around the issue point, and every
// scratch reload there waits on vmcnt, i.e. on the key rows that were just requested.
template <int NCHUNK>
__device__ __forceinline__ void glds_chunks(const double* gsrc_wave_base, unsigned lane_off, const double* lds_wave_base) {
  static_assert(NCHUNK >= 1 && NCHUNK <= 4, "instruction offsets are 13-bit signed");
  const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_wave_base);
  const unsigned long long base = (unsigned long long)(uintptr_t)gsrc_wave_base;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
  const unsigned long long sbase = ((unsigned long long)hi << 32) | lo;
  unsigned keep;
  if constexpr (NCHUNK == 1) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_dst) : "memory");
  } else if constexpr (NCHUNK == 2) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_dst) : "memory");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sbase), "s"(lds_dst) : "memory");
    static_assert(NCHUNK == 4, "1, 2 or 4 chunks");
  }
}

// The same with the LDS side given as a byte address (a workgroup-uniform unsigned: no generic-pointer cast, whose null check
// costs three scalar instructions per call) and the global side as a wave-uniform pointer the caller keeps running.
template <int NCHUNK>
__device__ __forceinline__ void glds_chunks_at(const double* gsrc_wave_base, unsigned lane_off, unsigned lds_byte_addr) {
  static_assert(NCHUNK == 1 || NCHUNK == 2 || NCHUNK == 4, "1, 2 or 4 chunks of 1 KB per wave");
  unsigned keep;
  if constexpr (NCHUNK == 1) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(gsrc_wave_base), "s"(lds_byte_addr) : "memory");
  } else if constexpr (NCHUNK == 2) {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(gsrc_wave_base), "s"(lds_byte_addr) : "memory");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(gsrc_wave_base), "s"(lds_byte_addr) : "memory");
  }
}

// Pointwise multiply-accumulate of a transform PAIR against its two key rows in LDS (FFT policies), as one
// stream: 8 steps of two complex positions (4 ds_read_b128: both columns), the reads of step k+1 issued
// before the FMAs of step k. Read in four blocks of 8 with the FMAs after each block (mac_row), every
// block exposed a fresh LDS latency because the FMAs of a block drain its reads (in-order return).
#define RS_MAC_FENCE() __builtin_amdgcn_sched_barrier(0)
// col0 / col1: offsets (in double2) of the column multiplied into s0 / s1 within a key row -- 0 and kN / 2 for (column 0, column 1);
// the duo kernel passes them swapped for its odd waves, so that s0 is always the column the wave itself inverts
__device__ __forceinline__ void mac_pair_stream(double (&s0)[kRegs], double (&s1)[kRegs], const double (&xa)[kRegs], const double (&xb)[kRegs],
                                                const double* keyA, const double* keyB, int lane, int col0 = 0, int col1 = kN / 2) {
  const double2* ka = reinterpret_cast<const double2*>(keyA);
  const double2* kb = reinterpret_cast<const double2*>(keyB);
  double2 u[2][4];
  auto issue = [&](int step, double2 (&w)[4]) {
    const double2* k0 = (step < 4 ? ka : kb) + col0;
    const double2* k1 = (step < 4 ? ka : kb) + col1;
    const int v = 2 * (step & 3);
    w[0] = k0[v * 64 + lane]; w[1] = k0[(v + 1) * 64 + lane];
    w[2] = k1[v * 64 + lane]; w[3] = k1[(v + 1) * 64 + lane];
  };
  auto fma = [&](int step, const double2 (&w)[4]) {
    const double (&x)[kRegs] = step < 4 ? xa : xb;
    const int v = 2 * (step & 3);
    fft_cmac(s0[v], s0[v + 8], x[v], x[v + 8], w[0].x, w[0].y);
    fft_cmac(s0[v + 1], s0[v + 9], x[v + 1], x[v + 9], w[1].x, w[1].y);
    fft_cmac(s1[v], s1[v + 8], x[v], x[v + 8], w[2].x, w[2].y);
    fft_cmac(s1[v + 1], s1[v + 9], x[v + 1], x[v + 9], w[3].x, w[3].y);
  };
  issue(0, u[0]);
#pragma unroll
  for (int step = 0; step < 8; ++step) {
    if (step + 1 < 8) issue(step + 1, u[(step + 1) & 1]);
    RS_MAC_FENCE();
    fma(step, u[step & 1]);
    RS_MAC_FENCE();
  }
}

#if RS_BS_PART & RS_DIAG_STAMP_PART   // diagnostic builds only (rs_diag.h): the phase sums, [workgroup < 256][wave][phase]
__device__ unsigned long long g_rs_stamps[256 * 8 * diag::kStampPhases];
#endif

template <class Xf, int WPB>
__global__ __launch_bounds__(64 * WPB) void blind_rotate_wg_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  constexpr int KPL = 2 * C::L;
  constexpr int kRowDoubles = 2 * kN;             // one key row: 2 columns x N doubles = 16 KB
  constexpr int kChunks = kRowDoubles / 128;      // 1 KB pieces = one wave-wide 16-byte load each
  constexpr int kChunksPerWave = kChunks / WPB;
  static_assert(kChunks % WPB == 0 && (kChunksPerWave == 1 || kChunksPerWave == 2 || kChunksPerWave == 4), "waves must split a key row evenly");
  __shared__ double s_tw[Xf::kWgTableDoubles + 1];
  __shared__ __attribute__((aligned(16))) double s_buf[WPB][Xf::kWgBufDoubles];
  __shared__ int32_t s_acc[WPB][2][kN];
  __shared__ __attribute__((aligned(16))) double s_key[2][kRowDoubles];
  constexpr int kWin = 64;                       // mask words live in a 64-step window, as in the split kernel
  __shared__ uint16_t s_bara[WPB][kWin];
  __shared__ int s_mail[kCohortSlots];   // XCD cohorts: the progress row requested a step ago (wave 0 only; rs_cohort.h)
  stage_tables(s_tw, a.tw, 64 * WPB, Xf::kWgTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const Field f = a.f;
  double* buf = s_buf[wave];
  int32_t* acc0 = s_acc[wave][0];
  int32_t* acc1 = s_acc[wave][1];
  typename Xf::State tw_table;
  Xf::init(tw_table, lane, s_tw, a.tw);
  FftTwKept<3> tw;   // all eight per-lane twiddles stay in registers for the whole kernel (32 registers; +2.4 % / +1.3 %, profiles/r03/n_*)
  fft_kept_load(tw, tw_table);
  const int n = a.n;
  constexpr uint32_t offset = gadget_offset<C>();
  double dev = 0.0;
  const long n_groups = (a.B + WPB - 1) / WPB;
  const int total_rows = n * KPL;   // key rows of one blind rotation (n <= 1024 steps x 2 l: far inside an int; scalar compares)
  RS_STAMP_DECL;   // -DRS_DIAG=1 (tools/stamp_profile.py): 0 step prologue, 1 digits + forward pair, 2 wait for the key rows + barrier, 3 multiply-
                   // accumulate, 4 barrier + next rows requested, 5 accumulator pre-read + inverse pair, 6 rounding + accumulator update, 7 group prologue / extract
  const unsigned lane_off = (unsigned)lane * 16u;   // my 1/WPB share of a key row: chunks of 1 KB, 16 bytes per lane

  int steps_done = 0;   // CMUX steps of the groups this workgroup has finished (XCD cohorts, rs_cohort.h)
  for (long group = blockIdx.x; group < n_groups; group += gridDim.x, steps_done += n) {
    const long ct = group * WPB + wave;
    const bool active = ct < a.B;
    const int32_t* row0 = a.in0 + (active ? ct : 0) * a.W;
    const int32_t* row1 = a.in1 ? a.in1 + (active ? ct : 0) * a.W : nullptr;
    auto word = [&](int i) -> int32_t {
      uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
      if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
      return (int32_t)v;
    };
    auto fill_window = [&](int i0) {   // bara of steps [i0, i0 + 64): only this wave reads its row
      const int i = i0 + lane;
      s_bara[wave][lane] = (active && i < n) ? (uint16_t)modswitch_2N(word(i)) : (uint16_t)0;
    };
    fill_window(0);
    if (active) {
      const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
      const int rot = 2 * kN - barb;  // in (0, 2N]
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc0[j] = 0;
        acc1[j] = test_vector(a, ct, j, rot);
      }
    }
    // all global reads of this prologue are complete and every wave has left the previous group's
    // last multiply-accumulate before the ring is refilled
    __syncthreads();
    // the pairs are requested in storage order, row R into slot 0 and R + 1 into slot 1: a running pointer and two fixed LDS
    // byte addresses instead of a 64-bit row index, its compare, a multiply-add and a generic-pointer cast per row (the same
    // change took the split lock-step kernel from 137 k to 145 k/s, profiles/r04/az_*)
    const double* src_next = a.bk_x + (size_t)(wave * kChunksPerWave) * 128;
    const unsigned key_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&s_key[0][0]) +
                             (unsigned)(wave * kChunksPerWave) * 1024u;
    auto issue_pair = [&] {
      glds_chunks_at<kChunksPerWave>(src_next, lane_off, key_lds);
      glds_chunks_at<kChunksPerWave>(src_next + kRowDoubles, lane_off, key_lds + (unsigned)(kRowDoubles * sizeof(double)));
      src_next += 2 * kRowDoubles;
    };
    issue_pair();
    RS_STAMP(7);

    // Rows are processed in PAIRS (R, R+1), the two digit transforms interleaved phase by phase.
    // Barrier 1 of a pair publishes both rows (each wave first waits for its own shares); barrier 2
    // says every wave has finished reading them, after which the next pair's loads are issued and
    // have the whole next transform pair to land.
    int R = 0;
    unsigned bara_next = s_bara[wave][0];   // read one step ahead: its LDS latency is not exposed
    for (int i = 0; i < n; ++i) {
      // XCD cohorts (cohort_step above): with l = 10 a step reads 320 KB of key per CU and an XCD's L2 keeps 12 steps; launches of the REDsec
      // set were seen at twice the 8-XCD floor of fabric traffic (60.8 GB, profiles/r04/pmc) when workgroups drifted further apart
      if (wave == 0) cohort_step<BlindRotateArgs>(steps_done + i, s_mail);
      const int32_t bara = __builtin_amdgcn_readfirstlane((int)bara_next);
      if (((i + 1) & (kWin - 1)) == 0 && i + 1 < n) { wave_lds_sync(); fill_window(i + 1); wave_lds_sync(); }
      bara_next = (i + 1 < n) ? s_bara[wave][(i + 1) & (kWin - 1)] : 0;
      const bool work = bara != 0;   // tfhe_blindRotate_FFT skips the identity CMUX
      double s0[kRegs], s1[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
      int32_t d[kRegs];
      RS_STAMP(0);

      // d holds the PREPARED rotated difference of one component (gadget offset added and field sign bits
      // flipped once per component, not per digit row). Written as straight-line code with a
      // compile-time component: inside the rolled pair loop (run-time component) the compiler issued
      // its 32 LDS reads one at a time, each followed by a full wait.
      auto load_d = [&](auto comp_c) {
        const int32_t* accc = decltype(comp_c)::value ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(accc, lane + 64 * r, bara));
      };
      auto pair = [&](int compA, int qA, int compB, int qB) {
        double xa[kRegs], xb[kRegs];
        if (work) {
          if constexpr (C::L % 2 != 0) {
            if (qA == 0) { if (compA) load_d(std::true_type{}); else load_d(std::false_type{}); }
          }
          Xf::digits(xa, d, qA);
          if constexpr (C::L % 2 != 0) {
            if (qB == 0) { if (compB) load_d(std::true_type{}); else load_d(std::false_type{}); }
          }
          Xf::digits(xb, d, qB);
          Xf::fwd_pair_wg(lane, xa, xb, tw, buf);
        }
        RS_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        RS_STAMP(2);
        if (work) {
          mac_pair_stream(s0, s1, xa, xb, s_key[0], s_key[1], lane);
        }
        RS_STAMP(3);
        __syncthreads();
        R += 2;
        if (R < total_rows) issue_pair();
        RS_STAMP(4);
      };
      if constexpr (C::L % 2 == 0) {
        if (work) load_d(std::false_type{});
#pragma unroll 1
        for (int q = 0; q < C::L; q += 2) pair(0, q, 0, q + 1);
        if (work) { Xf::mid(s0, s1, f); load_d(std::true_type{}); }
#pragma unroll 1
        for (int q = 0; q < C::L; q += 2) pair(1, q, 1, q + 1);
      } else {
        // odd l: the middle pair straddles the two accumulator components (no place for Xf::mid:
        // the workgroup form is only instantiated for policies whose mid() is empty)
#pragma unroll
        for (int p = 0; p < C::L; ++p) {
          const int rA = 2 * p, rB = 2 * p + 1;
          pair(rA / C::L, rA % C::L, rB / C::L, rB % C::L);
        }
      }

      if (work) {
        // the accumulator words are read BEFORE the inverse pair (the digit transforms are dead, there
        // are registers to spare): read after it, every read-modify-write of the update exposed an LDS
        // round trip behind the store in front of it
        uint32_t a0[kRegs], a1[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; ++r) { a0[r] = (uint32_t)acc0[lane + 64 * r]; a1[r] = (uint32_t)acc1[lane + 64 * r]; }
        wave_lds_sync();
        Xf::inverse_pair_wg(lane, s0, s1, tw, buf);
        RS_STAMP(5);
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          const int j = lane + 64 * r;
          acc0[j] = (int32_t)(a0[r] + (uint32_t)Xf::to_torus(s0[r], dev));
          acc1[j] = (int32_t)(a1[r] + (uint32_t)Xf::to_torus(s1[r], dev));
        }
        wave_lds_sync();
        RS_STAMP(6);
      }
    }

    if (active) {
      // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
      int32_t* out = a.u_out + ct * (kN + 1);
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        out[j] = (j == 0) ? acc0[0] : (int32_t)(0u - (uint32_t)acc0[kN - j]);
      }
      if (lane == 0) out[kN] = acc1[0];
    }
    RS_STAMP(7);
  }
  RS_STAMP_FLUSH(wave);
  if (wave == 0) cohort_leave<BlindRotateArgs>(steps_done);
  if (Xf::kCertificate) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Blind rotation, lock-step workgroup form on the SPLIT key (RS_MODE_FFT_SPLIT at throughput batch sizes, N = 1024).
// Same structure as blind_rotate_wg_kernel -- 8 waves walk 8 ciphertexts in lock step and share the key through LDS --
// but every key row comes as two 16 KB half-rows (the low and the high 16-bit half of the key, rs_general.h), each
// multiplied into its own pair of column sums: four inverse transforms per CMUX step instead of two, and the result
// acc += round(lo) + (round(hi) << 16) is exact by the a-priori bound of rs_general.h (no certificate).
// Four accumulators leave registers for ONE digit transform in flight (the unsplit kernel pairs them); the inverse
// transforms still run as software-pipelined pairs. LDS: 3 ring slots of 16 KB (the accumulators and exchange planes
// of 8 ciphertexts leave room for no more), so `bara` lives in a 64-step window refilled from global memory.
// Half-row h sits in slot h mod 3 and is requested two half-rows ahead: the barrier that publishes h also says every
// wave has finished h - 1, whose slot then takes h + 2.
// -------------------------------------------------------------------------------------------------
// s0 += x * (column at k0), s1 += x * (column at k1): the two columns of one key half-row
__device__ __forceinline__ void mac_half_stream_cols(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs], const double2* k0, const double2* k1, int lane) {
  double2 u[2][4];
  auto issue = [&](int step, double2 (&w)[4]) {
    const int v = 2 * step;
    w[0] = k0[v * 64 + lane]; w[1] = k0[(v + 1) * 64 + lane];
    w[2] = k1[v * 64 + lane]; w[3] = k1[(v + 1) * 64 + lane];
  };
  auto fma = [&](int step, const double2 (&w)[4]) {
    const int v = 2 * step;
    fft_cmac(s0[v], s0[v + 8], x[v], x[v + 8], w[0].x, w[0].y);
    fft_cmac(s0[v + 1], s0[v + 9], x[v + 1], x[v + 9], w[1].x, w[1].y);
    fft_cmac(s1[v], s1[v + 8], x[v], x[v + 8], w[2].x, w[2].y);
    fft_cmac(s1[v + 1], s1[v + 9], x[v + 1], x[v + 9], w[3].x, w[3].y);
  };
  issue(0, u[0]);
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    if (step + 1 < 4) issue(step + 1, u[(step + 1) & 1]);
    RS_MAC_FENCE();
    fma(step, u[step & 1]);
    RS_MAC_FENCE();
  }
}

__device__ __forceinline__ void mac_half_stream(double (&s0)[kRegs], double (&s1)[kRegs], const double (&x)[kRegs], const double* key, int lane) {
  const double2* k0 = reinterpret_cast<const double2*>(key);
  mac_half_stream_cols(s0, s1, x, k0, k0 + kN / 2, lane);
}

template <class C, int WPB>
__global__ __launch_bounds__(64 * WPB) void blind_rotate_wgs_kernel(BlindRotateArgs a) {
  using Xf = XfFft<C>;
  static_assert(WPB == 8 || WPB == 4, "a 16 KB half-row is fetched as 16 / WPB one-KB chunks per wave");
  constexpr int kChunks = 16 / WPB;
  constexpr int KPL = 2 * C::L;
  constexpr int kSlotDoubles = 2 * kN;   // one key half-row: 2 columns x N doubles = 16 KB
  constexpr int kWin = 64;
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ __attribute__((aligned(16))) double s_buf[WPB][Xf::kWgBufDoubles];
  __shared__ int32_t s_acc[WPB][2][kN];
  __shared__ __attribute__((aligned(16))) double s_key[3][kSlotDoubles];
  __shared__ uint16_t s_bara[WPB][kWin];
  __shared__ int s_mail[kCohortSlots];   // XCD cohorts: the progress row requested a step ago (wave 0 only; rs_cohort.h)
  stage_tables(s_tw, a.tw, 64 * WPB, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  double* buf = s_buf[wave];
  int32_t* acc0 = s_acc[wave][0];
  int32_t* acc1 = s_acc[wave][1];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  FftTwKept<9> tw_kept;
  fft_kept_load(tw_kept, tw);
  const int n = a.n;
  const long n_groups = (a.B + WPB - 1) / WPB;
  const int total_half = n * KPL * 2;   // half-rows of one blind rotation (n <= 1024 steps x 4 l: far inside an int; scalar compares)
  const unsigned lane_off = (unsigned)lane * 16u;
  auto sync_w = [] { wave_lds_sync(); };
  RS_WGS_STAMP_DECL;   // -DRS_DIAG=2 (tools/stamp_profile.py --split): 0 step prologue + rotated differences, 1 digits + forward transform,
                       // 2 key wait + barrier (low half), 3 multiply-accumulate low, 4 key wait + barrier (high half), 5 multiply-accumulate high,
                       // 6 two inverse pairs + update, 7 group prologue / extract
  int steps_done = 0;   // CMUX steps of the groups this workgroup has finished (XCD cohorts, rs_cohort.h)

  for (long group = blockIdx.x; group < n_groups; group += gridDim.x, steps_done += n) {
    const long ct = group * WPB + wave;
    const bool active = ct < a.B;
    const int32_t* row0 = a.in0 + (active ? ct : 0) * a.W;
    const int32_t* row1 = a.in1 ? a.in1 + (active ? ct : 0) * a.W : nullptr;
    auto word = [&](int i) -> int32_t {
      uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
      if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
      return (int32_t)v;
    };
    auto fill_window = [&](int i0) {   // bara of steps [i0, i0 + 64): only this wave reads its row
      const int i = i0 + lane;
      s_bara[wave][lane] = (active && i < n) ? (uint16_t)modswitch_2N(word(i)) : (uint16_t)0;
    };
    if (active) {
      const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
      const int rot = 2 * kN - barb;  // in (0, 2N]
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc0[j] = 0;
        acc1[j] = test_vector(a, ct, j, rot);
      }
    }
    fill_window(0);
    // every wave has left the previous group's last multiply-accumulate before the ring is refilled
    __syncthreads();
    int h_issue = 0;         // next half-row to request
    int slot_issue = 0;      // its slot, h_issue mod 3
    const double* src_next = a.bk_x + (size_t)(wave * kChunks) * 128;   // this wave's share of the next half-row
    const unsigned key_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&s_key[0][0]) +
                             (unsigned)(wave * kChunks) * 1024u;
    auto issue_next = [&]() {
      if (h_issue < total_half) {
        // half-rows are requested in storage order: a running pointer and the slot's byte address, seven scalar instructions
        // instead of the twenty-two of the general form (a 64-bit index compare, a shift-and-add pair, a pointer cast)
        glds_chunks_at<kChunks>(src_next, lane_off, key_lds + (unsigned)slot_issue * (unsigned)(kSlotDoubles * sizeof(double)));
        src_next += kSlotDoubles;
        if constexpr (diag::kNoKeyProbe) {   // diagnostic builds: every step reads the half-rows of step 0 (they stay in the L2s)
          if ((h_issue + 1) % (2 * KPL) == 0) src_next -= (size_t)(2 * KPL) * kSlotDoubles;
        }
        ++h_issue;
        slot_issue = slot_issue == 2 ? 0 : slot_issue + 1;
      }
    };
    issue_next();
    issue_next();
    int h = 0;               // half-row consumed next
    int slot = 0;
    // publishes half-row h (every wave first waits for its own share: at most the next half-row's two loads may still
    // be in flight) and frees the slot of h - 1 for h + 2
    // (a bare s_barrier behind explicit counts: __syncthreads() would drain every outstanding load, i.e. also the
    // half-row requested one barrier ago, and with it half of the prefetch distance)
    auto publish = [&]() {
      if (h + 1 < total_half) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(kChunks) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      issue_next();
    };
    auto consumed = [&]() { ++h; slot = slot == 2 ? 0 : slot + 1; };

    RS_WGS_STAMP(7);
    for (int i = 0; i < n; ++i) {
      if ((i & (kWin - 1)) == 0 && i > 0) { wave_lds_sync(); fill_window(i); }
      if (wave == 0) cohort_step<BlindRotateArgs>(steps_done + i, s_mail);
      wave_lds_sync();
      const int32_t bara = __builtin_amdgcn_readfirstlane((int)s_bara[wave][i & (kWin - 1)]);
      const bool work = bara != 0;   // tfhe_blindRotate_FFT skips the identity CMUX (the barriers still run)
      double sl0[kRegs], sl1[kRegs], sh0[kRegs], sh1[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { sl0[u] = 0.0; sl1[u] = 0.0; sh0[u] = 0.0; sh1[u] = 0.0; }
      int32_t d[kRegs];
      auto load_d = [&](auto comp_c) {
        const int32_t* accc = decltype(comp_c)::value ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(accc, lane + 64 * r, bara));
      };
      auto row = [&](int q) {
        double x[kRegs];
        if (work) {
          Xf::digits(x, d, q);
          ffwd_planar(lane, x, tw_kept, buf, sync_w);
        }
        RS_WGS_STAMP(1);
        publish();
        RS_WGS_STAMP(2);
        if (work) mac_half_stream(sl0, sl1, x, s_key[slot], lane);
        consumed();
        RS_WGS_STAMP(3);
        publish();
        RS_WGS_STAMP(4);
        if (work) mac_half_stream(sh0, sh1, x, s_key[slot], lane);
        consumed();
        RS_WGS_STAMP(5);
      };
      // (the forward transforms stay single: run as software-pipelined pairs -- two transforms beside the four 32-register column
      // sums -- the kernel does not fit 256 registers: 1,040 bytes of scratch per lane, compiled in round 4 and dropped)
      if (work) load_d(std::false_type{});
      RS_WGS_STAMP(0);
#pragma unroll 1
      for (int q = 0; q < C::L; ++q) row(q);
      if (work) load_d(std::true_type{});
      RS_WGS_STAMP(0);
#pragma unroll 1
      for (int q = 0; q < C::L; ++q) row(q);

      if (work) {
        Xf::inverse_pair_wg(lane, sl0, sl1, tw, buf);
        uint32_t lo0[kRegs], lo1[kRegs];
#pragma unroll
        for (int r = 0; r < kRegs; ++r) { lo0[r] = (uint32_t)f_to_torus32(sl0[r]); lo1[r] = (uint32_t)f_to_torus32(sl1[r]); }
        Xf::inverse_pair_wg(lane, sh0, sh1, tw, buf);
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          const int j = lane + 64 * r;
          acc0[j] = (int32_t)((uint32_t)acc0[j] + lo0[r] + ((uint32_t)f_to_torus32(sh0[r]) << 16));
          acc1[j] = (int32_t)((uint32_t)acc1[j] + lo1[r] + ((uint32_t)f_to_torus32(sh1[r]) << 16));
        }
        wave_lds_sync();
      }
      RS_WGS_STAMP(6);
    }

    if (active) {
      // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
      int32_t* out = a.u_out + ct * (kN + 1);
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        out[j] = (j == 0) ? acc0[0] : (int32_t)(0u - (uint32_t)acc0[kN - j]);
      }
      if (lane == 0) out[kN] = acc1[0];
    }
  }
  RS_WGS_STAMP(7);
  RS_WGS_STAMP_FLUSH(wave);
  if (wave == 0) cohort_leave<BlindRotateArgs>(steps_done);
}

// -------------------------------------------------------------------------------------------------
// Blind rotation, "duo" form on the SPLIT key (RS_MODE_FFT_SPLIT, 2 x #CUs < B <= 4 x #CUs, N = 1024, any l): a workgroup
// is 4 ciphertexts x 2 waves, as in blind_rotate_duo_kernel -- wave (c, h) owns accumulator component h of ciphertext c. It
// transforms the l digit rows of that component (one transform in flight: the four partial sums low / high half x two columns
// fill 128 registers, as in blind_rotate_wgs_kernel) and multiplies each into all four sums; then it hands the two partials of
// column 1 - h to its partner, adds the partner's partials of column h to its own, runs the two inverse transforms of column h
// (low and high half) as one software-pipelined pair and updates component h: acc += round(lo) + (round(hi) << 16).
// Against the 4-wave lock-step groups that served this batch range (one wave per ciphertext, one wave per SIMD) a wave does
// half the transforms of a CMUX step and the second wave slot of every SIMD is in use.
// Key: the 8 waves run in lock step; per (digit row q, key half) a PAIR of 16 KB half-rows -- the one of component 0 and the
// one of component 1 -- is fetched by direct global->LDS loads, 4 one-KB chunks per wave, into pair slot p & 1 (p numbers the
// pairs in the order they are consumed). The barrier that publishes pair p also says every wave has finished pair p - 1, whose
// slot then takes pair p + 1 (an L2-resident half-row lands within one multiply-accumulate phase: measured on the wgs
// kernel). The partials change hands through the same 64 KB once the last pair of a step has been consumed, low halves first,
// then high halves (8 waves x 8 KB each time), so the first pair of the next step is requested behind that exchange; it has
// the inverse transforms, the rotated difference and a forward transform to arrive. Barriers per CMUX step: 2 l + 4.
// -------------------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(512) void blind_rotate_duos_kernel(BlindRotateArgs a) {
  using Xf = XfFft<C>;
  constexpr int KPL = 2 * C::L;
  constexpr int kSlotDoubles = 2 * kN;   // one key half-row: 2 columns x N doubles = 16 KB
  constexpr int kCts = 4;
  constexpr int kWin = 64;
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ __attribute__((aligned(16))) double s_buf[8][Xf::kWgBufDoubles];
  __shared__ int32_t s_acc[kCts][2][kN];
  __shared__ __attribute__((aligned(16))) double s_key[4][kSlotDoubles];   // slot 2 (p & 1) + component
  __shared__ uint16_t s_bara[8][kWin];                                      // one window per wave (the two waves of a ciphertext fill the same values)
  stage_tables(s_tw, a.tw, 512, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int c = wave >> 1, h = wave & 1;
  double* buf = s_buf[wave];
  int32_t* acc = s_acc[c][h];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  const int n = a.n;
  const long n_groups = (a.B + kCts - 1) / kCts;
  const int total_pairs = n * C::L * 2;   // 32-bit counters: scalar compares (see blind_rotate_wgs_kernel)
  const unsigned lane_off = (unsigned)lane * 16u;
  auto sync_w = [] { wave_lds_sync(); };
  // own += x * column h, given += x * column 1 - h of the half-row in `slot`
  auto mac_cols = [&](double (&own)[kRegs], double (&given)[kRegs], const double (&x)[kRegs], const double* slot) {
    const double2* k = reinterpret_cast<const double2*>(slot);
    mac_half_stream_cols(own, given, x, k + h * (kN / 2), k + (1 - h) * (kN / 2), lane);
  };
  // pair p = (i L + k) 2 + half holds the half-rows ((i KPL + comp L + q_k) 2 + half) of comp = 0, 1, where q_k = (k + rot) mod L:
  // workgroup b walks the l digits of a step in the order rotated by b (the rows of a step are independent), so that the 256
  // workgroups of a launch do not pull the same half-rows through the same L2 channels at the same moments (600 sign
  // bootstraps 12.06 -> 11.25 ms, profiles/r03/v_ab_*). This wave fetches chunks [4 (wave & 3), +4) of component wave >> 2.
  const int rot = (int)(blockIdx.x % C::L);
  const int kcomp = duos_fetch_comp(wave);                                 // placement: rs_lds_plan.h (checked on the host)
  const size_t chunk_off = (size_t)duos_first_chunk(wave) * 128;
  const unsigned key_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)&s_key[0][0]) + (unsigned)chunk_off * 8u;
  int issued;
  int iss_i, iss_k;   // step and position (digit slot, half) of the next pair to request
  auto issue_reset = [&] { issued = 0; iss_i = 0; iss_k = 0; };
  auto issue_next = [&] {
    if (issued >= total_pairs) return;
    int q = (iss_k >> 1) + rot;
    if (q >= C::L) q -= C::L;
    // the half-row's byte offset fits 32 bits (at most n * 4 l half-rows of 16 KB: 229 MB for the REDsec set); the slot as an LDS byte address
    const unsigned hrow = (unsigned)(((iss_i * KPL + kcomp * C::L + q) << 1) + (iss_k & 1));
    glds_chunks_at<4>(reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.bk_x) + (size_t)(hrow * (unsigned)(kSlotDoubles * sizeof(double)))) + chunk_off, lane_off,
                      key_lds + (unsigned)duos_pair_slot(issued, kcomp) * (unsigned)(kSlotDoubles * sizeof(double)));
    ++issued;
    if (++iss_k == 2 * C::L) { iss_k = 0; ++iss_i; }
  };

  for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
    const long ct = group * kCts + c;
    const bool active = ct < a.B;
    const int32_t* row0 = a.in0 + (active ? ct : 0) * a.W;
    const int32_t* row1 = a.in1 ? a.in1 + (active ? ct : 0) * a.W : nullptr;
    auto word = [&](int i) -> int32_t {
      uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
      if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
      return (int32_t)v;
    };
    auto fill_window = [&](int i0) {   // bara of steps [i0, i0 + 64): read back by this wave only
      const int i = i0 + lane;
      s_bara[wave][lane] = (active && i < n) ? (uint16_t)modswitch_2N(word(i)) : (uint16_t)0;
    };
    if (active) {
      const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
      const int rot = 2 * kN - barb;  // in (0, 2N]
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc[j] = h ? test_vector(a, ct, j, rot) : 0;
      }
    }
    fill_window(0);
    __syncthreads();   // every wave has left the previous group's last exchange and extract before the key buffer is refilled
    int p = 0;         // pair consumed next; it has been requested, pair p + 1 has not
    issue_reset();
    issue_next();
    // publishes pair p (every wave first waits for its own share; nothing else of this wave is in flight) and requests pair
    // p + 1 into the other slot unless the step's exchange needs the buffer first (`hold`)
    auto publish = [&](bool hold) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (!hold) issue_next();
    };

    for (int i = 0; i < n; ++i) {
      if ((i & (kWin - 1)) == 0 && i > 0) { wave_lds_sync(); fill_window(i); }
      wave_lds_sync();
      const int32_t bara = __builtin_amdgcn_readfirstlane((int)s_bara[wave][i & (kWin - 1)]);
      const bool work = bara != 0;   // tfhe_blindRotate_FFT skips the identity CMUX (the barriers still run)
      // partial sums of the column this wave inverts (own = column h) and of the one its partner inverts, low / high key half
      double lo[kRegs], hi[kRegs], glo[kRegs], ghi[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { lo[u] = 0.0; hi[u] = 0.0; glo[u] = 0.0; ghi[u] = 0.0; }
      int32_t d[kRegs];
      if (work) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(acc, lane + 64 * r, bara));
      }
#pragma unroll 1
      for (int q = 0; q < C::L; ++q) {
        double x[kRegs];
        if (work) {
          int qd = q + rot;
          if (qd >= C::L) qd -= C::L;
          Xf::digits(x, d, qd);
          ffwd_planar(lane, x, tw, buf, sync_w);
        }
        publish(false);
        if (work) mac_cols(lo, glo, x, s_key[duos_pair_slot(p, h)]);
        ++p;
        publish(q + 1 == C::L);
        if (work) mac_cols(hi, ghi, x, s_key[duos_pair_slot(p, h)]);
        ++p;
      }
      // partial exchange through the key buffer (64 KB = 8 waves x 8 KB), low halves, then high halves: wave (c, h) hands over
      // its partials of column 1 - h and adds its partner's partials of column h to its own
      double* mine_xchg = &s_key[0][0] + duo_xchg_doubles(wave);
      const double* theirs = &s_key[0][0] + duo_xchg_doubles(duo_partner(wave));
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave has consumed the last pair
      if (work) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) mine_xchg[u * 64 + lane] = glo[u];
      }
      __syncthreads();
      if (work) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) lo[u] += theirs[u * 64 + lane];
      }
      __syncthreads();
      if (work) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) mine_xchg[u * 64 + lane] = ghi[u];
      }
      __syncthreads();
      if (work) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) hi[u] += theirs[u * 64 + lane];
      }
      __syncthreads();                             // partials consumed: the key buffer may be refilled
      issue_next();
      if (work) {
        uint32_t a0[kRegs];   // accumulator words read ahead of the inverse transforms (see the workgroup kernel)
#pragma unroll
        for (int r = 0; r < kRegs; ++r) a0[r] = (uint32_t)acc[lane + 64 * r];
        wave_lds_sync();
        Xf::inverse_pair_wg(lane, lo, hi, tw, buf);
#pragma unroll
        for (int r = 0; r < kRegs; ++r)
          acc[lane + 64 * r] = (int32_t)(a0[r] + (uint32_t)f_to_torus32(lo[r]) + ((uint32_t)f_to_torus32(hi[r]) << 16));
        wave_lds_sync();
      }
    }

    if (active) {
      // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
      int32_t* out = a.u_out + ct * (kN + 1);
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          const int j = lane + 64 * r;
          out[j] = (j == 0) ? acc[0] : (int32_t)(0u - (uint32_t)acc[kN - j]);
        }
      } else if (lane == 0) {
        out[kN] = acc[0];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Cooperative blind rotation on the SPLIT key (RS_MODE_FFT_SPLIT at latency batch sizes, B <= 2 x #CUs, N = 1024): G waves
// share ONE ciphertext as in blind_rotate_coop_kernel. Wave g transforms the digit rows [g R, (g+1) R) and multiplies each
// into FOUR partial sums (low / high key half x two columns; sum index = 2 half + column); every sum has one owner wave
// that keeps its own partial in registers, adds the other waves' partials from LDS and runs the inverse transform:
//   G = 4: wave s owns sum s; the rounded low and high results of a column meet in the accumulator by LDS integer
//          atomics (exact, order-independent);
//   G = 2: wave w owns both halves of column w -- its two inverse transforms run as a software-pipelined pair.
// Key half-rows stream from L2 into registers in four chunks per row (the first one requested across the transform).
// -------------------------------------------------------------------------------------------------
template <class C, int G>
__global__ __launch_bounds__(64 * G) void blind_rotate_coops_kernel(BlindRotateArgs a) {
  using Xf = XfFft<C>;
  constexpr int KPL = 2 * C::L;
  constexpr int R = KPL / G;
  static_assert((G == 2 || G == 4) && KPL % G == 0, "waves split the digit rows evenly within a component");
  constexpr int OWN = 4 / G;                  // sums per owner wave
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ __attribute__((aligned(16))) double s_buf[G][kBufDoubles];
  __shared__ double s_part[G][4 - OWN][kN];   // the sums a wave does NOT own (G = 4: 96 KB, G = 2: 32 KB)
  __shared__ int32_t s_acc[2][kN];
  stage_tables(s_tw, a.tw, 64 * G, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long ct = blockIdx.x;
  const Field f = a.f;
  double* buf = s_buf[wave];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
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
  auto owner = [](int sum) { return coops_owner<G>(sum); };              // placement: rs_lds_plan.h (checked on the host)
  auto slot = [](int sum, int g) { return coops_slot<G>(sum, g); };      // index among the sums wave g does not own
  if (wave < 2) {
    const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
    const int rot = 2 * kN - barb;
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      const int j = lane + 64 * r;
      s_acc[wave][j] = wave == 0 ? 0 : test_vector(a, ct, j, rot);
    }
  }
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    const int32_t bara = __builtin_amdgcn_readfirstlane(modswitch_2N(word(i)));
    if (bara == 0) continue;   // uniform over the workgroup
    double s[4][kRegs];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int u = 0; u < kRegs; ++u) s[k][u] = 0.0;
    int32_t d[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(s_acc[comp], lane + 64 * r, bara));
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
      const int row = row_begin + (int)((rr + blockIdx.x) % R);
      const int q = row - comp * C::L;
      // half-row (row, half) = [column 0: N doubles][column 1: N doubles], pairs (re, im) of position 8 lane + v at [v][lane]
      const double2* lo0 = reinterpret_cast<const double2*>(a.bk_x + ((size_t)i * KPL + row) * 4 * kN);
      const double2* lo1 = lo0 + kN / 2;
      const double2* hi0 = lo0 + kN;
      const double2* hi1 = hi0 + kN / 2;
      auto load4 = [&](const double2* k0, const double2* k1, int v0, double2 (&w0)[4], double2 (&w1)[4]) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { w0[v] = k0[(v0 + v) * 64 + lane]; w1[v] = k1[(v0 + v) * 64 + lane]; }
      };
      double x[kRegs];
      double2 wa0[4], wa1[4], wb0[4], wb1[4], wc0[4], wc1[4], wd0[4], wd1[4];
      load4(lo0, lo1, 0, wa0, wa1);
      load4(lo0, lo1, 4, wb0, wb1);
      load4(hi0, hi1, 0, wc0, wc1);
      load4(hi0, hi1, 4, wd0, wd1);
      Xf::fwd_digits(lane, x, d, q, 0u, tw, buf, f);
      Xf::mac(s[0], s[1], x, wa0, wa1, 0, f);
      Xf::mac(s[0], s[1], x, wb0, wb1, 4, f);
      Xf::mac(s[2], s[3], x, wc0, wc1, 0, f);
      Xf::mac(s[2], s[3], x, wd0, wd1, 4, f);
    }
    // partial sums a wave does not own go through LDS (position u*64 + lane is conflict-free)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (owner(k) != wave) {
        double* dst = s_part[wave][slot(k, wave)];
#pragma unroll
        for (int u = 0; u < kRegs; ++u) dst[u * 64 + lane] = s[k][u];
      }
    }
    __syncthreads();   // partials visible; every wave has finished reading the accumulator
    if constexpr (G == 4) {
      double x[kRegs];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (wave == k) {
#pragma unroll
          for (int u = 0; u < kRegs; ++u) x[u] = s[k][u];
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g != wave) {
          const double* src = s_part[g][slot(wave, g)];
#pragma unroll
          for (int u = 0; u < kRegs; ++u) x[u] += src[u * 64 + lane];
        }
      }
      Xf::inverse(lane, x, tw, buf, f);
      const int sh = wave >= 2 ? 16 : 0;
      int32_t* acc = s_acc[wave & 1];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) atomicAdd(reinterpret_cast<unsigned*>(acc) + lane + 64 * r, (uint32_t)f_to_torus32(x[r]) << sh);
    } else {
      double xa[kRegs], xb[kRegs];
      if (wave == 0) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) { xa[u] = s[0][u]; xb[u] = s[2][u]; }
      } else {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) { xa[u] = s[1][u]; xb[u] = s[3][u]; }
      }
      const double* pa = s_part[1 - wave][0];
      const double* pb = s_part[1 - wave][1];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { xa[u] += pa[u * 64 + lane]; xb[u] += pb[u * 64 + lane]; }
      Xf::inverse2(lane, xa, xb, tw, buf, f);
      int32_t* acc = s_acc[wave];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc[j] = (int32_t)((uint32_t)acc[j] + (uint32_t)f_to_torus32(xa[r]) + ((uint32_t)f_to_torus32(xb[r]) << 16));
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

// (An eight-wave form of this kernel -- blind_rotate_coops8_kernel, round 4: rows over 8 waves as in blind_rotate_coop8_kernel, the
// four sums met by LDS f64 atomics -- was built, bit-exact, and is SLOWER: 4.90 / 6.93 ms against 4.08 ms for 196 sign bootstraps
// (profiles/r04/i_ab_coop8_atomics_and_coops8.txt). Four 32-register sums beside a transform leave a wave of a two-wave SIMD (256
// registers) no room to keep a key row in flight across the transform, which is what the four-wave form's 412 registers buy. Removed.)
// -------------------------------------------------------------------------------------------------
// Blind rotation, "duo" workgroup form (mid-size batches: 2 x #CUs < B < 8 x #CUs, even l).
// One wave per ciphertext leaves half the wave slots empty there and every wave streams the whole key
// by itself (the 1,024-neuron MNIST layer was bound by ~10 TB/s of key reads). Here a workgroup is
// 4 ciphertexts x 2 waves: wave (c, h) owns accumulator component h of ciphertext c -- it transforms
// the l digit rows of that component (in software-pipelined pairs), accumulates partial sums for both
// output columns, receives column h's other partial from its partner, runs that column's inverse
// transform and updates component h. The 8 waves run in lock step and share the key rows through LDS:
// per pair of rows a "quad" (2 rows of each component, 64 KB) is fetched by direct global->LDS loads,
// 1/8 per wave. Barriers per CMUX step: 2 per row pair + 2 around the partial exchange, which goes
// through the (then idle) quad buffer.
// -------------------------------------------------------------------------------------------------
template <class Xf>
__global__ __launch_bounds__(512) void blind_rotate_duo_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  static_assert(C::L % 2 == 0, "rows are processed in pairs within one component");
  constexpr int KPL = 2 * C::L;
  constexpr int kRowDoubles = 2 * kN;
  constexpr int kCts = 4;
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ __attribute__((aligned(16))) double s_buf[8][Xf::kWgBufDoubles];
  __shared__ int32_t s_acc[kCts][2][kN];
  __shared__ __attribute__((aligned(16))) double s_key[4][kRowDoubles];   // slot 2 h + k: row k of the pair, component h
  __shared__ uint16_t s_bara[kCts][kSmall];
  stage_tables(s_tw, a.tw, 512, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int c = wave >> 1, h = wave & 1;
  const int own_col = h ? kN / 2 : 0, given_col = h ? 0 : kN / 2;   // offsets (double2) of the two columns within a key row
  const Field f = a.f;
  double* buf = s_buf[wave];
  int32_t* acc = s_acc[c][h];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  FftTwKept<6> tw_kept;
  fft_kept_load(tw_kept, tw);
  const int n = a.n;
  constexpr uint32_t offset = gadget_offset<C>();
  double dev = 0.0;
  const long n_groups = (a.B + kCts - 1) / kCts;
  RS_DUO_STAMP_DECL;   // -DRS_DIAG=4 (tools/stamp_coop8.py duo): 0 step prologue + rotated difference, 1 digits + forward pair,
                       // 2 key wait + barrier 1, 3 multiply-accumulate, 4 barrier 2 + next quad, 5 partial exchange (2 barriers), 6 inverse + update, 7 group prologue / extract

  const unsigned lane_off = (unsigned)lane * 16u;
  // (Row pairs walked in per-workgroup rotated orders, as in the split forms: no effect on the 1,024-neuron MNIST layer,
  // profiles/r03/v_ab_duo_row_rotation.txt -- every workgroup walks them in storage order.)
  // quad (i, p): rows i KPL + hh L + 2 p + k; this wave fetches half of slot (wave >> 1)
  auto issue_quad = [&](int i, int p) {
    const int slot = duo_quad_slot(wave), chunk0 = duo_quad_first_chunk(wave);   // placement: rs_lds_plan.h (checked on the host)
    const long R = (long)i * KPL + (slot >> 1) * C::L + 2 * p + (slot & 1);
    const double* src = a.bk_x + (size_t)R * kRowDoubles + (size_t)chunk0 * 128;
    double* dst = s_key[slot] + chunk0 * 128;
    glds_chunks<4>(src, lane_off, dst);
    glds_chunks<4>(src + 4 * 128, lane_off, dst + 4 * 128);
  };
  // (The eight 1 KB requests of a wave's share of the next quad spread over the six segments of the next forward pair instead of one burst
  // behind the barrier: 7.96 -> 9.97 ms at 1,024 ciphertexts, profiles/r04/aq_*: the rows arrive late and the asm statements cut the pair's schedule.)
  for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
    const long ct = group * kCts + c;
    const bool active = ct < a.B;
    if (active) {
      const int32_t* row0 = a.in0 + ct * a.W;
      const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
      auto word = [&](int i) -> int32_t {
        uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
        if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
        return (int32_t)v;
      };
      for (int i = lane + 64 * h; i < n; i += 128) s_bara[c][i] = (uint16_t)modswitch_2N(word(i));
      const int32_t barb = modswitch_2N((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst));
      const int rot = 2 * kN - barb;  // in (0, 2N]
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc[j] = h ? test_vector(a, ct, j, rot) : 0;
      }
    }
    __syncthreads();   // bara complete; previous group's last reads of the quad buffer are over
    issue_quad(0, 0);
    RS_DUO_STAMP(7);

    unsigned bara_next = active ? s_bara[c][0] : 0;   // read one step ahead
    for (int i = 0; i < n; ++i) {
      const int32_t bara = __builtin_amdgcn_readfirstlane((int)bara_next);
      bara_next = (active && i + 1 < n) ? s_bara[c][i + 1] : 0;
      const bool work = bara != 0;   // tfhe_blindRotate_FFT skips the identity CMUX
      // own: partial sum of the column this wave inverts (column h); given: of the column its partner inverts. Addressed through the
      // key-row offsets (own_col, given_col) instead of (column 0, column 1) selected by h afterwards: those selects were 310
      // v_cndmask per CMUX step (the compiler cannot know h at compile time)
      double own[kRegs], given[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { own[u] = 0.0; given[u] = 0.0; }
      int32_t d[kRegs];
      if (work) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(acc, lane + 64 * r, bara));
      }
      RS_DUO_STAMP(0);
#pragma unroll 1
      for (int p = 0; p < C::L / 2; ++p) {
        double xa[kRegs], xb[kRegs];
        if (work) {
          Xf::digits(xa, d, 2 * p);
          Xf::digits(xb, d, 2 * p + 1);
          Xf::fwd_pair_wg(lane, xa, xb, tw_kept, buf);
        }
        RS_DUO_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // quad (i, p) published
        RS_DUO_STAMP(2);
        if (work) {
          mac_pair_stream(own, given, xa, xb, s_key[2 * h], s_key[2 * h + 1], lane, own_col, given_col);
        }
        RS_DUO_STAMP(3);
        __syncthreads();                         // every wave has finished reading it
        if (p + 1 < C::L / 2) issue_quad(i, p + 1);
        RS_DUO_STAMP(4);
      }
      // partial exchange through the idle quad buffer: wave (c, h) hands over its partial of column 1 - h
      double* xchg = &s_key[0][0] + duo_xchg_doubles(wave);
      if (work) {
#pragma unroll
        for (int u = 0; u < kRegs; ++u) xchg[u * 64 + lane] = given[u];
      }
      __syncthreads();
      double (&mine)[kRegs] = own;
      if (work) {
        const double* theirs = &s_key[0][0] + duo_xchg_doubles(duo_partner(wave));
#pragma unroll
        for (int u = 0; u < kRegs; ++u) mine[u] += theirs[u * 64 + lane];
      }
      __syncthreads();                           // partials consumed: the quad buffer may be refilled
      if (i + 1 < n) issue_quad(i + 1, 0);
      RS_DUO_STAMP(5);
      if (work) {
        uint32_t a0[kRegs];   // accumulator words read ahead of the inverse transform (see the workgroup kernel)
#pragma unroll
        for (int r = 0; r < kRegs; ++r) a0[r] = (uint32_t)acc[lane + 64 * r];
        wave_lds_sync();
        Xf::inverse_wg(lane, mine, tw, buf, f);
#pragma unroll
        for (int r = 0; r < kRegs; ++r) acc[lane + 64 * r] = (int32_t)(a0[r] + (uint32_t)Xf::to_torus(mine[r], dev));
        wave_lds_sync();
      }
      RS_DUO_STAMP(6);
    }

    if (active) {
      // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
      int32_t* out = a.u_out + ct * (kN + 1);
      if (h == 0) {
#pragma unroll
        for (int r = 0; r < kRegs; ++r) {
          const int j = lane + 64 * r;
          out[j] = (j == 0) ? acc[0] : (int32_t)(0u - (uint32_t)acc[kN - j]);
        }
      } else if (lane == 0) {
        out[kN] = acc[0];
      }
    }
  }
  RS_DUO_STAMP(7);
  RS_DUO_STAMP_FLUSH(wave);
  if (Xf::kCertificate) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Cooperative blind rotation (latency form, B <= 2 x #CUs): G waves share ONE ciphertext.
// Wave g transforms the digit polynomials [g R, (g+1) R) (R = 2l / G, so each wave stays within one
// accumulator component) and accumulates its partial column sums; the partials meet in LDS, waves 0
// and 1 sum one column each, run the inverse transform and update the shared accumulator. Two
// workgroup barriers per CMUX step.
// -------------------------------------------------------------------------------------------------
template <class Xf, int G>
__global__ __launch_bounds__(64 * G) void blind_rotate_coop_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  if (recompute_not_needed(a)) return;
  constexpr int KPL = 2 * C::L;
  constexpr int R = KPL / G;
  static_assert(KPL % G == 0 && G % 2 == 0, "waves must split the digit rows evenly within a component");
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[G][kBufDoubles];
  __shared__ double s_part[G][2][kN];
  __shared__ int32_t s_acc[2][kN];
  stage_tables(s_tw, a.tw, 64 * G, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long ct = blockIdx.x;
  const Field f = a.f;
  double* buf = s_buf[wave];
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  const int32_t* row0 = a.in0 + ct * a.W;
  const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
  const int n = a.n;
  const int grp = wave;
  const int comp = grp / (G / 2);
  const int row_begin = grp * R;
  double dev = 0.0;
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
      s_acc[wave][j] = wave == 0 ? 0 : test_vector(a, ct, j, rot);
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
    const double* bk_i = a.bk_x + (size_t)diag::key_step(i) * KPL * 2 * kN;
    int32_t d[kRegs];
#pragma unroll
    for (int r = 0; r < kRegs; ++r) {
      d[r] = rotated_diff(s_acc[comp], lane + 64 * r, bara);
      if (Xf::kPreparedDigits) d[r] = gadget_prepare<C>(d[r]);
    }
#pragma unroll 1
    for (int rr = 0; rr < R; ++rr) {
      // workgroups walk their rows of a step in different orders (the rows of a step are independent), so that the whole chip
      // does not pull the same key rows through the same L2 channels at the same moments: the 196-neuron MNIST layer
      // 3.33 -> 3.15 ms (profiles/r03/v_ab_coop_row_rotation.txt)
      const int row = row_begin + (int)((rr + blockIdx.x) % R);
      const int q = row - comp * C::L;
      const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)(row * 2) * kN);
      const double2* bp1 = bp0 + kN / 2;
      double x[kRegs];
      // the whole key row is requested across the transform (3.21 -> 3.01 ms once the rows were rotated, v_ab_coop_whole_row.txt)
      double2 w0[8], w1[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
      Xf::fwd_digits(lane, x, d, q, offset, tw, buf, f);
      Xf::mac8(s0, s1, x, w0, w1, f);
    }
    // partial sums exchanged through LDS: position u*64 + lane is conflict-free
#pragma unroll
    for (int u = 0; u < kRegs; ++u) {
      s_part[wave][0][u * 64 + lane] = Xf::partial(s0[u], f);
      s_part[wave][1][u * 64 + lane] = Xf::partial(s1[u], f);
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
      Xf::inverse(lane, x, tw, buf, f);
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        s_acc[wave][j] = (int32_t)((uint32_t)s_acc[wave][j] + (uint32_t)Xf::to_torus(x[r], dev));
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
  if (Xf::kCertificate && wave < 2) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Cooperative blind rotation, EIGHT waves per ciphertext (latency form for B <= #CUs, FFT mode; round 4).
// The four-wave form above runs one wave per SIMD: a lone wave issues its FP64 and LDS instructions one behind
// the other (78 % of a two-wave SIMD's rate per wave-slot), and its critical path per CMUX step is
// R = 2l/4 forward transforms plus one inverse. Here the 2l digit rows of a step go to 8 waves -- the components
// alternate (wave & 1), and of the four waves of a component the OLDER ones (waves 0-3, which win the issue
// arbitration against their SIMD partners 4-7) take the extra rows, so that every SIMD (waves s and s + 4) carries
// the same number of rows and finishes them together (l = 10: 3+2 on every SIMD; rs_lds_plan.h; round-4 phase stamps:
// with the extra rows on younger waves two SIMDs finished 1,100 cycles late, 196 ciphertexts 2.53 -> 2.43 ms) -- and the
// two inverse transforms to two waves with the fewest rows, on different SIMDs (waves 6 and 7). For l < 4 (six rows for
// eight waves) the first form's deal stays -- waves 0-3 component 0, waves 4-7 component 1, inverse transforms on the
// two waves without rows (3 and 4): measured faster there (2.66 against 2.73 ms).
// Partial column sums meet by LDS floating-point atomics (ds_add_f64, no return value) in s_sum[2][N]; after the barrier each
// inverse wave reads its 16 values, clears them for the next step, transforms and updates the accumulator (a first form parked
// the 14 partials in idle transform buffers and s_part slots: 2.92 ms where the atomics take 2.64, MEASUREMENTS.md R4). Integer
// results are independent of the summation order (the sums are rounded to the exact integers, certificate-tracked as
// everywhere); two workgroup barriers per step. LDS: 8 KB tables + 8 x 9 KB buffers + 16 KB sums + 8 KB accumulator = 104 KB.
// -------------------------------------------------------------------------------------------------
template <class Xf>
__global__ __launch_bounds__(512) void blind_rotate_coop8_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  static_assert(Xf::kCertificate, "FFT policy only: the exact-NTT reduction schedule is validated for four partials");
  if (recompute_not_needed(a)) return;
  constexpr int G = kCoop8Waves, L = C::L, KPL = 2 * L;
  constexpr int kInvA = coop8_inv_a(L), kInvB = coop8_inv_b(L);   // placement: rs_lds_plan.h (checked on the host)
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[G][kBufDoubles];
  __shared__ double s_sum[2][kN];   // the two column sums, added up by LDS floating-point atomics (zero between steps)
  __shared__ int32_t s_acc[2][kN];
  stage_tables(s_tw, a.tw, 64 * G, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long ct = blockIdx.x;
  const Field f = a.f;
  double* buf = s_buf[wave];
  for (int e = threadIdx.x; e < 2 * kN; e += 64 * G) (&s_sum[0][0])[e] = 0.0;
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  const int32_t* row0 = a.in0 + ct * a.W;
  const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
  const int n = a.n;
  // rows [first, first + cnt) of this wave's component (digit index q = first + rr, TGSW row comp * L + q)
  const int comp = coop8_comp(L, wave), cnt = coop8_row_count(L, wave), first = coop8_row_first(L, wave);
  [[maybe_unused]] const int r_first = cnt > 0 ? (int)(blockIdx.x % (unsigned)cnt) : 0;
  double dev = 0.0;
  RS_C8_STAMP_DECL;   // -DRS_DIAG=8 (tools/stamp_coop8.py): 0 mask word, 1 rotated difference, 2 rows (forward +
                      // multiply-accumulate), 3 atomics issued, 4 barrier 1, 5 inverse + accumulator update, 6 barrier 2, 7 prologue / extract
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
      s_acc[wave][j] = wave == 0 ? 0 : test_vector(a, ct, j, rot);
    }
  }
  __syncthreads();
  constexpr uint32_t offset = gadget_offset<C>();
  RS_C8_STAMP(7);
  for (int i = 0; i < n; ++i) {
    // (requesting the mask word of step i + 1 here, a step ahead, was measured: 2.43 -> 2.47 ms for 196 ciphertexts; the load hits the L1)
    const int32_t bara = __builtin_amdgcn_readfirstlane(modswitch_2N(word(i)));
    RS_C8_STAMP(0);
    if (bara == 0) continue;   // uniform over the workgroup
    double s0[kRegs], s1[kRegs];
#pragma unroll
    for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
    if (cnt > 0) {
      const double* bk_i = a.bk_x + (size_t)diag::key_step(i) * KPL * 2 * kN;
      int32_t d[kRegs];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) d[r] = gadget_prepare<C>(rotated_diff(s_acc[comp], lane + 64 * r, bara));
      RS_C8_STAMP(1);
      [[maybe_unused]] int r_run = r_first;   // rr = 0 starts at blockIdx.x mod cnt in every step
#pragma unroll 1
      for (int rr = 0; rr < cnt; ++rr) {
        const int q = first + r_run;                                       // per-workgroup row order, as in the four-wave form
        r_run = r_run + 1 == cnt ? 0 : r_run + 1;
        const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)((comp * L + q) * 2) * kN);
        const double2* bp1 = bp0 + kN / 2;
        double x[kRegs];
        double2 w0[8], w1[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
        Xf::fwd_digits(lane, x, d, q, offset, tw, buf, f);
        Xf::mac8(s0, s1, x, w0, w1, f);
      }
      RS_C8_STAMP(2);
    }
    // every wave adds its two partial sums into the column sums with ds_add_f64 (no return value: 32 instructions that overlap
    // the other waves' transforms); the order of the floating-point additions is free -- the sums are rounded to the exact
    // integers afterwards, with the certificate watching the distance as everywhere
    if (cnt > 0) {
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { unsafeAtomicAdd(&s_sum[0][u * 64 + lane], s0[u]); unsafeAtomicAdd(&s_sum[1][u * 64 + lane], s1[u]); }
    }
    RS_C8_STAMP(3);
    __syncthreads();   // sums complete; every wave has finished reading the accumulator
    RS_C8_STAMP(4);
    if (wave == kInvA || wave == kInvB) {
      double* sum = s_sum[wave == kInvA ? 0 : 1];
      double x[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) x[u] = sum[u * 64 + lane];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) sum[u * 64 + lane] = 0.0;   // for the next step (same lane, same address: in order)
      Xf::inverse(lane, x, tw, buf, f);
      int32_t* acc = s_acc[wave == kInvA ? 0 : 1];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc[j] = (int32_t)((uint32_t)acc[j] + (uint32_t)Xf::to_torus(x[r], dev));
      }
    }
    RS_C8_STAMP(5);
    __syncthreads();   // accumulator updated, sums zero
    RS_C8_STAMP(6);
    continue;
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
  RS_C8_STAMP(7);
  RS_C8_STAMP_FLUSH(wave);
  if (wave == kInvA || wave == kInvB) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// The same form with a LISTED step (round 6), for the deals with at most one row per wave whose inverse waves carry no row
// (coop8_listed: l < 4, default-128). There a step is a serial chain -- mask word -> rotated difference -> ONE row -> atomics ->
// inverse -- and three of its links are shorter here:
// * the CMUX steps that are not the identity are listed ONCE, in the prologue, in LDS (s_steps: (i << 16) | bara, compacted by
//   wave ballots; n <= kCoop8MaxSteps, the launcher's condition for this kernel): a step reads its entry a whole step ahead
//   instead of waiting ~400 cycles for a global load of the ciphertext word in front of its first instruction;
// * the prepared rotated difference (X^bara - 1) * acc + gadget offset of BOTH components is built once per step by all 512
//   threads (4 coefficients each) into s_d[2][N] behind the accumulator update, and the row waves read their 16 values from
//   there (one more workgroup barrier: three per step); in the kernel above each wave of a component rebuilds all 1,024 of
//   them (32 LDS reads + ~160 vector instructions per lane);
// * a wave's key row is requested a phase early, behind barrier 1 of the step before: it travels while the two inverse waves
//   work and the CU's vector-memory path is otherwise idle.
// 196 default-128 ciphertexts: 2.65 -> 2.49 (list + shared difference) -> 2.21-2.23 ms (early request), same box
// (profiles/r06/c_*). For l >= 4 (the REDsec set: five rows per SIMD) the same step was built and measured at +1 %
// (2.42 against 2.40 ms) and stays with the kernel above: there the rows phase is bound by the SIMDs' instruction issue -- 5 rows
// x ~2.2 k cycles, the throughput kernel's own cost per transform -- and what is taken out of the phases in front of it shows up
// again as contention inside it (phase stamps: rotated difference 1.35-1.8 k -> 0.75 k, rows 8.6 / 9.7 k -> 9.3 / 10.9 k cycles);
// an inverse wave that also carries rows cannot request early without standing ~2,000 cycles in the vector-memory issue queue in
// front of its transform (3.1 k -> 5.1 k cycles, +8 %).
// LDS: 104 KB as above + 8 KB rotated difference + 8 KB step list = 120 KB.
// -------------------------------------------------------------------------------------------------
template <class Xf>
__global__ __launch_bounds__(512) void blind_rotate_coop8_listed_kernel(BlindRotateArgs a) {
  using C = typename Xf::Cfg;
  static_assert(Xf::kCertificate, "FFT policy only: the exact-NTT reduction schedule is validated for four partials");
  if (recompute_not_needed(a)) return;
  constexpr int G = kCoop8Waves, L = C::L, KPL = 2 * L;
  constexpr int kInvA = coop8_inv_a(L), kInvB = coop8_inv_b(L);   // placement: rs_lds_plan.h (checked on the host)
  static_assert(coop8_listed(L) && coop8_row_count(L, kInvA) == 0 && coop8_row_count(L, kInvB) == 0, "the early key request assumes inverse waves without rows");
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[G][kBufDoubles];
  __shared__ double s_sum[2][kN];   // the two column sums, added up by LDS floating-point atomics (zero between steps)
  __shared__ int32_t s_acc[2][kN];
  __shared__ int32_t s_d[2][kN];                   // gadget_prepare((X^bara - 1) * acc) of the step about to run, both components
  __shared__ uint32_t s_steps[kCoop8MaxSteps + 1];   // the steps with bara != 0, in order: (i << 16) | bara
  __shared__ int s_step_count;
  stage_tables(s_tw, a.tw, 64 * G, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long ct = blockIdx.x;
  const Field f = a.f;
  double* buf = s_buf[wave];
  for (int e = threadIdx.x; e < 2 * kN; e += 64 * G) (&s_sum[0][0])[e] = 0.0;
  typename Xf::State tw;
  Xf::init(tw, lane, s_tw, a.tw);
  const int32_t* row0 = a.in0 + ct * a.W;
  const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
  const int n = a.n;
  // rows [first, first + cnt) of this wave's component (digit index q = first + rr, TGSW row comp * L + q)
  const int comp = coop8_comp(L, wave), cnt = coop8_row_count(L, wave), first = coop8_row_first(L, wave);
  [[maybe_unused]] const int r_first = cnt > 0 ? (int)(blockIdx.x % (unsigned)cnt) : 0;
  double dev = 0.0;
  RS_C8L_STAMP_DECL;   // -DRS_DIAG=256 (tools/stamp_coop8.py): 0 step entry + shared rotated difference, 1 its barrier, 2 rows (forward +
                      // multiply-accumulate), 3 atomics issued, 4 barrier 1, 5 inverse + accumulator update, 6 barrier 2, 7 prologue / extract
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
      s_acc[wave][j] = wave == 0 ? 0 : test_vector(a, ct, j, rot);
    }
  } else if (wave == 2) {
    // the step list: 64 mask words at a time, the non-zero ones compacted in order behind those of the chunks before
    int count = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const int32_t bara = i < n ? modswitch_2N(word(i)) : 0;
      const unsigned long long live = __ballot(bara != 0);
      if (bara != 0) s_steps[count + __popcll(live & ((1ull << lane) - 1ull))] = ((uint32_t)i << 16) | (uint32_t)bara;
      count += __popcll(live);
    }
    if (lane == 0) { s_steps[count] = 0u; s_step_count = count; }
  }
  __syncthreads();
  constexpr uint32_t offset = gadget_offset<C>();
  const int trips = __builtin_amdgcn_readfirstlane(s_step_count);
  uint32_t entry = trips > 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)s_steps[0]) : 0u;
  double2 w0[8], w1[8];
  auto request_row = [&](int step_i, int q) {
    const double* bk_i = a.bk_x + (size_t)diag::key_step(step_i) * KPL * 2 * kN;
    const double2* bp0 = reinterpret_cast<const double2*>(bk_i + (size_t)((comp * L + q) * 2) * kN);
    const double2* bp1 = bp0 + kN / 2;
#pragma unroll
    for (int v = 0; v < 8; ++v) { w0[v] = bp0[v * 64 + lane]; w1[v] = bp1[v * 64 + lane]; }
  };
  if (cnt > 0 && trips > 0) request_row((int)(entry >> 16), first + r_first);
  RS_C8L_STAMP(7);
  for (int k = 0; k < trips; ++k) {
    const int i = (int)(entry >> 16), bara = (int)(entry & 0xffffu);
    const uint32_t entry_next = s_steps[k + 1];   // requested a whole step ahead (the entry behind the last one exists: 0)
    // the prepared rotated difference of both components, 4 coefficients per thread (waves 0-3: component 0, waves 4-7: 1)
    {
      const int c = coop8_diff_comp(wave);
#pragma unroll
      for (int m = 0; m < kCoop8DiffPerThread; ++m) {
        const int j = coop8_diff_coeff((int)threadIdx.x, m);
        s_d[c][j] = gadget_prepare<C>(rotated_diff(s_acc[c], j, bara));
      }
    }
    RS_C8L_STAMP(0);
    __syncthreads();   // s_d complete; the accumulator is not read again before its update
    RS_C8L_STAMP(1);
    double s0[kRegs], s1[kRegs];
#pragma unroll
    for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
    if (cnt > 0) {
      int32_t d[kRegs];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) d[r] = s_d[comp][lane + 64 * r];
      int r_run = r_first;   // rr = 0 is row blockIdx.x mod cnt of the wave's share in every step (per-workgroup row order, as in the four-wave form)
#pragma unroll 1
      for (int rr = 0; rr < cnt; ++rr) {
        const int q = first + r_run;
        r_run = r_run + 1 == cnt ? 0 : r_run + 1;
        if (rr > 0) request_row(i, q);   // the first row's key is already on its way
        double x[kRegs];
        Xf::fwd_digits(lane, x, d, q, offset, tw, buf, f);
        Xf::mac8(s0, s1, x, w0, w1, f);
      }
      RS_C8L_STAMP(2);
    }
    // every wave adds its two partial sums into the column sums with ds_add_f64 (no return value: 32 instructions that overlap
    // the other waves' transforms); the order of the floating-point additions is free -- the sums are rounded to the exact
    // integers afterwards, with the certificate watching the distance as everywhere
    if (cnt > 0) {
#pragma unroll
      for (int u = 0; u < kRegs; ++u) { unsafeAtomicAdd(&s_sum[0][u * 64 + lane], s0[u]); unsafeAtomicAdd(&s_sum[1][u * 64 + lane], s1[u]); }
    }
    RS_C8L_STAMP(3);
    __syncthreads();   // sums complete; every row wave has finished reading the rotated difference
    RS_C8L_STAMP(4);
    entry = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry_next);
    if (cnt > 0 && k + 1 < trips) request_row((int)(entry >> 16), first + r_first);   // the next step's row, while the inverse waves work
    if (wave == kInvA || wave == kInvB) {
      double* sum = s_sum[wave == kInvA ? 0 : 1];
      double x[kRegs];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) x[u] = sum[u * 64 + lane];
#pragma unroll
      for (int u = 0; u < kRegs; ++u) sum[u * 64 + lane] = 0.0;   // for the next step (same lane, same address: in order)
      Xf::inverse(lane, x, tw, buf, f);
      int32_t* acc = s_acc[wave == kInvA ? 0 : 1];
#pragma unroll
      for (int r = 0; r < kRegs; ++r) {
        const int j = lane + 64 * r;
        acc[j] = (int32_t)((uint32_t)acc[j] + (uint32_t)Xf::to_torus(x[r], dev));
      }
    }
    RS_C8L_STAMP(5);
    __syncthreads();   // accumulator updated, sums zero
    RS_C8L_STAMP(6);
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
  RS_C8L_STAMP(7);
  RS_C8L_STAMP_FLUSH(wave);
  if (wave == kInvA || wave == kInvB) publish_certificate(dev, a.dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Debug tap: out = a_small * b_torus (negacyclic, mod 2^32) through forward/pointwise/inverse.
// -------------------------------------------------------------------------------------------------
template <class Xf, int WPB>
__global__ __launch_bounds__(64 * WPB) void polymul_kernel(const int32_t* __restrict__ a_small, const int32_t* __restrict__ b_torus,
                                                            int32_t* __restrict__ out, double* __restrict__ scratch,
                                                            const double* __restrict__ tw_g, Field f, double scale, long count,
                                                            unsigned long long* dev_flag) {
  __shared__ double s_tw[Xf::kTableDoubles + 1];
  __shared__ double s_buf[WPB][kBufDoubles];
  stage_tables(s_tw, tw_g, 64 * WPB, Xf::kTableDoubles);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const long idx = (long)blockIdx.x * WPB + wave;
  if (idx >= count) return;
  double* buf = s_buf[wave];
  typename Xf::State st;
  Xf::init(st, lane, s_tw, tw_g);
  double xa[kRegs], xb[kRegs];
#pragma unroll
  for (int r = 0; r < kRegs; ++r) {
    xa[r] = (double)a_small[idx * kN + lane + 64 * r];
    xb[r] = (double)b_torus[idx * kN + lane + 64 * r];
  }
  // key side exactly as bk_transform_kernel: through global memory in the key layout
  Xf::fwd_generic(lane, xb, st, buf, f);
  double2* key = reinterpret_cast<double2*>(scratch + idx * kN);
  Xf::key_store(key, lane, xb, scale, f);
  double2 wa[4], wb[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) { wa[v] = key[v * 64 + lane]; wb[v] = key[(v + 4) * 64 + lane]; }
  Xf::fwd_generic(lane, xa, st, buf, f);
  double s0[kRegs], s1[kRegs];
#pragma unroll
  for (int u = 0; u < kRegs; ++u) { s0[u] = 0.0; s1[u] = 0.0; }
  Xf::mac(s0, s1, xa, wa, wa, 0, f);
  Xf::mac(s0, s1, xa, wb, wb, 4, f);
  Xf::inverse(lane, s0, st, buf, f);
  double dev = 0.0;
#pragma unroll
  for (int r = 0; r < kRegs; ++r) out[idx * kN + lane + 64 * r] = Xf::to_torus(s0[r], dev);
  if (Xf::kCertificate) publish_certificate(dev, dev_flag, lane);
}

// -------------------------------------------------------------------------------------------------
// Launchers. cfg: 0 = default-128-shaped gadget, 1 = REDsec-shaped; mode: 0 = exact NTT, 1 = FFT.
// -------------------------------------------------------------------------------------------------
#if RS_BS_PART & 1
template <class Xf, int WPB>
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
  hipLaunchKernelGGL((blind_rotate_kernel<Xf, WPB>), dim3((unsigned)blocks), dim3(64 * WPB), 0, st, args);
  return hipGetLastError();
}

#endif  // RS_BS_PART & 1
// XCD cohorts of the lock-step kernels (cohort_step, rs_cohort.h): the ONLY place that hands a kernel a progress table. Every
// launch path starts from arguments whose `progress` is null and calls this for the launches that may use one: those whose
// workgroups sweep the key more than once (groups > grid), on a device whose workgroups are dealt round-robin over EIGHT XCDs --
// the protocol's xcd = blockIdx & 7. The one MI355X configuration that is true for is the whole chip as one partition (SPX,
// 256 CUs = 8 x 32); under CPX / DPX / QPX partitions a table row would mix workgroups served by different L2s, which
// could only wait for each other with no L2 to share, so there the workgroups run free. `step_bytes` = key bytes a CMUX step
// reads; the lag keeps a cohort inside about a third of its XCD's 4 MB L2.
static hipError_t cohort_setup(BlindRotateArgs& w, int* table, long step_bytes, long groups, long grid, long num_cus, const LaunchOpts& o, hipStream_t st) {
  w.progress = nullptr; w.cohort_every = 0; w.cohort_lag = 0;
  if (!table || o.no_cohort || num_cus != 256 || grid > 8L * kCohortSlots || groups <= grid) return hipSuccess;
  w.cohort_lag = (int32_t)std::max<long>(1, (4L << 20) / 3 / step_bytes - 1);
  w.cohort_every = w.cohort_lag >= 4 ? 2 : 1;
  w.progress = table;
  return hipMemsetAsync(table, 0x7f, 8 * kCohortSlots * sizeof(int), st);
}
#if RS_BS_PART & 1

hipError_t launch_coop8_listed(int cfg, const BlindRotateArgs& a, hipStream_t st);   // part 4

template <class Xf>
static hipError_t launch_br_xf(const BlindRotateArgs& a_in, int wpb, long num_cus, bool coop4, const LaunchOpts& o, hipStream_t st, LaunchInfo* info) {
  int* const cohort_table = a_in.progress;   // the caller's offer; only cohort_setup puts it back into a launch's arguments
  BlindRotateArgs a = a_in;
  a.progress = nullptr; a.cohort_every = 0; a.cohort_lag = 0;
  constexpr long kStepBytes = 2L * Xf::Cfg::L * 16384;   // a CMUX step reads 2l rows of 16 KB
  LaunchInfo li;
  auto done = [&](int form, int w, long resident) {
    li.form = form; li.waves_per_block = w; li.resident = resident;
    if (info) *info = li;
    return hipGetLastError();
  };
  // latency form: several waves per ciphertext while the batch cannot fill the chip by itself
  if (!o.no_coop) {
    if constexpr (Xf::kWorkgroupForm) {
      // at most one ciphertext per CU: eight waves share it (two per SIMD), see blind_rotate_coop8_kernel
      if (!o.no_coop8 && a.B <= num_cus) {
        if constexpr (coop8_listed(Xf::Cfg::L)) {   // its own object (RS_BS_PART bit 4): see there
          if (!o.no_coop8_listed && a.n <= kCoop8MaxSteps) {
            if (hipError_t e = launch_coop8_listed(std::is_same_v<typename Xf::Cfg, CfgDefault128> ? 0 : 1, a, st); e != hipSuccess) return e;
            return done(kFormCoop8Listed, 8, 1);
          }
        }
        hipLaunchKernelGGL((blind_rotate_coop8_kernel<Xf>), dim3((unsigned)a.B), dim3(512), 0, st, a);
        return done(kFormCoop8, 8, 1);
      }
    }
    if (coop4 && a.B <= num_cus) {
      if constexpr ((2 * Xf::Cfg::L) % 4 == 0) {
        hipLaunchKernelGGL((blind_rotate_coop_kernel<Xf, 4>), dim3((unsigned)a.B), dim3(256), 0, st, a);
        return done(kFormCoop4, 4, 1);
      }
    }
    if (a.B <= 2L * num_cus) {
      hipLaunchKernelGGL((blind_rotate_coop_kernel<Xf, 2>), dim3((unsigned)a.B), dim3(128), 0, st, a);
      return done(kFormCoop2, 2, 1);
    }
  }
  if constexpr (Xf::kWorkgroupForm) {
    // throughput form: lock-step workgroups of 8 ciphertexts, key rows shared in LDS. Taken as soon as the batch exceeds
    // FOUR ciphertexts per CU: a partly filled single round of it (10.2 ms for up to 2,048 default-128 ciphertexts) beats two
    // rounds of the half-size forms (12.8-13.1 ms at 1,536; tools/midsize_rate.py).
    if (!o.no_wg && a.B > 4L * num_cus) {
      // The grid walks the batch in rounds of 8 x #CUs ciphertexts. A last round of at most 4 x #CUs of them is cut off
      // and runs in the form that batch size would take by itself (cooperative / duo / half-size groups: 3.1-8.2 ms against
      // 14.8 ms for a whole round of the REDsec set, tools/midsize_rate.py); every ciphertext is independent of the split.
      const long cap = 8L * num_cus, tail = a.B % cap;
      if (!o.no_tail && a.B > cap && tail > 0 && tail <= 4L * num_cus) {
        BlindRotateArgs m = a, t = a;
        m.B = a.B - tail;
        t.B = tail;
        t.in0 = a.in0 + m.B * a.W;
        if (a.in1) t.in1 = a.in1 + m.B * a.W;
        t.u_out = a.u_out + m.B * (kN + 1);
        if (a.lut) t.lut_first = (int32_t)((a.lut_first + m.B) % a.lut_count);
        if (hipError_t ce = cohort_setup(m, cohort_table, kStepBytes, m.B / 8, num_cus, num_cus, o, st); ce != hipSuccess) return ce;
        hipLaunchKernelGGL((blind_rotate_wg_kernel<Xf, 8>), dim3((unsigned)num_cus), dim3(512), 0, st, m);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;   // (t, the cut-off last round, runs in a form without cohorts: its progress is null)
        e = launch_br_xf<Xf>(t, wpb, num_cus, coop4, o, st, nullptr);
        if (e != hipSuccess) return e;
        return done(kFormWorkgroup, 8, 8 * num_cus);
      }
      const long groups = (a.B + 7) / 8;
      const long grid = groups < num_cus ? groups : num_cus;
      BlindRotateArgs w = a;
      if (hipError_t e = cohort_setup(w, cohort_table, kStepBytes, groups, grid, num_cus, o, st); e != hipSuccess) return e;
      hipLaunchKernelGGL((blind_rotate_wg_kernel<Xf, 8>), dim3((unsigned)grid), dim3(512), 0, st, w);
      return done(kFormWorkgroup, 8, 8 * grid);   // the workgroups sweep the key together: 8 x grid ciphertexts per sweep
    }
  }
  if constexpr (Xf::kWorkgroupForm && Xf::Cfg::L % 2 != 0) {
    // odd l (no duo form), 2 x #CUs < B <= 4 x #CUs: half-size lock-step groups, 4 ciphertexts x 1 wave per workgroup = one
    // wave per SIMD, which delivers 78 % of the full form's rate per CU and shares the key rows (2-3 % faster than the
    // per-wave kernel, which streams them per wave)
    if (!o.no_wg && !o.no_wg4 && a.B > 2L * num_cus) {
      const long groups = (a.B + 3) / 4;
      const long grid = groups < num_cus ? groups : num_cus;
      hipLaunchKernelGGL((blind_rotate_wg_kernel<Xf, 4>), dim3((unsigned)grid), dim3(256), 0, st, a);
      return done(kFormWorkgroup, 4, 4 * grid);
    }
  }
  if constexpr (Xf::kWorkgroupForm && Xf::Cfg::L % 2 == 0) {
    // mid-size batches: 4 ciphertexts x 2 waves per workgroup (no_duo falls back to one wave per ciphertext)
    if (!o.no_wg && !o.no_duo && a.B > 2L * num_cus) {
      const long groups = (a.B + 3) / 4;
      const long grid = groups < num_cus ? groups : num_cus;
      hipLaunchKernelGGL((blind_rotate_duo_kernel<Xf>), dim3((unsigned)grid), dim3(512), 0, st, a);
      return done(kFormDuo, 8, 4 * grid);
    }
  }
  hipError_t e;
  switch (wpb) {
    case 1: e = launch_br<Xf, 1>(a, 1L << 40, st); break;
    case 2: e = launch_br<Xf, 2>(a, 1L << 40, st); break;
    case 4: e = launch_br<Xf, 4>(a, 1L << 40, st); break;
    default: wpb = 8; e = launch_br<Xf, 8>(a, num_cus, st); break;   // ~150 KB LDS: exactly one workgroup per CU
  }
  const long waves = a.B < (long)wpb * num_cus ? a.B : (long)wpb * num_cus;
  li.form = kFormPerWave; li.waves_per_block = wpb; li.resident = waves;   // waves drift apart: an upper bound on key sharing
  if (info) *info = li;
  return e;
}

hipError_t launch_blind_rotate(int cfg, int mode, const BlindRotateArgs& a, int wpb, int num_cus, const LaunchOpts& opts, hipStream_t st,
                               LaunchInfo* info) {
  if (a.B <= 0) return hipSuccess;
  if (mode == 0) {
    return cfg == 0 ? launch_br_xf<XfNtt<CfgDefault128>>(a, wpb, num_cus, false, opts, st, info)
                    : launch_br_xf<XfNtt<CfgRedsecV2>>(a, wpb, num_cus, true, opts, st, info);
  }
  return cfg == 0 ? launch_br_xf<XfFft<CfgDefault128>>(a, wpb, num_cus, false, opts, st, info)
                  : launch_br_xf<XfFft<CfgRedsecV2>>(a, wpb, num_cus, true, opts, st, info);
}

// The split duo form's launch (mid-size batches of the split mode): lives in part 1, called from part 2.
hipError_t launch_split_duos(int cfg, const BlindRotateArgs& a, long grid, hipStream_t st) {
  if (cfg == 0) hipLaunchKernelGGL((blind_rotate_duos_kernel<CfgDefault128>), dim3((unsigned)grid), dim3(512), 0, st, a);
  else if (cfg == 1) hipLaunchKernelGGL((blind_rotate_duos_kernel<CfgRedsecV2>), dim3((unsigned)grid), dim3(512), 0, st, a);
  else if (cfg == 2) hipLaunchKernelGGL((blind_rotate_duos_kernel<CfgRedsecSmall>), dim3((unsigned)grid), dim3(512), 0, st, a);
  else return hipErrorNotSupported;
  return hipGetLastError();
}
#endif  // RS_BS_PART & 1

#if RS_BS_PART & 4
// blind_rotate_coop8_listed_kernel lives in an object of its own (built with part 1's flags). Instantiated beside the other FFT
// kernels it changed THEIR code -- 15 of part 1's device functions came out a few instructions different, the REDsec set's
// coop8 kernel among them (tools/codeobj_digest.py), with the source of none of them touched -- and the kernels of the BASELINE
// configurations are to stay the instructions that were measured. cfg: 0 default-128, 1 the REDsec set (no listed deal).
hipError_t launch_coop8_listed(int cfg, const BlindRotateArgs& a, hipStream_t st) {
  if constexpr (coop8_listed(CfgDefault128::L)) {
    if (cfg == 0) {
      hipLaunchKernelGGL((blind_rotate_coop8_listed_kernel<XfFft<CfgDefault128>>), dim3((unsigned)a.B), dim3(512), 0, st, a);
      return hipGetLastError();
    }
  }
  return hipErrorInvalidValue;
}
#endif  // RS_BS_PART & 4

#if RS_BS_PART & 2
hipError_t launch_split_duos(int cfg, const BlindRotateArgs& a, long grid, hipStream_t st);
// Split-key workgroup form (N = 1024; cfg 0 / 1 = the two shipped gadgets, 2 = redsec_params_small's l=3 Bgbit=10): a.bk_x = the split key of rs_general.h,
// a.tw = the FFT tables of rs_fft.h. Returns hipErrorNotSupported for an unknown gadget id (caller: general kernel).
hipError_t launch_blind_rotate_split_wg(int cfg, const BlindRotateArgs& a_in, int num_cus, const LaunchOpts& o, hipStream_t st, LaunchInfo* info) {
  int* const cohort_table = a_in.progress;   // the caller's offer; only cohort_setup puts it back into a launch's arguments
  BlindRotateArgs a = a_in;
  a.progress = nullptr; a.cohort_every = 0; a.cohort_lag = 0;
  // any batch size: even a single group of it walks its CMUX chain in 57 us per step (REDsec set) against the 92 us of a lone
  // wave of the general kernel (sign1024x1 in split mode: 65.7 -> 40 ms). Up to 4 ciphertexts per CU the groups are 4 waves:
  // one wave per SIMD on twice the CUs.
  auto coop = [&](auto c) {
    using C = decltype(c);
    LaunchInfo li;
    li.form = kFormSplitCoop; li.resident = 1;

    if constexpr ((2 * C::L) % 4 == 0) {
      if (a.B <= num_cus) {
        hipLaunchKernelGGL((blind_rotate_coops_kernel<C, 4>), dim3((unsigned)a.B), dim3(256), 0, st, a);
        li.waves_per_block = 4;
        if (info) *info = li;
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((blind_rotate_coops_kernel<C, 2>), dim3((unsigned)a.B), dim3(128), 0, st, a);
    li.waves_per_block = 2;
    if (info) *info = li;
    return hipGetLastError();
  };
  if (!o.no_coop && a.B <= 2L * num_cus) {   // latency form: several waves per ciphertext, as in the unsplit modes
    if (cfg == 0) return coop(CfgDefault128{});
    if (cfg == 1) return coop(CfgRedsecV2{});
    if (cfg == 2) return coop(CfgRedsecSmall{});
    return hipErrorNotSupported;
  }
  if (!o.no_duo && a.B <= 4L * num_cus) {   // mid-size batches: 4 ciphertexts x 2 waves per workgroup (no_duo: the 4-wave lock-step groups)
    const long grid = std::min<long>((a.B + 3) / 4, num_cus);
    if (cfg < 0 || cfg > 2) return hipErrorNotSupported;
    if (info) { info->form = kFormSplitDuo; info->waves_per_block = 8; info->resident = 4 * grid; }
    return launch_split_duos(cfg, a, grid, st);
  }
  const int wpb = (a.B <= 4L * num_cus && !o.no_wg4) ? 4 : 8;
  const long groups = (a.B + wpb - 1) / wpb;
  const long grid = groups < num_cus ? groups : num_cus;
  BlindRotateArgs w = a;
  // a CMUX step reads 2 * 2l half-rows of 16 KB
  if (hipError_t e = cohort_setup(w, cohort_table, 4L * (cfg == 1 ? 10 : 3) * 16384, groups, grid, num_cus, o, st); e != hipSuccess) return e;
  auto go = [&](auto c) {
    using C = decltype(c);
    if (wpb == 8) hipLaunchKernelGGL((blind_rotate_wgs_kernel<C, 8>), dim3((unsigned)grid), dim3(512), 0, st, w);
    else hipLaunchKernelGGL((blind_rotate_wgs_kernel<C, 4>), dim3((unsigned)grid), dim3(256), 0, st, w);
  };
  if (cfg == 0) go(CfgDefault128{});
  else if (cfg == 1) go(CfgRedsecV2{});
  else if (cfg == 2) go(CfgRedsecSmall{});
  else return hipErrorNotSupported;
  if (info) { info->form = kFormSplitWorkgroup; info->waves_per_block = wpb; info->resident = wpb * grid; }
  return hipGetLastError();
}

#endif  // RS_BS_PART & 2

#if RS_BS_PART & 1
hipError_t launch_bk_transform(int cfg, int mode, const int32_t* bk, double* bk_x, const double* tw, Field f, double scale,
                               long n_polys, hipStream_t st) {
  constexpr int WPB = 4;
  const dim3 grid((unsigned)((n_polys + WPB - 1) / WPB)), block(64 * WPB);
  if (mode == 1) {
    hipLaunchKernelGGL((bk_transform_kernel<XfFft<CfgDefault128>, WPB>), grid, block, 0, st, bk, bk_x, tw, f, scale, n_polys);
  } else if (cfg == 0) {
    hipLaunchKernelGGL((bk_transform_kernel<XfNtt<CfgDefault128>, WPB>), grid, block, 0, st, bk, bk_x, tw, f, scale, n_polys);
  } else {
    hipLaunchKernelGGL((bk_transform_kernel<XfNtt<CfgRedsecV2>, WPB>), grid, block, 0, st, bk, bk_x, tw, f, scale, n_polys);
  }
  return hipGetLastError();
}

hipError_t launch_polymul(int cfg, int mode, const int32_t* a_small, const int32_t* b_torus, int32_t* out, double* scratch,
                          const double* tw, Field f, double scale, long count, unsigned long long* dev_flag, hipStream_t st) {
  constexpr int WPB = 4;
  const dim3 grid((unsigned)((count + WPB - 1) / WPB)), block(64 * WPB);
  if (mode == 1) {
    hipLaunchKernelGGL((polymul_kernel<XfFft<CfgDefault128>, WPB>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, f, scale, count, dev_flag);
  } else if (cfg == 0) {
    hipLaunchKernelGGL((polymul_kernel<XfNtt<CfgDefault128>, WPB>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, f, scale, count, dev_flag);
  } else {
    hipLaunchKernelGGL((polymul_kernel<XfNtt<CfgRedsecV2>, WPB>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, f, scale, count, dev_flag);
  }
  return hipGetLastError();
}

#endif  // RS_BS_PART & 1

}  // namespace rs

#if RS_BS_PART & RS_DIAG_STAMP_PART
// diagnostic builds only (not part of include/redsec_hip.h): copy the per-wave phase sums to the host and clear them
extern "C" int rs_debug_read_stamps(unsigned long long* host, size_t count) {
  const size_t all = sizeof(rs::g_rs_stamps) / sizeof(unsigned long long);
  if (count > all) count = all;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(rs::g_rs_stamps), count * sizeof(unsigned long long)) != hipSuccess) return 1;
  static unsigned long long zeros[256 * 8 * rs::diag::kStampPhases];
  return hipMemcpyToSymbol(HIP_SYMBOL(rs::g_rs_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
}
#endif
