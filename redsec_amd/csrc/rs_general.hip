// rs_general.hip -- kernels of the general ring path (rs_general.h): N in {1024, 2048, 4096, 8192}, any gadget,
// key split into two 16-bit halves so that the FP64 FFT product is exact by an a-priori bound (RS_MODE_FFT_SPLIT).
//
//   gen_bk_transform_kernel    key polynomial -> two transformed half polynomials (the bkFFT analogue)
//   gen_blind_rotate_kernel    gate pre-combination + modswitch + n CMUX steps + sample extract
//                              (tfhe_bootstrap_woKS_FFT / tfhe_blindRotateAndExtract_FFT; REDsec: lib/BinOps_enc.cpp:185,191)
//   gen_polymul_kernel         debug/parity tap through the same transform path
//
// One workgroup of T = N/16 threads owns one ciphertext: its TRLWE accumulator (2 x N int32) and the two exchange
// planes live in LDS (17 N bytes), every thread keeps 8 complex values of the transform in flight and the
// 2 halves x 2 columns of pointwise sums in registers. Workgroups are persistent over the batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "rs_diag.h"
#include "rs_general.h"
#include "rs_kernels.h"

namespace rs {

namespace {

__device__ __forceinline__ void gen_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int T>
__device__ __forceinline__ void gen_sync() {
  if constexpr (T == 64) gen_wave_sync(); else __syncthreads();
}

// key loads as SGPR base + 32-bit lane offset (global_load ... v_off, s[base:base+1]): the row pointer is the same in every
// lane, and left as a 64-bit per-lane pointer the compiler kept a VGPR pair and two adds per (half, column, position) stream
// The thread index as an opaque value: address arithmetic that depends on it is then recomputed where it is used (a few shifts
// and adds) instead of being hoisted out of the CMUX loop and held -- or spilled -- for the whole kernel.
__device__ __forceinline__ int gen_local(int t) {
  asm volatile("" : "+v"(t));
  return t;
}
typedef double GenD2 __attribute__((ext_vector_type(2)));
typedef const GenD2 __attribute__((address_space(1)))* GenKeyPtr;   // global address space kept through the integer round trip
__device__ __forceinline__ GenKeyPtr gen_uniform_ptr(const double2* p) {
  const uint64_t b = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (GenKeyPtr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ GenKeyPtr gen_uniform_ptr(GenKeyPtr p) {   // of an already uniform value: folds to the scalar itself
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return (GenKeyPtr)(((uint64_t)hi << 32) | lo);
}
// `base` already includes the (half, column, position) displacement; made opaque here so that the optimiser does not move
// that constant over to the lane offset (which turns every stream back into a 64-bit per-lane pointer)
__device__ __forceinline__ double2 gen_key_load(GenKeyPtr base, uint32_t lane_bytes) {
  typedef const char __attribute__((address_space(1)))* BytePtr;
  asm volatile("" : "+v"(lane_bytes));   // not hoisted, so that base + offset is selected as ONE load with a scalar base
  const GenD2 v = *(GenKeyPtr)((BytePtr)gen_uniform_ptr(base) + lane_bytes);
  return make_double2(v.x, v.y);
}

__device__ __forceinline__ void gen_publish(double dev, unsigned long long* flag) {
  if (!flag) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(dev, off, 64);
    dev = o > dev ? o : dev;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(flag, (unsigned long long)__double_as_longlong(dev));
}

}  // namespace

// Levels 0..8 of the twiddle table staged in LDS (8 KB; rs_general.h, gen_pass_tw, kGenStageTw). Measured at full size
// (profiles/r03/f_general_ab_twiddles_in_lds.jsonl): redsec_params_medium +8.9 % (two workgroups per CU, every pass used to wait
// for an L1/L2 round trip of its own), redsec_params_large -3.9 % (one 512-thread workgroup per CU on 139 KB of LDS already: the
// extra LDS reads queue behind its exchanges) -- so N = 8192 keeps reading the table from global memory.
template <int LOGN>
__device__ __forceinline__ const double* gen_stage_twiddles(double* s_twn, const double* tw, int t) {
  if constexpr (kGenStageTw<LOGN>) {
    for (int i = t; i < 2 * kGenTwLds; i += Gen<LOGN>::T) s_twn[i] = tw[i];
    __syncthreads();
    return s_twn;
  } else {
    return tw;
  }
}

template <int LOGN>
__global__ __launch_bounds__(Gen<LOGN>::T) void gen_bk_transform_kernel(const int32_t* __restrict__ bk, double* __restrict__ bk_x,
                                                                         const double* __restrict__ tw, long n_polys) {
  using G = Gen<LOGN>;
  constexpr int N = G::N, M = G::M, T = G::T;
  __shared__ double s_re[G::kPlane], s_im[G::kPlane];
  __shared__ double s_twn[kGenStageTw<LOGN> ? 2 * kGenTwLds : 2];
  const int t = threadIdx.x;
  auto sync = [] { gen_sync<T>(); };
  auto wsync = [] { gen_wave_sync(); };
  const double* twn = gen_stage_twiddles<LOGN>(s_twn, tw, t);
  for (long poly = blockIdx.x; poly < n_polys; poly += gridDim.x) {
    const int32_t* src = bk + poly * N;
#pragma unroll 1
    for (int piece = 0; piece < 2; ++piece) {
      double x[kRegs];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int32_t lo, hi;
        gen_split_key(src[t + T * r], lo, hi);
        x[r] = (double)(piece ? hi : lo);
        gen_split_key(src[t + T * r + M], lo, hi);
        x[r + 8] = (double)(piece ? hi : lo);
      }
      gen_fft_fwd<LOGN>(x, t, tw, twn, s_re, s_im, sync, wsync);
      // [row pair index][piece][column][8][T] complex, scaled by 1/M (a power of two: exact)
      double2* dst = reinterpret_cast<double2*>(bk_x) + ((size_t)(poly >> 1) * 4 + (size_t)piece * 2 + (size_t)(poly & 1)) * M;
#pragma unroll
      for (int r = 0; r < 8; ++r) dst[r * T + t] = make_double2(x[r] * (1.0 / M), x[r + 8] * (1.0 / M));
    }
  }
}

// Choices of gen_blind_rotate_kernel, each from a same-box A/B (profiles/r03/f_*, h_*, profiles/r04/x_*; MEASUREMENTS.md section 4.5): two key
// positions requested ahead of the one being multiplied (three spill and lose); a row's first key position requested in front of its
// transform's last exchange (not in front of the transform, behind it, or behind the last exchange); the two inverse transforms of a
// column as a pipelined pair; no whole-row prefetch at the start of a step (1.2-1.5 % slower); no de-phasing of an XCD's workgroups
// at kernel start (no effect). Diagnostic builds (rs_diag.h, RS_DIAG bit 16) replace every key load by register values.
// A key chunk of the row at `p`; under the no-key probe a value made of registers instead (no load at all; results wrong).
__device__ __forceinline__ double2 gen_key_or_probe(GenKeyPtr p, uint32_t tb, double pa, double pb) {
  if constexpr (diag::kNoKeyProbe) { (void)p; (void)tb; return make_double2(pa, pb); }
  else { (void)pa; (void)pb; return gen_key_load(p, tb); }
}
template <int LOGN>
__global__ __launch_bounds__(Gen<LOGN>::T) __attribute__((amdgpu_waves_per_eu(2, 2))) void gen_blind_rotate_kernel(GenArgs a) {
  using G = Gen<LOGN>;
  constexpr int N = G::N, M = G::M, T = G::T;
  __shared__ double s_re[G::kPlane], s_im[G::kPlane];
  __shared__ int32_t s_acc[2][N];
  __shared__ double s_twn[kGenStageTw<LOGN> ? 2 * kGenTwLds : 2];
  const int t = threadIdx.x;
  auto sync = [] { gen_sync<T>(); };
  auto wsync = [] { gen_wave_sync(); };
  const double* twn = gen_stage_twiddles<LOGN>(s_twn, a.tw, t);
  const int l = a.l, bgbit = a.bgbit, n = a.n;
  // The LAST pass's twiddles of this thread stay in registers for the whole kernel where that pass reads the table from GLOBAL
  // memory (N >= 2048: its levels lie above the LDS-staged ones): same-box +3.7 % at N = 4096 and +3.4 % at N = 8192 -- those loads
  // were requested one exchange ahead of their use and mostly waited for (profiles/r03/h_general_ab_last_pass_twiddles_kept.txt,
  // h_general_ab_which_pass_kept.txt). Keeping an LDS-served pass costs more in spilled registers than its reads save (pass 1 or 2
  // at N = 4096: -0.7 % / -3.8 %).
  constexpr bool kLastPassGlobal = !kGenStageTw<LOGN> || (2 << (Gen<LOGN>::LOGM - 1)) > kGenTwLds;
  constexpr int kKeepPass = (Gen<LOGN>::P >= 2 && kLastPassGlobal) ? Gen<LOGN>::P - 1 : -1;
  GenPassTw tw_kept;
  if constexpr (kKeepPass == 1) gen_pass_tw<LOGN, 1>(tw_kept, t, a.tw, twn);
  if constexpr (kKeepPass == 2 && Gen<LOGN>::P > 2) gen_pass_tw<LOGN, 2>(tw_kept, t, a.tw, twn);
  if constexpr (kKeepPass == 3 && Gen<LOGN>::P > 3) gen_pass_tw<LOGN, 3>(tw_kept, t, a.tw, twn);
  const GenKeptTw kept1{kKeepPass >= 1 && kKeepPass < Gen<LOGN>::P ? &tw_kept : nullptr, kKeepPass};
  const uint32_t goff = gen_gadget_offset(l, bgbit);
  double dev = 0.0;

  // (XCD cohorts, rs_cohort.h, were tried here in round 4 -- wave 0 posting the step count and waiting for its XCD's slowest workgroup -- and
  // cost 7 % at N = 4096 and 3 % at N = 8192 whether switched on or off: the few registers of the cohort state spill 10-20 more dwords per lane in
  // kernels that were full (profiles/r04/x_ab_general_cohorts_and_xcd_rotation.txt). The fabric traffic of these kernels sits at the 8-XCD
  // floor without them.)
  for (long ct = blockIdx.x; ct < a.B; ct += gridDim.x) {
    const int32_t* row0 = a.in0 + ct * a.W;
    const int32_t* row1 = a.in1 ? a.in1 + ct * a.W : nullptr;
    // gate pre-combination (0, bconst) + c0*in0 + c1*in1, evaluated word by word as it is consumed
    auto word = [&](int i) -> int32_t {
      uint32_t v = (uint32_t)a.c0 * (uint32_t)row0[i];
      if (row1) v += (uint32_t)a.c1 * (uint32_t)row1[i];
      return (int32_t)v;
    };
    {
      const int32_t barb = gen_modswitch((int32_t)((uint32_t)word(n) + (uint32_t)a.bconst), LOGN);
      const int rot = 2 * N - barb;   // in (0, 2N]
      const int32_t* lut = a.lut ? a.lut + (size_t)((ct + a.lut_first) % a.lut_count) * N : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = t + T * (r & 7) + (r >> 3) * M;
        s_acc[0][j] = 0;
        if (!lut) {
          s_acc[1][j] = gen_rotated_const(a.mu, j, rot, LOGN);
        } else {   // (X^rot * lut)_j: the ciphertext's own test polynomial (tfhe_blindRotateAndExtract_FFT)
          const int aa = rot & (N - 1), nb = (rot >> LOGN) & 1;
          const uint32_t x = (uint32_t)lut[(j - aa) & (N - 1)];
          s_acc[1][j] = (int32_t)((((j < aa) ? 1 : 0) ^ nb) ? 0u - x : x);
        }
      }
    }
    sync();

    int32_t bara_next = gen_modswitch(word(0), LOGN);
    for (int i = 0; i < n; ++i) {
      const int32_t bara = __builtin_amdgcn_readfirstlane(bara_next);   // the same word in every thread
      bara_next = (i + 1 < n) ? gen_modswitch(word(i + 1), LOGN) : 0;
      if (bara == 0) continue;   // tfhe_blindRotate_FFT skips the identity CMUX (uniform over the workgroup)
      double S[2][2][kRegs];     // [key half][column]
      const uint32_t tb = (uint32_t)t * (uint32_t)sizeof(double2);
      // prepared rotated difference of one component, shared by its l digit rows
      auto prep = [&](int comp, int32_t (&v)[kRegs]) {
          // all 32 accumulator words first, then the arithmetic: left to itself the compiler read them one at a time, each
          // behind a full s_waitcnt (24 exposed LDS round trips per component; the ISA timeline of profiles/r03 shows it)
          const int32_t* accc = s_acc[comp];
          const int aa = bara & (N - 1), nb = (bara >> LOGN) & 1;
          uint32_t rot[kRegs], own[kRegs];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = t + T * (r & 7) + (r >> 3) * M;
            rot[r] = (uint32_t)accc[(j - aa) & (N - 1)];
            own[r] = (uint32_t)accc[j];
          }
          gen_wave_sync();   // compiler-only: keep the reads together, ahead of their uses
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = t + T * (r & 7) + (r >> 3) * M;
            const int neg = (j < aa ? 1 : 0) ^ nb;
            v[r] = gen_gadget_prepare((int32_t)((neg ? (0u - rot[r]) : rot[r]) - own[r]), goff);   // gen_rotated_diff, same arithmetic
          }
      };
      // one digit row: transform, then both key halves x both columns multiplied into S
      auto row = [&](int comp, int q, const int32_t (&v)[kRegs]) {
          double x[kRegs];
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = (double)gen_gadget_digit(v[r], q, bgbit);
          GenKeyPtr kp = gen_uniform_ptr(reinterpret_cast<const double2*>(a.bk_x) + ((size_t)i * 2 * l + (size_t)comp * l + q) * 4 * M);
          constexpr int LA = 2, NB = LA + 1;   // key positions requested ahead of the one being multiplied
          double2 w[NB][4];
          // both halves x both columns of position r; position 0 is requested in front of the transform's last exchange, position
          // r + LA before the FMAs of position r (the compiler's own schedule waited for each group of four in full before its 16
          // FMAs: eight exposed L2 round trips per row, most of a CMUX step on the large rings)
          constexpr int kFirstGroups = LOGN >= 13 ? 2 : 1;   // N = 8192 (with the row rotation): 458.5 -> 445.7 ms; N = 4096: +0.4 % at best
          constexpr int FG = kFirstGroups < LA ? kFirstGroups : LA;   // positions requested at the tail hook
          auto first = [&] {
#pragma unroll
            for (int g = 0; g < FG; ++g) {
#pragma unroll
              for (int hc = 0; hc < 4; ++hc) {
                w[g][hc] = gen_key_or_probe(kp + (size_t)hc * M + g * T, tb, 1.0 + hc, 2.0 + g);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          };
          gen_fft_fwd<LOGN>(x, gen_local(t), a.tw, twn, s_re, s_im, sync, wsync, first, kept1);
#pragma unroll
          for (int g = FG; g < LA; ++g) {
#pragma unroll
            for (int hc = 0; hc < 4; ++hc) {
              w[g][hc] = gen_key_or_probe(kp + (size_t)hc * M + g * T, tb, x[hc + g], x[hc + 8]);
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            if (r + LA < (diag::kHalfKeyProbe ? 4 : 8)) {
#pragma unroll
              for (int hc = 0; hc < 4; ++hc) {
                w[(r + LA) % NB][hc] = gen_key_or_probe(kp + (size_t)hc * M + (r + LA) * T, tb, x[hc + 1], x[hc + 4]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hc = 0; hc < 4; ++hc) fft_cmac(S[hc >> 1][hc & 1][r], S[hc >> 1][hc & 1][r + 8], x[r], x[r + 8], w[r % NB][hc].x, w[r % NB][hc].y);
            __builtin_amdgcn_sched_barrier(0);
          }
      };
      int32_t v[kRegs];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int u = 0; u < kRegs; ++u) S[h][c][u] = 0.0;
      // Workgroup b walks the 2 l rows of a step in its own order (the rows are independent; the sums are exact in any order): started
      // together, all workgroups otherwise pull the same key rows through the same L2 channels at the same moments. Mode 1 (digit
      // rotation + component order from b): N = 8192 516.5 -> 458.8 ms per 512 (+12.6 %); at N = 4096 digit rotation costs 8-10 %
      // (two workgroups per CU and 512 per launch lose more L2 locality than they gain), component order alone +0.4 %
      // (profiles/r03/h_general_ab_row_rotation_modes.txt). Modes: 0 none, 1 both, 2 digits, 3 components.
      {
        constexpr int kRot = LOGN >= 13 ? 1 : 3;
        // (round 4: the order taken from the XCD, blockIdx.x & 7, instead of the workgroup -- same rows at the same time inside an L2, different
        // rows in different XCDs -- measured: N = 4096 +-0, N = 8192 -3.7 % against mode 1; profiles/r04/x_ab_general_cohorts_and_xcd_rotation.txt)
        const int qrot = (kRot == 1 || kRot == 2) ? (int)(blockIdx.x % (unsigned)l) : 0;
        const int cfirst = kRot == 1 ? (int)((blockIdx.x / (unsigned)l) & 1u) : (kRot == 3 ? (int)(blockIdx.x & 1u) : 0);
#pragma unroll 1
        for (int cc = 0; cc < 2; ++cc) {
          const int comp = cc ^ cfirst;
          prep(comp, v);
#pragma unroll 1
          for (int qi = 0; qi < l; ++qi) {
            int q = qi + qrot;
            if (q >= l) q -= l;
            row(comp, q, v);
          }
        }
      }

      // every thread passed at least one barrier since its reads of the accumulator: the update cannot overtake them
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        gen_fft_inv2<LOGN>(S[0][c], S[1][c], gen_local(t), a.tw, twn, s_re, s_im, sync, wsync, kept1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = t + T * (r & 7) + (r >> 3) * M;
          const uint32_t lo = (uint32_t)fft_round_torus32(S[0][c][r], dev);
          const uint32_t hi = (uint32_t)fft_round_torus32(S[1][c][r], dev);
          s_acc[c][j] = (int32_t)((uint32_t)s_acc[c][j] + lo + (hi << 16));
        }
      }
      sync();
    }

    // tLweExtractLweSampleIndex(index 0): a'[0] = acc_a[0], a'[j] = -acc_a[N-j], b' = acc_b[0]
    int32_t* out = a.u_out + ct * (N + 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = t + T * (r & 7) + (r >> 3) * M;
      out[j] = (j == 0) ? s_acc[0][0] : (int32_t)(0u - (uint32_t)s_acc[0][N - j]);
    }
    if (t == 0) out[N] = s_acc[1][0];
    sync();   // the accumulator is re-initialised by the next ciphertext
  }
  gen_publish(dev, a.dev_flag);
}

// out = a_small * b_torus (negacyclic, mod 2^32) through key split, forward, pointwise, inverse, recombination
template <int LOGN>
__global__ __launch_bounds__(Gen<LOGN>::T) void gen_polymul_kernel(const int32_t* __restrict__ a_small, const int32_t* __restrict__ b_torus,
                                                                    int32_t* __restrict__ out, double* __restrict__ scratch,
                                                                    const double* __restrict__ tw, long count, unsigned long long* dev_flag) {
  using G = Gen<LOGN>;
  constexpr int N = G::N, M = G::M, T = G::T;
  __shared__ double s_re[G::kPlane], s_im[G::kPlane];
  __shared__ double s_twn[kGenStageTw<LOGN> ? 2 * kGenTwLds : 2];
  const int t = threadIdx.x;
  auto sync = [] { gen_sync<T>(); };
  auto wsync = [] { gen_wave_sync(); };
  const double* twn = gen_stage_twiddles<LOGN>(s_twn, tw, t);
  double dev = 0.0;
  for (long idx = blockIdx.x; idx < count; idx += gridDim.x) {
    const int32_t* pa = a_small + idx * N;
    const int32_t* pb = b_torus + idx * N;
    double2* key = reinterpret_cast<double2*>(scratch) + (size_t)idx * 2 * M;   // the two halves, in the key layout
#pragma unroll 1
    for (int piece = 0; piece < 2; ++piece) {
      double x[kRegs];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int32_t lo, hi;
        gen_split_key(pb[t + T * (r & 7) + (r >> 3) * M], lo, hi);
        x[r] = (double)(piece ? hi : lo);
      }
      gen_fft_fwd<LOGN>(x, t, tw, twn, s_re, s_im, sync, wsync);
#pragma unroll
      for (int r = 0; r < 8; ++r) key[(size_t)piece * M + r * T + t] = make_double2(x[r] * (1.0 / M), x[r + 8] * (1.0 / M));
    }
    double xa[kRegs];
#pragma unroll
    for (int r = 0; r < 16; ++r) xa[r] = (double)pa[t + T * (r & 7) + (r >> 3) * M];
    gen_fft_fwd<LOGN>(xa, t, tw, twn, s_re, s_im, sync, wsync);
    double S[2][kRegs];
#pragma unroll
    for (int piece = 0; piece < 2; ++piece) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const double2 w = key[(size_t)piece * M + r * T + t];   // this thread's own stores
        S[piece][r] = 0.0; S[piece][r + 8] = 0.0;
        fft_cmac(S[piece][r], S[piece][r + 8], xa[r], xa[r + 8], w.x, w.y);
      }
    }
    gen_fft_inv<LOGN>(S[0], t, tw, twn, s_re, s_im, sync, wsync);
    gen_fft_inv<LOGN>(S[1], t, tw, twn, s_re, s_im, sync, wsync);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = t + T * (r & 7) + (r >> 3) * M;
      const uint32_t lo = (uint32_t)fft_round_torus32(S[0][r], dev);
      const uint32_t hi = (uint32_t)fft_round_torus32(S[1][r], dev);
      out[idx * N + j] = (int32_t)(lo + (hi << 16));
    }
    sync();
  }
  gen_publish(dev, dev_flag);
}

// ---- launchers ----
// persistent grid: as many workgroups as the device keeps resident (asked from the runtime once per kernel)
template <class K>
static long gen_resident(K kernel, int threads, int num_cus) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return (long)per_cu * num_cus;
}
// resident workgroups of the CURRENT device (contexts of a fleet may sit on devices or partitions with different CU counts):
// cached per (device, ring), filled at first use; a benign race writes the same value twice
static long gen_grid(int logn, long work, int num_cus) {
  static std::atomic<long> resident[16][kGenMaxLogN + 1];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<long>& slot = resident[dev & 15][logn];
  long r = slot.load(std::memory_order_relaxed);
  if (r == 0 || dev > 15) {
    const int T = (1 << logn) / 16;
    switch (logn) {
      case 10: r = gen_resident(gen_blind_rotate_kernel<10>, T, num_cus); break;
      case 11: r = gen_resident(gen_blind_rotate_kernel<11>, T, num_cus); break;
      case 12: r = gen_resident(gen_blind_rotate_kernel<12>, T, num_cus); break;
      default: r = gen_resident(gen_blind_rotate_kernel<13>, T, num_cus); break;
    }
    if (dev <= 15) slot.store(r, std::memory_order_relaxed);
  }
  return work < r ? work : r;
}
long gen_resident_ciphertexts(int logn, int num_cus) { return gen_grid(logn, 1L << 40, num_cus); }

hipError_t launch_gen_blind_rotate(int logn, const GenArgs& a, int num_cus, hipStream_t st) {
  if (a.B <= 0) return hipSuccess;
  const dim3 grid((unsigned)gen_grid(logn, a.B, num_cus)), block((1u << logn) / 16);
  switch (logn) {
    case 10: hipLaunchKernelGGL((gen_blind_rotate_kernel<10>), grid, block, 0, st, a); break;
    case 11: hipLaunchKernelGGL((gen_blind_rotate_kernel<11>), grid, block, 0, st, a); break;
    case 12: hipLaunchKernelGGL((gen_blind_rotate_kernel<12>), grid, block, 0, st, a); break;
    case 13: hipLaunchKernelGGL((gen_blind_rotate_kernel<13>), grid, block, 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gen_bk_transform(int logn, const int32_t* bk, double* bk_x, const double* tw, long n_polys, int num_cus, hipStream_t st) {
  if (n_polys <= 0) return hipSuccess;
  const dim3 grid((unsigned)std::min<long>(n_polys, 8L * num_cus)), block((1u << logn) / 16);
  switch (logn) {
    case 10: hipLaunchKernelGGL((gen_bk_transform_kernel<10>), grid, block, 0, st, bk, bk_x, tw, n_polys); break;
    case 11: hipLaunchKernelGGL((gen_bk_transform_kernel<11>), grid, block, 0, st, bk, bk_x, tw, n_polys); break;
    case 12: hipLaunchKernelGGL((gen_bk_transform_kernel<12>), grid, block, 0, st, bk, bk_x, tw, n_polys); break;
    case 13: hipLaunchKernelGGL((gen_bk_transform_kernel<13>), grid, block, 0, st, bk, bk_x, tw, n_polys); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_gen_polymul(int logn, const int32_t* a_small, const int32_t* b_torus, int32_t* out, double* scratch, const double* tw,
                              long count, unsigned long long* dev_flag, int num_cus, hipStream_t st) {
  if (count <= 0) return hipSuccess;
  const dim3 grid((unsigned)gen_grid(logn, count, num_cus)), block((1u << logn) / 16);
  switch (logn) {
    case 10: hipLaunchKernelGGL((gen_polymul_kernel<10>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, count, dev_flag); break;
    case 11: hipLaunchKernelGGL((gen_polymul_kernel<11>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, count, dev_flag); break;
    case 12: hipLaunchKernelGGL((gen_polymul_kernel<12>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, count, dev_flag); break;
    case 13: hipLaunchKernelGGL((gen_polymul_kernel<13>), grid, block, 0, st, a_small, b_torus, out, scratch, tw, count, dev_flag); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace rs
