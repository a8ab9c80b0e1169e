// rs_diag.h -- everything DIAGNOSTIC about the blind-rotation kernels, behind ONE guard: -DRS_DIAG=<bits>.
// The product build never defines RS_DIAG: every macro below is then empty and diag::kNoKeyProbe is false, so the kernels
// carry no trace of it (tools/codeobj_digest.py: the code objects are those of the tree before this file existed).
//
//   RS_DIAG bit   what the diagnostic build does
//   1             phase stamps in blind_rotate_wg_kernel  (tools/stamp_profile.py)
//   2             phase stamps in blind_rotate_wgs_kernel (the array then lives in part 2 of rs_bootstrap.hip, RS_BS_PART)
//   4             phase stamps in blind_rotate_duo_kernel (tools/stamp_coop8.py duo)
//   8             phase stamps in blind_rotate_coop8_kernel (tools/stamp_coop8.py)
//   256           phase stamps in blind_rotate_coop8_listed_kernel (tools/stamp_coop8.py; the array then lives in part 4)
//   16            NO-KEY TIMING PROBE, RESULTS ARE WRONG: every CMUX step reads the key rows of step 0, which stay in the L2s --
//                 bounds what key streaming can cost a kernel (split lock-step, cooperative, coop8 and general-ring kernels);
//                 rs_api.cpp then also switches the enforced split certificate off (the sums are garbage by construction)
//   32            EXCHANGE-VOLUME TIMING PROBE, RESULTS ARE WRONG: every planar exchange moves only its re plane (half the
//                 plane stores and loads of a transform), and each transform pays 16 v_permlane32_swap on its data registers
//                 instead -- the instruction mix of a transform built from two 4-stage register passes on 32-lane halves (ONE
//                 LDS exchange per transform) plus one cross-half swap stage. Prices that formulation before it is written.
//   64            with 32: the same probe without the swaps (what the exchange volume alone is worth)
//   128           HALF-KEY TIMING PROBE of the general-ring kernels, RESULTS ARE WRONG: only the first four of a row's eight key
//                 positions are loaded (the other four multiply whatever the position registers hold) -- what halving the bytes on the
//                 L2 -> CU path is worth, i.e. the most a form that shares every row between two ciphertexts of a CU could gain
//
// Phase stamps (cdna_hip_programming.md section 7, in-kernel stamps): s_memtime behind s_waitcnt lgkmcnt(0) at up to eight
// phase boundaries, summed per wave into g_rs_stamps and read back by rs_debug_read_stamps (not part of include/redsec_hip.h).
// The stamps drain LDS reads: read the SHARES of a stamped run, never quote its run time.
//
// Every experiment switch of rounds 1-4 (RS_T_*, RS_WG_*, RS_GEN_* ...) that was measured and not adopted is gone from the
// sources together with its code path; the verdicts are in MEASUREMENTS.md. tests/test_abi.py holds csrc/ to that.
#pragma once

namespace rs {
namespace diag {
#ifdef RS_DIAG
constexpr int kBits = RS_DIAG;
#else
constexpr int kBits = 0;
#endif
constexpr bool kNoKeyProbe = (kBits & 16) != 0;
constexpr bool kHalfExchangeProbe = (kBits & 32) != 0;
constexpr bool kHalfExchangeSwaps = (kBits & 64) == 0;
constexpr bool kHalfKeyProbe = (kBits & 128) != 0;
constexpr bool kWrongOnPurpose = kNoKeyProbe || kHalfExchangeProbe || kHalfKeyProbe;   // timing probes: rs_api.cpp then gates and enforces nothing
// key rows of CMUX step i: step 0's under the probe
#if defined(__HIPCC__)
__host__ __device__
#endif
constexpr int key_step(int i) { return kNoKeyProbe ? 0 : i; }
constexpr int kStampPhases = 8;
}  // namespace diag
}  // namespace rs

#ifdef RS_DIAG
#define RS_STAMPS_ON(bit) (((RS_DIAG) & (bit)) != 0)
#if defined(__HIPCC__)
// the phase sums, [workgroup < 256][wave][phase]; defined in the object of rs_bootstrap.hip that launches the stamped kernel
namespace rs { extern __device__ unsigned long long g_rs_stamps[256 * 8 * diag::kStampPhases]; }
#endif
#define RS_STAMP_DECL_ unsigned long long st_sum_[rs::diag::kStampPhases] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_last_; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last_) :: "memory")
#define RS_STAMP_(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
  st_sum_[k] += t_ - st_last_; st_last_ = t_; } while (0)
#define RS_STAMP_FLUSH_(wave) do { if (lane == 0 && blockIdx.x < 256) { for (int k_ = 0; k_ < rs::diag::kStampPhases; ++k_) \
  rs::g_rs_stamps[((size_t)blockIdx.x * 8 + (wave)) * rs::diag::kStampPhases + k_] = st_sum_[k_]; } } while (0)
#else
#define RS_STAMPS_ON(bit) 0
#endif

// one macro triple per stamped kernel; empty unless its RS_DIAG bit is set
#if RS_STAMPS_ON(1)
#define RS_STAMP_DECL RS_STAMP_DECL_
#define RS_STAMP(k) RS_STAMP_(k)
#define RS_STAMP_FLUSH(wave) RS_STAMP_FLUSH_(wave)
#else
#define RS_STAMP_DECL ((void)0)
#define RS_STAMP(k) ((void)0)
#define RS_STAMP_FLUSH(wave) ((void)0)
#endif
#if RS_STAMPS_ON(2)
#define RS_WGS_STAMP_DECL RS_STAMP_DECL_
#define RS_WGS_STAMP(k) RS_STAMP_(k)
#define RS_WGS_STAMP_FLUSH(wave) RS_STAMP_FLUSH_(wave)
#else
#define RS_WGS_STAMP_DECL ((void)0)
#define RS_WGS_STAMP(k) ((void)0)
#define RS_WGS_STAMP_FLUSH(wave) ((void)0)
#endif
#if RS_STAMPS_ON(4)
#define RS_DUO_STAMP_DECL RS_STAMP_DECL_
#define RS_DUO_STAMP(k) RS_STAMP_(k)
#define RS_DUO_STAMP_FLUSH(wave) RS_STAMP_FLUSH_(wave)
#else
#define RS_DUO_STAMP_DECL ((void)0)
#define RS_DUO_STAMP(k) ((void)0)
#define RS_DUO_STAMP_FLUSH(wave) ((void)0)
#endif
#if RS_STAMPS_ON(8)
#define RS_C8_STAMP_DECL RS_STAMP_DECL_
#define RS_C8_STAMP(k) RS_STAMP_(k)
#define RS_C8_STAMP_FLUSH(wave) RS_STAMP_FLUSH_(wave)
#else
#define RS_C8_STAMP_DECL ((void)0)
#define RS_C8_STAMP(k) ((void)0)
#define RS_C8_STAMP_FLUSH(wave) ((void)0)
#endif
#if RS_STAMPS_ON(256)
#define RS_C8L_STAMP_DECL RS_STAMP_DECL_
#define RS_C8L_STAMP(k) RS_STAMP_(k)
#define RS_C8L_STAMP_FLUSH(wave) RS_STAMP_FLUSH_(wave)
#else
#define RS_C8L_STAMP_DECL ((void)0)
#define RS_C8L_STAMP(k) ((void)0)
#define RS_C8L_STAMP_FLUSH(wave) ((void)0)
#endif
// which object of rs_bootstrap.hip (RS_BS_PART) owns the stamp array: the part that holds the stamped kernel (one bit per build)
#if RS_STAMPS_ON(2)
#define RS_DIAG_STAMP_PART 2
#elif RS_STAMPS_ON(256)
#define RS_DIAG_STAMP_PART 4
#elif RS_STAMPS_ON(1 | 4 | 8)
#define RS_DIAG_STAMP_PART 1
#else
#define RS_DIAG_STAMP_PART 0
#endif
