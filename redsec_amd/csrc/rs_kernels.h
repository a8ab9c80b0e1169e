// rs_kernels.h -- kernel argument blocks and launchers (internal; the public boundary is
// include/redsec_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rs_ntt.h"

namespace rs {

struct BlindRotateArgs {
  const int32_t* in0;   // [B][W]
  const int32_t* in1;   // [B][W] or nullptr
  int32_t c0, c1;       // x = c0*in0 + c1*in1 (word-wise, wrapping)
  int32_t bconst;       // added to the b word
  int32_t mu;           // test-vector value
  const double* bk_x;   // transform-domain key of the active mode: [n][2l][2][8][64][2]
  const double* tw;     // twiddle tables of the active mode
  Field f;
  int32_t n;
  int32_t W;            // n + 1
  long B;
  int32_t* u_out;       // [B][N+1] extracted samples
  unsigned int* counter;  // persistent-wave work counter (device), or nullptr
  unsigned long long* dev_flag;  // FFT mode: this CALL's certificate slot (max rounding distance, double bits), or nullptr
  // Programmable form (tfhe_blindRotateAndExtract_FFT's test polynomial): ciphertext b starts from
  // lut[(b % lut_count)][N] instead of the constant mu. nullptr = constant test vector.
  const int32_t* lut = nullptr;
  int32_t lut_count = 0;
  int32_t lut_first = 0;   // table of ciphertext 0 (a caller's shard offset modulo lut_count)
  // Conditional exact recomputation (exact-NTT kernels launched behind an FFT call): the grid reads the
  // FFT call's certificate slot and returns at once unless it reached `gate_limit_bits`; either way it folds
  // the slot into the stream's running maximum and counts the recomputed calls.
  const unsigned long long* gate_flag = nullptr;
  unsigned long long gate_limit_bits = 0;
  unsigned long long* running_flag = nullptr;
  unsigned long long* fallback_count = nullptr;
  // XCD cohorts (blind_rotate_wgs_kernel): progress[xcd * kCohortSlots + slot] = CMUX steps workgroup (xcd, slot) has done,
  // INT_MAX-like for a workgroup that is absent or finished (the launcher fills the table with 0x7f bytes). A workgroup that
  // is more than `cohort_lag` steps ahead of the slowest workgroup on its XCD waits (bounded) every `cohort_every` steps, so
  // the workgroups of an XCD stay within what their L2 holds of the key. nullptr: free-running.
  int* progress = nullptr;
  int32_t cohort_every = 0, cohort_lag = 0;
};
constexpr int kCohortSlots = 64;   // workgroups per XCD the table has room for (256 CUs / 8 XCDs = 32)

// Launch policy switches, read from the environment ONCE at rs_create (A/B experiments only).
struct LaunchOpts {
  bool no_coop = false, no_wg = false, no_duo = false, no_persist = false, no_conv_tiled = false, no_wg4 = false, no_tail = false, no_coop8 = false, no_coop8_listed = false, ks_atomics = false, force_host_staged = false, no_cohort = false;
};
// What a blind-rotate launch actually ran: kernel form and how many ciphertexts share one sweep of the key
// from L2/HBM (R of SURVEY.md section 8d).
enum { kFormPerWave = 0, kFormWorkgroup = 1, kFormDuo = 2, kFormCoop2 = 3, kFormCoop4 = 4, kFormGeneral = 5, kFormSplitWorkgroup = 6, kFormSplitCoop = 7, kFormSplitDuo = 8, kFormCoop8 = 9, kFormCoop8Listed = 10 };
struct LaunchInfo { int form = -1; int waves_per_block = 0; long resident = 0; };

struct KeyswitchArgs {
  const int32_t* u0;    // [B][N+1]
  const int32_t* u1;    // optional second addend (bootsMUX)
  int32_t bconst;       // added to the b word of u
  const int32_t* ksk;   // [N][t][base][W]
  int32_t W, t, basebit;
  long B;
  int32_t* out;         // [B][W]
  int32_t N = kN;       // ring degree of the extracted samples (the tiled kernels are N = 1024 only)
  // Small batches: the N input coefficients are cut into slices (blockIdx.z) so that enough workgroups exist; with a scratch
  // of keyswitch_scratch_words() words every slice stores its partial sums there ([slice][W][B], ciphertext fastest: coalesced)
  // and keyswitch_reduce_kernel sums them into `out`. Without one (or one too small) the slices meet by integer atomics in a
  // zeroed output (the round-1 form: 17 G uncoalesced atomics/s were what bounded it, profiles/r03/y_ab_*).
  uint32_t* scratch = nullptr;
  size_t scratch_words = 0;
};
size_t keyswitch_scratch_words(const KeyswitchArgs& a);   // 0 when the launch will not be sliced

// General ring path (rs_general.h / rs_general.hip): any N = 2^logn in [1024, 8192], any gadget, split key.
struct GenArgs {
  const int32_t* in0;   // [B][W]
  const int32_t* in1;   // [B][W] or nullptr
  int32_t c0, c1, bconst, mu;
  const double* bk_x;   // [n][2l][2 halves][2 columns][8][N/16][2]
  const double* tw;     // gen_make_twiddles(logn)
  int32_t n, W, l, bgbit;
  long B;
  int32_t* u_out;       // [B][N+1]
  const int32_t* lut = nullptr;   // programmable form, as in BlindRotateArgs
  int32_t lut_count = 0, lut_first = 0;
  unsigned long long* dev_flag = nullptr;   // optional: largest rounding distance (diagnostic; exactness does not depend on it)
};

struct ConvShape { int32_t H, Wd, Cin, Cout, fh, fw, stride_h, stride_w, off_h, off_w, Ho, Wo; };
struct PoolShape { int32_t H, Wd, C, win_h, win_w, stride_h, stride_w, off_h, off_w, Ho, Wo; };

// cfg: 0 = CfgDefault128 (l=3, Bgbit=7), 1 = CfgRedsecV2 (l=10, Bgbit=3); mode: 0 = exact NTT, 1 = FFT
hipError_t launch_blind_rotate(int cfg, int mode, const BlindRotateArgs& a, int waves_per_block, int num_cus, const LaunchOpts& opts,
                               hipStream_t st, LaunchInfo* info = nullptr);
hipError_t launch_blind_rotate_split_wg(int cfg, const BlindRotateArgs& a, int num_cus, const LaunchOpts& opts, hipStream_t st, LaunchInfo* info);
hipError_t launch_bk_transform(int cfg, int mode, const int32_t* bk, double* bk_x, const double* tw, Field f, double scale,
                               long n_polys, hipStream_t st);
hipError_t launch_keyswitch(const KeyswitchArgs& a, hipStream_t st);
hipError_t launch_gen_blind_rotate(int logn, const GenArgs& a, int num_cus, hipStream_t st);
long gen_resident_ciphertexts(int logn, int num_cus);   // workgroups (= ciphertexts) the general kernel keeps resident
hipError_t launch_gen_bk_transform(int logn, const int32_t* bk, double* bk_x, const double* tw, long n_polys, int num_cus, hipStream_t st);
hipError_t launch_gen_polymul(int logn, const int32_t* a_small, const int32_t* b_torus, int32_t* out, double* scratch, const double* tw,
                              long count, unsigned long long* dev_flag, int num_cus, hipStream_t st);
hipError_t launch_polymul(int cfg, int mode, const int32_t* a_small, const int32_t* b_torus, int32_t* out, double* scratch,
                          const double* tw, Field f, double scale, long count, unsigned long long* dev_flag, hipStream_t st);
hipError_t launch_lincomb(int32_t* out, const int32_t* x, int32_t cx, const int32_t* y, int32_t cy, int32_t bconst, int W, long B,
                          hipStream_t st);
hipError_t launch_synthetic_words(int32_t* out, uint64_t seed, size_t total, hipStream_t st);
hipError_t launch_gather_rows(int32_t* out, const int32_t* in, const int32_t* idx, int W, long B, hipStream_t st);
hipError_t launch_linear_fc(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, int K, int M, int W,
                            int32_t zero_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st);
size_t conv_tiled_scratch_words(const ConvShape& s);
hipError_t launch_conv_ternary_tiled(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, const ConvShape& s, int W,
                                     const int32_t* bias_b, int bias_depth, uint32_t* scratch, hipStream_t st);
hipError_t launch_conv_ternary(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero, const ConvShape& s, int W,
                               int32_t zero_tap_b, int32_t pad_tap_b, const int32_t* bias_b, int bias_depth, hipStream_t st);
hipError_t launch_fp64_rate(double* out, int num_cus, int iters, double* lane_ops, hipStream_t st);
hipError_t launch_sumpool(int32_t* out, const int32_t* in, const PoolShape& s, int W, const int32_t* bias_b, int bias_depth,
                          hipStream_t st);

}  // namespace rs
