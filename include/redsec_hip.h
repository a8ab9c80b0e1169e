/*
 * redsec_hip.h -- C ABI of the MI355X gate-bootstrapping backend (libredsec_hip.so).
 *
 * This is the drop-in boundary for REDsec's encrypted hot path. Each entry point replaces a call
 * that the reference makes into the external TFHE library (paths relative to /root/reference):
 *
 *   rs_bootstrap / rs_bootstrap_dev     tfhe_bootstrap_FFT             lib/BinOps_enc.cpp:185,191
 *                                       (Quantize::execute loops       lib/BinFunc.cpp:1056-1071,
 *                                                                       lib/IntFunc.cpp:871-887)
 *   rs_gate / rs_gate_dev               bootsAND/OR/XOR/...            lib/BinOps_enc.cpp:49-52,104-113,
 *                                                                       153-166,205; lib/IntOps_enc.cpp:63
 *   rs_mux / rs_mux_dev                 bootsMUX                       lib/IntFunc.cpp:962
 *   rs_lincomb_dev, rs_linear_*         lweAddTo/lweSubTo/lweAddMulTo/ lib/BinFunc.cpp:195-320,
 *                                       lweNoiselessTrivial loops      lib/IntFunc.cpp:207-308,643-700
 *   rs_create / rs_load_keys            new_tfheGateBootstrappingCloudKeySet_fromFile + the
 *                                       bkFFT precomputation           nets/mnist/sign1024x1/net.cpp:53-55
 *
 * Plain pointers and sizes only. "_dev" entry points take DEVICE pointers (hipMalloc / torch CUDA
 * tensors) and enqueue on the given hipStream_t (passed as void*; NULL = default stream) without
 * synchronising; the others take HOST pointers and are synchronous.
 *
 * Ciphertext layout: an LWE sample of dimension n is W = n+1 consecutive int32 words
 * (a[0..n-1], b); a batch is int32[B][W], row-major, contiguous. Torus32 arithmetic wraps mod 2^32.
 *
 * Errors: every function returns 0 on success or a negative rs_status; rs_last_error() gives the
 * message for the calling thread. There is NO CPU fallback: without a HIP device every compute
 * entry point fails with RS_ERR_NO_DEVICE.
 */
#ifndef REDSEC_HIP_H
#define REDSEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rs_status {
  RS_OK = 0,
  RS_ERR_INVALID = -1,     /* bad argument / unsupported parameter set */
  RS_ERR_NO_DEVICE = -2,   /* no HIP device or device is not gfx950-compatible */
  RS_ERR_HIP = -3,         /* a HIP runtime call failed */
  RS_ERR_STATE = -4,       /* keys not loaded, etc. */
  RS_ERR_INEXACT = -5      /* RS_MODE_FFT_SPLIT: the enforced rounding certificate failed (see rs_split_bound); sticky until
                              rs_certify(..., reset = 1). The check of call k on a stream is looked at by call k + 1 on that
                              stream, rs_sync, rs_certify, rs_release_stream, rs_destroy and every host-pointer call -- NOT by
                              the *_dev call itself (it is asynchronous): a device-pointer caller must pass one of those
                              before trusting the LAST call's output (rs_destroy returns the error too) */
} rs_status;

/* Mirrors TFheGateBootstrappingParameterSet (ks_t, ks_basebit, in_out_params->n, tgsw_params->l,
 * Bgbit, tlwe_params->N, k) as constructed at client/gen_secure_keyset.cpp:82-90. */
typedef struct rs_params {
  int32_t n;          /* LWE dimension */
  int32_t N;          /* ring degree: 1024, 2048, 4096 or 8192 */
  int32_t k;          /* must be 1 */
  int32_t bk_l;       /* gadget length l and base 2^Bgbit, l * Bgbit <= 32. N = 1024 with 3/7 or 10/3 (the shipped sets) */
  int32_t bk_Bgbit;   /* has all three modes below; every other set runs in RS_MODE_FFT_SPLIT only */
  int32_t ks_t;
  int32_t ks_basebit;
} rs_params;

typedef struct rs_ctx rs_ctx;

typedef enum rs_gate_op {
  RS_NAND = 0, RS_OR = 1, RS_AND = 2, RS_NOR = 3, RS_XOR = 4, RS_XNOR = 5,
  RS_ANDNY = 6, RS_ANDYN = 7, RS_ORNY = 8, RS_ORYN = 9
} rs_gate_op;

const char* rs_last_error(void);
const char* rs_version(void);

/* Parameter sets shipped with the reference / TFHE. */
int rs_params_default128(rs_params* p);        /* TFHE default 128-bit set (NAND microbench) */
int rs_params_redsec_small_v2(rs_params* p);   /* client/gen_secure_keyset.cpp:70-91 (the set the client ships with) */
int rs_params_redsec_small(rs_params* p);      /* client/gen_secure_keyset.cpp:47-68:  n=500  N=1024 l=3 Bgbit=10 */
int rs_params_redsec_medium(rs_params* p);     /* client/gen_secure_keyset.cpp:28-45:  n=3072 N=4096 l=3 Bgbit=10 */
int rs_params_redsec_large(rs_params* p);      /* client/gen_secure_keyset.cpp:9-26:   n=6144 N=8192 l=3 Bgbit=10 */

/* Context bound to one HIP device (device index as in hipSetDevice). */
int rs_create(rs_ctx** out, const rs_params* p, int device);
int rs_destroy(rs_ctx* ctx);

/* Upload the evaluation key (HOST pointers):
 *   bk  int32[n][(k+1)*l][k+1][N]   TGSW rows, row p = c*l + j (bk->bk[i].all_sample[p].a[col])
 *   ksk int32[k*N][t][1<<basebit][n+1]   bk->ks->ks[i][j][v] as (a[0..n-1], b)
 * Transforms bk to the transform domain on the device (the bkFFT analogue). */
int rs_load_keys(rs_ctx* ctx, const int32_t* bk, const int32_t* ksk);
/* The same with a SYNTHETIC key generated on the device: bk word k = high half of splitmix64(seed + k) (csrc/rs_ntt.h,
 * synthetic_key_word; redsec_amd/client.py restates it in numpy), ksk word k likewise from seed ^ 0x6b73. Not an encryption of
 * anything: for benchmarks and parity tests of the large rings, whose real keys are gigabytes on the host (redsec_params_large:
 * 2.4 GB of bk + 7.2 GB of ksk) -- every kernel does exactly the work it does on a real key, and the oracle is fed the same words. */
int rs_load_synthetic_keys(rs_ctx* ctx, uint64_t seed);

/* Arithmetic of the external product (both keys are resident after rs_load_keys; switching is free):
 *   RS_MODE_FFT        folded 512-point complex FP64 FFT -- the arithmetic class of TFHE's own
 *                      tGswFFTExternMulToTLwe -- rounded to the nearest integer. The true product is an
 *                      integer and the FFT error is two orders of magnitude below 1/2, so rounding returns
 *                      exactly the integer result. Every bootstrapped call records the largest distance to an
 *                      integer it rounded (its "certificate") and is followed, on the same stream, by the
 *                      exact-NTT kernels GATED on that certificate: they return at once while it is below the
 *                      limit and otherwise overwrite the call's result with the exact one before anything
 *                      downstream reads it. No host round trip; holds for the *_dev calls and the host calls alike.
 *   RS_MODE_EXACT_NTT  exact negacyclic NTT over a 51-bit prime carried in FP64: exact by construction,
 *                      2.3x the FP64 operations.
 *   RS_MODE_FFT_SPLIT  the same FP64 FFT with the key split into two signed 16-bit halves (twice the pointwise
 *                      products and inverse transforms). Every half product stays below 2^40, where the FFT's
 *                      WORST-CASE error is below 1/2 for every parameter set the reference defines (a-priori bound
 *                      derived in csrc/rs_general.h, rs_split_bound: 3e-4 ... 0.011 for N = 1024, 0.10 for N = 4096,
 *                      0.30 for N = 8192): rounding is exact for EVERY input. The general kernels (N >= 2048, or any
 *                      gadget outside the shipped ones) also ENFORCE a rounding certificate: a call that rounds a value
 *                      1/4 or more away from an integer makes the next call on its stream, rs_sync, rs_certify and the
 *                      host-pointer calls fail with RS_ERR_INEXACT (never observed: measured distances are 4e-6).
 *                      General kernels: any N in {1024 ... 8192}, any gadget; the only mode of the sets outside the
 *                      specialised N = 1024 kernels.
 * Results are identical word for word in all modes (= the CPU oracle).
 * Default RS_MODE_FFT where available (environment REDSEC_MODE=exact | split selects another at context creation;
 * the environment is read ONCE, in rs_create). rs_set_mode must not race with launches of the same context. */
enum { RS_MODE_EXACT_NTT = 0, RS_MODE_FFT = 1, RS_MODE_FFT_SPLIT = 2 };
int rs_set_mode(rs_ctx* ctx, int mode);
int rs_get_mode(rs_ctx* ctx, int* mode);
/* The a-priori bound on |computed - true coefficient| of a split-key product for this context's parameters (derived in
 * csrc/rs_general.h from the rounding model of the FP64 butterflies actually used; nothing quoted). The mode is offered
 * when it is below 1/2, i.e. when rounding to the nearest integer is exact for every input. */
int rs_split_bound(rs_ctx* ctx, double* bound);
/* A call whose certificate reaches this limit is recomputed exactly on the device (default 0.25: an error
 * of +-1 needs a distance > 0.5). limit = 0 forces the recomputation of every call (tests). */
#define RS_CERTIFICATE_LIMIT 0.25
int rs_set_certificate_limit(rs_ctx* ctx, double limit);
/* Synchronises `stream` and reports what its calls did since the last reset: the largest rounding distance
 * and how many calls were recomputed exactly (expected: 0). Either pointer may be NULL. */
int rs_certify(rs_ctx* ctx, void* stream, double* max_distance, int64_t* recomputed_calls, int reset);
/* The same over ALL streams of the context (device-wide synchronisation). */
int rs_rounding_certificate(rs_ctx* ctx, double* max_distance, int reset);
int rs_fft_fallbacks(rs_ctx* ctx, int64_t* count);

/* Streams. A context keeps one private slice of mutable state per stream it has seen (extracted-sample
 * workspace, work counter, certificate slots, convolution scratch, timing events), so *_dev calls on
 * DIFFERENT streams of one context may be issued concurrently, also from different host threads; calls on
 * one stream are ordered by the stream. The synchronous host-pointer calls are serialised per context.
 * Workspace is grown on demand (a device-wide wait); rs_reserve pre-sizes the default stream's,
 * rs_reserve_stream a given stream's, for batches of up to max_batch ciphertexts. */
int rs_reserve(rs_ctx* ctx, size_t max_batch);
int rs_reserve_stream(rs_ctx* ctx, size_t max_batch, void* stream);

/* out[b] = tfhe_bootstrap_FFT(mu, in[b]) for b < B. */
int rs_bootstrap_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, int32_t mu, size_t B, void* stream);
int rs_bootstrap(rs_ctx* ctx, int32_t* out, const int32_t* in, int32_t mu, size_t B);

/* Programmable bootstrap: tfhe_blindRotateAndExtract_FFT with the test polynomial lut[(lut_first + b) % lut_count]
 * (DEVICE int32[lut_count][N]; lut_first lets a caller that shards a batch keep the batch-wide assignment)
 * followed by lweKeySwitch. With pbar = the phase of in[b] mod-switched to
 * [0, 2N): out[b] encrypts lut[pbar] for pbar < N and -lut[pbar - N] beyond. Serves the corrected
 * Quantize::relu_shift (lib/IntFunc.cpp:934-973, lib/BinFunc.cpp:1120-1162): one bootstrap per neuron
 * evaluates clamp((slope x + bias) >> slope_bits, 0, 2^shift_bits - 1), see DESIGN.md "ReLU semantics". */
int rs_bootstrap_lut_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, const int32_t* lut, size_t lut_count, size_t lut_first,
                         size_t B, void* stream);

/* out[b] = boots<OP>(a[b], b[b]) (mu = 1/8 encoding). */
int rs_gate_dev(rs_ctx* ctx, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, size_t B, void* stream);
int rs_gate(rs_ctx* ctx, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, size_t B);

/* Same gate, but the output encodes the result as +-mu instead of +-1/8 (the test-vector value is a
 * free parameter of the bootstrap). Used by the max-pool OR chain, whose last OR must hand +-1/4096
 * to the next linear stage (lib/BinFunc.cpp:880-925 feeds +-1/4096 sign outputs to bootsOR, which
 * assumes +-1/8: see DESIGN.md "max-pool semantics"). */
int rs_gate_mu_dev(rs_ctx* ctx, rs_gate_op op, int32_t* out, const int32_t* a, const int32_t* b, int32_t mu, size_t B, void* stream);

/* out[i] = bootsMUX(a[i], b[i], c[i]) = a ? b : c. */
int rs_mux_dev(rs_ctx* ctx, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* c, size_t B, void* stream);
int rs_mux(rs_ctx* ctx, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* c, size_t B);

/* Pieces of the bootstrap, exposed for parity tests and for callers that fuse differently:
 *   blind rotate + sample extract (tfhe_bootstrap_woKS_FFT): in int32[B][n+1] -> u int32[B][k*N+1]
 *   keyswitch (lweKeySwitch):                                 u int32[B][k*N+1] -> out int32[B][n+1] */
int rs_bootstrap_wo_ks_dev(rs_ctx* ctx, int32_t* u, const int32_t* in, int32_t mu, size_t B, void* stream);
int rs_keyswitch_dev(rs_ctx* ctx, int32_t* out, const int32_t* u, size_t B, void* stream);

/* Debug/parity tap: negacyclic product of a small-coefficient polynomial with a torus polynomial
 * through exactly the device transform path used by the external product (HOST pointers). */
int rs_debug_polymul(rs_ctx* ctx, int32_t* out, const int32_t* a_small, const int32_t* b_torus, size_t count);
/* Debug tap of the XCD cohort protocol of the lock-step kernels (no reference counterpart: the reference has no batch kernel):
 * copies the progress table of `stream`'s last cohort launch to out[8 * 64] (HOST) after synchronising the stream. Entry
 * [xcd * 64 + slot] of workgroup (blockIdx & 7, blockIdx >> 3) reads 0x40000000 + the CMUX steps that workgroup walked once it
 * has left; entries no workgroup owned read 0x7f7f7f7f. RS_ERR_STATE before the first such launch. */
int rs_debug_cohort_table(rs_ctx* ctx, void* stream, int32_t* out);
/* Box calibration (no reference counterpart): the FP64 fused-multiply-add lane-operations per second the device sustains right
 * now at the occupancy of the blind-rotation kernels (8 waves per CU, 16 independent chains per lane; best of three 10-ms
 * launches on the default stream). bench.py reports it beside roofline_valu: boxes of one pool differ by several per cent. */
int rs_debug_fp64_rate(rs_ctx* ctx, double* lane_ops_per_s);

/* ---- linear stage on LWE words (no bootstrap), DEVICE pointers ---------------------------------
 * out[m] = bias_b[m % bias_depth] (on the b word, optional) + zero_tap_b * (#zero taps of m)
 *          + sum_k s(k,m) * in[k]
 * with s = +1 where sign[k*M+m] == 1, -1 where 0, and the tap replaced by the constant where
 * zero[k*M+m] == 1 (zero may be NULL). Fully-connected form of Convolution::execute
 * (lib/BinFunc.cpp:217-320: zero_tap_b = 0; lib/IntFunc.cpp:227-308: zero_tap_b = -1/4096). */
int rs_linear_fc_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero,
                     int32_t K, int32_t M, int32_t zero_tap_b, const int32_t* bias_b, int32_t bias_depth,
                     void* stream);

/* General 2-D convolution over ciphertext feature maps (NHWC), same semantics as above per tap;
 * out-of-bounds taps under same-padding contribute pad_tap_b on the b word.
 * in  int32[H][Wd][Cin][W], out int32[Ho][Wo][Cout][W], sign/zero uint8[fh][fw][Cin][Cout]
 * (get_filter_i, lib/BinFunc.cpp:388). */
typedef struct rs_conv_shape {
  int32_t H, Wd, Cin, Cout, fh, fw, stride_h, stride_w, off_h, off_w, Ho, Wo;
} rs_conv_shape;
int rs_conv_ternary_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero,
                        const rs_conv_shape* shape, int32_t zero_tap_b, int32_t pad_tap_b,
                        const int32_t* bias_b, int32_t bias_depth, void* stream);

/* Windowed LWE sum (SumPooling::execute, lib/BinFunc.cpp:677-732, lib/IntFunc.cpp:643-700):
 * in int32[H][Wd][C][W] -> out int32[Ho][Wo][C][W]; taps outside the image are skipped. */
typedef struct rs_pool_shape {
  int32_t H, Wd, C, win_h, win_w, stride_h, stride_w, off_h, off_w, Ho, Wo;
} rs_pool_shape;
int rs_sumpool_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, const rs_pool_shape* shape,
                   const int32_t* bias_b, int32_t bias_depth, void* stream);

/* out[i] = ca * a[i] + cb * b[i] word-wise, plus bconst on the b word (b may be NULL).
 * lweAddTo / lweSubTo / lweAddMulTo / lweNoiselessTrivial compositions (lib/BinOps_enc.cpp:37-41,
 * 121-143; lib/IntOps_enc.cpp:35-56). */
int rs_lincomb_dev(rs_ctx* ctx, int32_t* out, const int32_t* a, int32_t ca, const int32_t* b, int32_t cb,
                   int32_t bconst, size_t B, void* stream);

/* out[i] = in[row_index[i]] for i < B (row_index on the device; a negative index gives the trivial
 * zero sample). Gathers the window taps of MaxPooling::execute (lib/BinFunc.cpp:896-921). */
int rs_gather_rows_dev(rs_ctx* ctx, int32_t* out, const int32_t* in, const int32_t* row_index, size_t B, void* stream);

/* Device memory helpers for hosts that do not link HIP themselves (the C++ layer mirror). */
int rs_dev_alloc(rs_ctx* ctx, void** ptr, size_t bytes);
int rs_dev_free(rs_ctx* ctx, void* ptr);
int rs_copy_to_dev(rs_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int rs_copy_to_host(rs_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Device-to-device copy between the devices of two contexts (the same device is fine): the slice exchange of a
 * stage sharded over several GPUs of one process. Synchronous. */
int rs_copy_dev_to_dev(rs_ctx* dst_ctx, void* dst_dev, rs_ctx* src_ctx, const void* src_dev, size_t bytes);
/* The slice exchange of a stage sharded over the n contexts of one process (the reference's shape: one host process driving
 * NUM_GPUS devices, lib/GPU/Layer.cuh:15,22-37, which has no merge step at all). bufs[d] is context d's full replica
 * int32[rows][row_words]; context d has just computed rows shard(d) = the d-th of n balanced contiguous slices (sizes differ
 * by at most one, lower ranks first) into it, on ITS default stream. Afterwards every replica holds every slice.
 * Entirely asynchronous and event-ordered: each context records an event on its default stream at the start of the call
 * (behind the kernels that wrote its slice AND behind whatever it queued earlier that still reads bufs[d]'s previous
 * contents); every destination first waits for its own event, then pulls slice e on a private copy stream as soon as e's
 * event has fired, so slice e moves while e+1 is still being computed; finally every context's default stream waits for
 * all copies, so later launches (and buffer reuse) on any of them are ordered behind the exchange. Nothing blocks the host.
 * Path per device pair: contexts on one device copy device-to-device; different devices use hipMemcpyPeerAsync over xGMI
 * when hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess succeed (asked once per pair), and otherwise the library stages the
 * slice through pinned host memory itself (source D2H once, each such destination H2D) -- slower, same result
 * (RS_FORCE_HOST_STAGED=1 at rs_create forces that path everywhere: how a one-GPU box tests it). The operation list is
 * host logic (csrc/rs_host.h exchange_plan), checked on the CPU. */
int rs_allgather_rows(rs_ctx* const* ctxs, int n_ctx, int32_t* const* bufs, size_t rows, size_t row_words);
int rs_sync(rs_ctx* ctx);
/* Drops the private state a context keeps for `stream` (workspace, certificate slots, events; see "Streams" above) after
 * synchronising it. For callers that create and destroy many streams: a later stream with the same handle value would
 * otherwise inherit the old one's workspace size and running certificate. The default stream's state cannot be released. */
int rs_release_stream(rs_ctx* ctx, void* stream);

/* Time (ms) of the kernels enqueued by the last *_dev / host call, by HIP events on the stream the
 * kernels ran on; -1 if not available. Index: 0 blind-rotate, 1 keyswitch. */
int rs_set_timing(rs_ctx* ctx, int enable);
int rs_last_kernel_ms(rs_ctx* ctx, float* blind_rotate_ms, float* keyswitch_ms);               /* default stream */
int rs_last_kernel_ms_stream(rs_ctx* ctx, void* stream, float* blind_rotate_ms, float* keyswitch_ms);

/* Facts used by bench.py's roofline accounting. rs_last_launch: what the last blind rotation on `stream`
 * actually ran -- form 0 per-wave, 1 lock-step workgroups, 2 duo, 3 / 4 cooperative (2 / 4 waves per
 * ciphertext), 5 general (one workgroup of N/16 threads per ciphertext), 6 lock-step workgroups on the split key (8 or 4 waves),
 * 7 cooperative on the split key (2 / 4 waves per ciphertext), 8 duo on the split key, 9 cooperative with 8 waves per ciphertext,
 * 10 the same with the listed step (gadgets with l < 4) -- its waves per workgroup, and `resident` = ciphertexts sharing one sweep of the key (R of the
 * algorithmic-bytes formula). rs_info's waves_per_block is that of the default stream's last launch. */
int rs_last_launch(rs_ctx* ctx, void* stream, int32_t* form, int32_t* waves_per_block, int64_t* resident);
int rs_info(rs_ctx* ctx, int64_t* bk_device_bytes, int64_t* ksk_device_bytes, int32_t* waves_per_block,
            int32_t* num_cus);

#ifdef __cplusplus
}
#endif
#endif
