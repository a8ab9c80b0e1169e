/*
 * redsec_oracle.h -- CPU oracle for the REDsec gate-bootstrapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under redsec_amd/ (the product) may include, link or call
 * this. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * and the timed CPU baseline -- never as the thing measured or shipped.
 *
 * PARITY UNPINNED at the TFHE boundary: the arithmetic REDsec calls lives in the third-party
 * library TFHE v1.1 (libtfhe-spqlios-fma; pinned only by prose in /root/reference/README.md:9-10,
 * linked at /root/reference/lib/Makefile:3). It is not vendored in the reference and not installed
 * in this image, and the reference holds no golden ciphertexts, keys or known-answer tests for it
 * (SURVEY.md section 8c). This file therefore restates TFHE v1.1's *published algorithm*
 * (CGGI gate bootstrapping: Chillotti, Gama, Georgieva, Izabachene, "TFHE: Fast Fully Homomorphic
 * Encryption over the Torus", J. Cryptology 2020; the library's lwe-functions / tgsw-functions /
 * lwe-bootstrapping-functions-fft / boolean-gates sources) in exact integer arithmetic, anchored on
 * REDsec's own call sites (cited per function below). What IS pinned:
 *   - self-validating known-answer tests (tests/test_oracle_*.py): truth tables, sign bootstrap over
 *     the message space, schoolbook-vs-NTT polynomial products, keyswitch phase preservation;
 *   - the network logic around the bootstraps against the reference's own plaintext build
 *     (oracle/_ref, built from /root/reference by oracle/Makefile) and tests/golden/ fixtures.
 * Where TFHE's FFT path rounds (double-precision Lagrange FFT), this oracle is exact: it computes
 * the same function as TFHE's exact (non-FFT) bootstrap, so ciphertext words may differ from a
 * real libtfhe run in the low bits of the noise while decrypting identically.
 *
 * Conventions (TFHE v1.1):
 *   Torus32 = int32_t, all arithmetic wraps mod 2^32.
 *   An LWE sample of dimension n is W = n+1 consecutive int32 words: a[0..n-1], then b.
 *   phase = b - sum a_i s_i.
 */
#ifndef REDSEC_ORACLE_H
#define REDSEC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_params {
  int32_t n;          /* LWE dimension (in_out_params->n)                         */
  int32_t N;          /* ring degree (tlwe_params->N), a power of two (1024 ... 8192)  */
  int32_t k;          /* TRLWE mask polynomials; every REDsec/TFHE set uses 1     */
  int32_t bk_l;       /* gadget length l                                          */
  int32_t bk_Bgbit;   /* log2 of gadget base Bg                                   */
  int32_t ks_t;       /* keyswitch decomposition length t                         */
  int32_t ks_basebit; /* log2 of keyswitch base                                   */
  double lwe_stdev;   /* LweParams alpha_min: noise of KSK rows                   */
  double bk_stdev;    /* TLweParams alpha_min: noise of bootstrapping-key rows    */
} ro_params;

/* TFHE default 128-bit gate-bootstrapping set (new_default_gate_bootstrapping_parameters(128)):
 * n=630 N=1024 k=1 l=3 Bgbit=7 t=8 basebit=2, ks_stdev 2^-15, bk_stdev 2^-25. */
void ro_params_default128(ro_params* p);
/* REDsec shipped set: /root/reference/client/gen_secure_keyset.cpp:70-91 (redsec_params_small_v2). */
void ro_params_redsec_small_v2(ro_params* p);
/* The other sets of client/gen_secure_keyset.cpp: :47-68 (n=500 N=1024 l=3 Bgbit=10 t=18 basebit=1), :28-45 (n=3072
 * N=4096), :9-26 (n=6144 N=8192). */
void ro_params_redsec_small(ro_params* p);
void ro_params_redsec_medium(ro_params* p);
void ro_params_redsec_large(ro_params* p);

/* ---- torus helpers (TFHE lwe-functions / numeric_functions) ---- */
int32_t ro_modswitch_to_torus32(int32_t mu, int32_t Msize);
int32_t ro_modswitch_from_torus32(int32_t phase, int32_t Msize);
int32_t ro_approx_phase(int32_t phase, int32_t Msize);

/* ---- deterministic PRNG (xoshiro256**), test-only; TFHE's own generator cannot be reproduced ---- */
typedef struct ro_rng { uint64_t s[4]; int has_spare; double spare; } ro_rng;
void ro_rng_seed(ro_rng* r, uint64_t seed);
uint32_t ro_rng_u32(ro_rng* r);
double ro_rng_gaussian(ro_rng* r, double sigma);
int32_t ro_gaussian32(ro_rng* r, int32_t message, double sigma);

/* ---- keys ---- */
/* Sizes (in int32 words):
 *   lwe_key  [n]                      s in {0,1}
 *   tlwe_key [k*N]                    s' in {0,1}
 *   bk       [n][(k+1)*l][k+1][N]     row p = c*l + j encrypts s_i * 2^(32-(j+1)Bgbit) on component c
 *   ksk      [k*N][t][base][n+1]      ksk[i][j][v] = LWE_s( v * s'_i * 2^(32-(j+1)basebit) ), v=0 trivial 0
 */
size_t ro_bk_words(const ro_params* p);
size_t ro_ksk_words(const ro_params* p);
void ro_synthetic_key_words(uint64_t seed, uint64_t first, uint64_t count, int32_t* out);   /* test generator, see the .c file */
void ro_keygen(const ro_params* p, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key, int32_t* bk, int32_t* ksk);

/* ---- LWE sample ops (lweSymEncrypt / lwePhase / lweSymDecrypt / lweNoiselessTrivial) ---- */
void ro_lwe_encrypt(int32_t* out, int32_t mu, double alpha, const int32_t* key, int32_t n, ro_rng* rng);
int32_t ro_lwe_phase(const int32_t* sample, const int32_t* key, int32_t n);
int32_t ro_lwe_decrypt(const int32_t* sample, const int32_t* key, int32_t n, int32_t Msize);
void ro_lwe_trivial(int32_t* out, int32_t mu, int32_t n);

/* ---- exact negacyclic products (two independent implementations, cross-checked in tests) ---- */
/* out[j] = sum_i a[i]*b[j-i] (negacyclic, X^N = -1), coefficients mod 2^32. Schoolbook O(N^2). */
void ro_negacyclic_mul_schoolbook(int32_t* out, const int32_t* a_small, const int32_t* b_torus, int32_t N);
/* Same via a 64-bit NTT mod P = 2^64-2^32+1; exact while |sum| < P/2. */
void ro_negacyclic_mul_ntt(int32_t* out, const int32_t* a_small, const int32_t* b_torus, int32_t N);

/* ---- evaluation context: BK pre-transformed to the NTT domain ---- */
typedef struct ro_ctx ro_ctx;
ro_ctx* ro_ctx_create(const ro_params* p, const int32_t* bk, const int32_t* ksk);
void ro_ctx_destroy(ro_ctx* c);
/* use_schoolbook != 0 switches the external product to the O(N^2) definitional path (slow). */
void ro_ctx_set_schoolbook(ro_ctx* c, int use_schoolbook);
/* use_fft != 0 switches the external product to a double-precision folded complex FFT rounded to the
 * nearest integer -- the arithmetic CLASS of TFHE's own CPU library (Lagrange half-complex FFT), and
 * the fast CPU path bench.py times as `cpu_baseline`. Exact after rounding with overwhelming
 * probability (checked against the exact NTT path in tests/test_oracle_kat.py); the PARITY CHECKS
 * always use the exact paths. The first call transforms the bootstrapping key (not thread-safe). */
void ro_ctx_set_fft(ro_ctx* c, int use_fft);

/* tfhe_bootstrap_woKS_FFT semantics, exact: in [n+1] -> out [k*N+1] under the extracted key. */
void ro_bootstrap_wo_ks(const ro_ctx* c, int32_t* out_extracted, int32_t mu, const int32_t* in);
/* lweKeySwitch: in [k*N+1] -> out [n+1]. */
void ro_keyswitch(const ro_ctx* c, int32_t* out, const int32_t* in_extracted);
/* tfhe_bootstrap_FFT = woKS + keyswitch (REDsec call site: BinOps_enc.cpp:182-192). */
void ro_bootstrap(const ro_ctx* c, int32_t* out, int32_t mu, const int32_t* in);

/* Programmable bootstrap (tfhe_blindRotateAndExtract_FFT with an arbitrary test polynomial + lweKeySwitch):
 * out decrypts to testvect[pbar] for the mod-switched phase pbar in [0, N), -testvect[pbar - N] in [N, 2N).
 * Batch: ciphertext b uses luts[b % lut_count] (N words each). */
void ro_bootstrap_lut(const ro_ctx* c, int32_t* out, const int32_t* testvect, const int32_t* in);
void ro_bootstrap_lut_batch(const ro_ctx* c, int32_t* out, const int32_t* luts, size_t lut_count, const int32_t* in, size_t B);

/* Debug tap: run the blind rotate for the first `steps` key indices only and return the TRLWE
 * accumulator (k+1 polynomials of N words). steps<0 means all n. */
void ro_blind_rotate_acc(const ro_ctx* c, int32_t* acc_out, int32_t mu, const int32_t* in, int32_t steps);

/* ---- gates (TFHE boolean-gates.cpp; constants mirrored at /root/reference/lib/GPU/gates.cu:246-286) ---- */
typedef enum ro_gate_op {
  RO_NAND = 0, RO_OR = 1, RO_AND = 2, RO_NOR = 3, RO_XOR = 4, RO_XNOR = 5,
  RO_ANDNY = 6, RO_ANDYN = 7, RO_ORNY = 8, RO_ORYN = 9
} ro_gate_op;
/* tmp = (0, c) + sa*ca + sb*cb (wrapping), written to out[n+1]; returns nothing else. */
void ro_gate_precombine(ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb, int32_t n);
void ro_gate(const ro_ctx* c, ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb);
/* bootsMUX(a,b,c) = a ? b : c : two woKS bootstraps summed + (0,1/8), one keyswitch. */
void ro_mux(const ro_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc);

/* ---- batches (OpenMP over independent ciphertexts; row stride W = n+1 words) ---- */
void ro_bootstrap_batch(const ro_ctx* c, int32_t* out, int32_t mu, const int32_t* in, size_t B);
void ro_gate_batch(const ro_ctx* c, ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb, size_t B);
void ro_mux_batch(const ro_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc, size_t B);
int ro_max_threads(void);
void ro_set_threads(int t);

/* ---- REDsec linear stage on LWE words (no bootstrap) ---- */
/* out[m] = pad_const_count[m]*pad + sum_k sgn[m][k] * in[k]   with sgn in {-1,0,+1} given as
 * sign byte (1 => +, 0 => -) and zero byte (1 => tap contributes zero_const instead).
 * This is the ENCRYPTED branch of BinFunc::Convolution::execute (BinFunc.cpp:195-320) and
 * IntFunc::Convolution::execute (IntFunc.cpp:207-308) for a fully-connected shape (window 1x1):
 *   BinFunc: zero/padding taps contribute 0; IntFunc: they contribute trivial -1/4096 on b.
 * in: [K][W], out: [M][W], sign/zero: [K][M] (filter index ((fh*ww+fw)*in_dep+di)*OutDepth+od). */
void ro_linear_fc(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero,
                  int32_t K, int32_t M, int32_t W, int32_t zero_tap_b);
/* x[i] += trivial(bias[i % depth]) : Quantize::execute's add_int (BinFunc.cpp:1064-1066). */
void ro_add_bias(int32_t* x, const int32_t* bias_torus, int32_t count, int32_t depth, int32_t W);

#ifdef __cplusplus
}
#endif
#endif
