/*
 * redsec_oracle.c -- CPU oracle (exact-integer CGGI restatement). TEST INFRASTRUCTURE ONLY.
 * See redsec_oracle.h for the scope statement and the "parity unpinned" note.
 *
 * Every function cites (a) the REDsec call site it serves (paths relative to /root/reference) and
 * (b) the TFHE v1.1 routine whose published behaviour it restates. TFHE itself is absent from the
 * reference tree and from this image; the restatement is validated by the known-answer tests in
 * tests/test_oracle_*.py.
 */
#include "redsec_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * Parameter sets
 * ---------------------------------------------------------------------------------------------- */

/* TFHE v1.1 new_default_gate_bootstrapping_parameters(128) [tfhe_gate_bootstrapping.cpp]. */
void ro_params_default128(ro_params* p) {
  p->n = 630; p->N = 1024; p->k = 1;
  p->bk_l = 3; p->bk_Bgbit = 7;
  p->ks_t = 8; p->ks_basebit = 2;
  p->lwe_stdev = ldexp(1.0, -15);
  p->bk_stdev = ldexp(1.0, -25);
}

/* client/gen_secure_keyset.cpp:70-91 redsec_params_small_v2 (the set main() selects at :97). */
void ro_params_redsec_small_v2(ro_params* p) {
  p->n = 350; p->N = 1024; p->k = 1;
  p->bk_l = 10; p->bk_Bgbit = 3;
  p->ks_t = 9; p->ks_basebit = 3;
  p->lwe_stdev = ldexp(1.0, -25);  /* kskey_std_dev: LweParams alpha_min */
  p->bk_stdev = ldexp(1.0, -30);   /* bootkey_std_dev */
}

/* client/gen_secure_keyset.cpp:47-68, 28-45, 9-26: the sets the reference defines beside the one it ships
 * ("for wide networks, the medium and large parameters are better suited", :96). */
void ro_params_redsec_small(ro_params* p) {
  p->n = 500; p->N = 1024; p->k = 1;
  p->bk_l = 3; p->bk_Bgbit = 10;
  p->ks_t = 18; p->ks_basebit = 1;
  p->lwe_stdev = ldexp(1.0, -25);
  p->bk_stdev = ldexp(1.0, -36);
}
void ro_params_redsec_medium(ro_params* p) {
  p->n = 3072; p->N = 4096; p->k = 1;
  p->bk_l = 3; p->bk_Bgbit = 10;
  p->ks_t = 18; p->ks_basebit = 1;
  p->lwe_stdev = ldexp(1.0, -40);
  p->bk_stdev = ldexp(1.0, -45);
}
void ro_params_redsec_large(ro_params* p) {
  p->n = 6144; p->N = 8192; p->k = 1;
  p->bk_l = 3; p->bk_Bgbit = 10;
  p->ks_t = 18; p->ks_basebit = 1;
  p->lwe_stdev = ldexp(1.0, -41);
  p->bk_stdev = ldexp(1.0, -46);
}

/* ------------------------------------------------------------------------------------------------
 * Torus helpers. TFHE numeric_functions.cpp: modSwitchToTorus32 / modSwitchFromTorus32 /
 * approxPhase. REDsec call sites: BinOps_enc.cpp:137,184,190,293; client/encrypt_image.cpp:77;
 * client/decrypt_image.cpp:52. The 2N=2048 instance is corroborated by lib/GPU/gates.cu:39-42.
 * ---------------------------------------------------------------------------------------------- */
int32_t ro_modswitch_to_torus32(int32_t mu, int32_t Msize) {
  uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
  uint64_t phase64 = (uint64_t)(int64_t)mu * interv;
  return (int32_t)(phase64 >> 32);
}

int32_t ro_modswitch_from_torus32(int32_t phase, int32_t Msize) {
  uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
  uint64_t half_interval = interv / 2;
  uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half_interval;
  return (int32_t)(phase64 / interv);
}

int32_t ro_approx_phase(int32_t phase, int32_t Msize) {
  uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
  uint64_t half_interval = interv / 2;
  uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half_interval;
  phase64 -= phase64 % interv;
  return (int32_t)(phase64 >> 32);
}

/* ------------------------------------------------------------------------------------------------
 * PRNG: xoshiro256** seeded by splitmix64; Box-Muller gaussian. TFHE draws from
 * std::default_random_engine, which cannot be reproduced; keys and fresh ciphertexts are INPUTS
 * to the hot path, so any generator serves the oracle.
 * ---------------------------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t* x) {
  uint64_t z = (*x += UINT64_C(0x9E3779B97F4A7C15));
  z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
  z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
  return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

void ro_rng_seed(ro_rng* r, uint64_t seed) {
  uint64_t x = seed;
  for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&x);
  r->has_spare = 0; r->spare = 0.0;
}
static uint64_t ro_rng_u64(ro_rng* r) {
  uint64_t* s = r->s;
  uint64_t result = rotl64(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t; s[3] = rotl64(s[3], 45);
  return result;
}
uint32_t ro_rng_u32(ro_rng* r) { return (uint32_t)(ro_rng_u64(r) >> 32); }

double ro_rng_gaussian(ro_rng* r, double sigma) {
  if (r->has_spare) { r->has_spare = 0; return r->spare * sigma; }
  double u1, u2;
  do { u1 = (double)(ro_rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); } while (u1 <= 0.0);
  u2 = (double)(ro_rng_u64(r) >> 11) * (1.0 / 9007199254740992.0);
  double mag = sqrt(-2.0 * log(u1));
  double two_pi = 6.283185307179586476925286766559;
  r->spare = mag * sin(two_pi * u2); r->has_spare = 1;
  return mag * cos(two_pi * u2) * sigma;
}

/* TFHE numeric_functions.cpp: dtot32 + gaussian32. */
static int32_t dtot32(double d) {
  return (int32_t)(int64_t)((d - (double)(int64_t)d) * 4294967296.0);
}
int32_t ro_gaussian32(ro_rng* r, int32_t message, double sigma) {
  double err = ro_rng_gaussian(r, sigma);
  return (int32_t)((uint32_t)message + (uint32_t)dtot32(err));
}

/* ------------------------------------------------------------------------------------------------
 * LWE samples. TFHE lwe-functions.cpp: lweSymEncrypt / lwePhase / lweSymDecrypt /
 * lweNoiselessTrivial. REDsec call sites: client/encrypt_image.cpp:77 (alpha = 2^-15),
 * client/decrypt_image.cpp:52, BinOps_enc.cpp:138,294.
 * ---------------------------------------------------------------------------------------------- */
void ro_lwe_encrypt(int32_t* out, int32_t mu, double alpha, const int32_t* key, int32_t n, ro_rng* rng) {
  uint32_t b = (uint32_t)ro_gaussian32(rng, mu, alpha);
  for (int32_t i = 0; i < n; ++i) {
    uint32_t a = ro_rng_u32(rng);
    out[i] = (int32_t)a;
    b += a * (uint32_t)key[i];
  }
  out[n] = (int32_t)b;
}

int32_t ro_lwe_phase(const int32_t* sample, const int32_t* key, int32_t n) {
  uint32_t axs = 0;
  for (int32_t i = 0; i < n; ++i) axs += (uint32_t)sample[i] * (uint32_t)key[i];
  return (int32_t)((uint32_t)sample[n] - axs);
}

int32_t ro_lwe_decrypt(const int32_t* sample, const int32_t* key, int32_t n, int32_t Msize) {
  return ro_approx_phase(ro_lwe_phase(sample, key, n), Msize);
}

void ro_lwe_trivial(int32_t* out, int32_t mu, int32_t n) {
  memset(out, 0, sizeof(int32_t) * (size_t)n);
  out[n] = mu;
}

/* ------------------------------------------------------------------------------------------------
 * Exact negacyclic products.
 * ---------------------------------------------------------------------------------------------- */

/* Definitional: TFHE polynomials.cpp torusPolynomialMultNaive (int poly x torus poly in
 * Z[X]/(X^N+1), coefficients wrap mod 2^32). */
void ro_negacyclic_mul_schoolbook(int32_t* out, const int32_t* a_small, const int32_t* b_torus, int32_t N) {
  for (int32_t j = 0; j < N; ++j) {
    uint32_t acc = 0;
    for (int32_t i = 0; i <= j; ++i) acc += (uint32_t)a_small[i] * (uint32_t)b_torus[j - i];
    for (int32_t i = j + 1; i < N; ++i) acc -= (uint32_t)a_small[i] * (uint32_t)b_torus[N + j - i];
    out[j] = (int32_t)acc;
  }
}

/* 64-bit NTT over P = 2^64 - 2^32 + 1 (generator 7). Independent of the product's arithmetic
 * (the HIP path uses a different prime and FP64 registers), which is the point of an oracle. */
#define GL_P UINT64_C(0xFFFFFFFF00000001)
typedef unsigned __int128 u128;

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return (a >= b) ? a - b : a + (GL_P - b); }
/* 128 -> 64 reduction using 2^64 = 2^32 - 1 and 2^96 = -1 (mod P). */
static inline uint64_t gl_reduce128(u128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hh = hi >> 32, hl = hi & UINT64_C(0xFFFFFFFF);
  uint64_t t0 = lo - hh;
  if (lo < hh) t0 -= UINT64_C(0xFFFFFFFF); /* borrowed 2^64: give back 2^32-1, i.e. add P */
  uint64_t t1 = hl * UINT64_C(0xFFFFFFFF);
  uint64_t r = t0 + t1;
  if (r < t1) r += UINT64_C(0xFFFFFFFF);   /* carried 2^64 = 2^32-1 */
  if (r >= GL_P) r -= GL_P;
  return r;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }
static uint64_t gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) { if (e & 1) r = gl_mul(r, b); b = gl_mul(b, b); e >>= 1; }
  return r;
}
static inline uint64_t gl_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v : GL_P - (uint64_t)(-v); }
static inline int64_t gl_to_centered(uint64_t v) { return (v > GL_P / 2) ? -(int64_t)(GL_P - v) : (int64_t)v; }

typedef struct gl_tables {
  int32_t N;
  uint64_t* psi_rev;     /* psi^bitrev(i), i<N  */
  uint64_t* psi_inv_rev; /* psi^-bitrev(i)      */
  uint64_t n_inv;
} gl_tables;

static uint32_t bitrev(uint32_t x, int bits) {
  uint32_t r = 0;
  for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

static gl_tables* gl_tables_create(int32_t N) {
  gl_tables* t = (gl_tables*)malloc(sizeof(gl_tables));
  int logN = 0; while ((1 << logN) < N) ++logN;
  t->N = N;
  t->psi_rev = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
  t->psi_inv_rev = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
  uint64_t psi = gl_pow(7, (GL_P - 1) / (uint64_t)(2 * N)); /* primitive 2N-th root */
  uint64_t psi_inv = gl_pow(psi, GL_P - 2);
  for (int32_t i = 0; i < N; ++i) {
    uint32_t r = bitrev((uint32_t)i, logN);
    t->psi_rev[i] = gl_pow(psi, r);
    t->psi_inv_rev[i] = gl_pow(psi_inv, r);
  }
  t->n_inv = gl_pow((uint64_t)N, GL_P - 2);
  return t;
}
static void gl_tables_destroy(gl_tables* t) {
  if (!t) return;
  free(t->psi_rev); free(t->psi_inv_rev); free(t);
}

/* Merged-twist negacyclic forward transform (Cooley-Tukey butterflies, natural -> bit-reversed). */
static void gl_ntt_forward(uint64_t* a, const gl_tables* tb) {
  int32_t N = tb->N, t = N;
  for (int32_t m = 1; m < N; m <<= 1) {
    t >>= 1;
    for (int32_t i = 0; i < m; ++i) {
      int32_t j1 = 2 * i * t;
      uint64_t S = tb->psi_rev[m + i];
      for (int32_t j = j1; j < j1 + t; ++j) {
        uint64_t U = a[j], V = gl_mul(a[j + t], S);
        a[j] = gl_add(U, V);
        a[j + t] = gl_sub(U, V);
      }
    }
  }
}
/* Inverse (Gentleman-Sande butterflies, bit-reversed -> natural), scaled by 1/N. */
static void gl_ntt_inverse(uint64_t* a, const gl_tables* tb) {
  int32_t N = tb->N, t = 1;
  for (int32_t m = N; m > 1; m >>= 1) {
    int32_t j1 = 0, h = m >> 1;
    for (int32_t i = 0; i < h; ++i) {
      uint64_t S = tb->psi_inv_rev[h + i];
      for (int32_t j = j1; j < j1 + t; ++j) {
        uint64_t U = a[j], V = a[j + t];
        a[j] = gl_add(U, V);
        a[j + t] = gl_mul(gl_sub(U, V), S);
      }
      j1 += 2 * t;
    }
    t <<= 1;
  }
  for (int32_t j = 0; j < N; ++j) a[j] = gl_mul(a[j], tb->n_inv);
}

void ro_negacyclic_mul_ntt(int32_t* out, const int32_t* a_small, const int32_t* b_torus, int32_t N) {
  gl_tables* tb = gl_tables_create(N);
  uint64_t* fa = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
  uint64_t* fb = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)N);
  for (int32_t i = 0; i < N; ++i) { fa[i] = gl_from_i64(a_small[i]); fb[i] = gl_from_i64(b_torus[i]); }
  gl_ntt_forward(fa, tb); gl_ntt_forward(fb, tb);
  for (int32_t i = 0; i < N; ++i) fa[i] = gl_mul(fa[i], fb[i]);
  gl_ntt_inverse(fa, tb);
  for (int32_t i = 0; i < N; ++i) out[i] = (int32_t)(uint32_t)(uint64_t)gl_to_centered(fa[i]);
  free(fa); free(fb); gl_tables_destroy(tb);
}

/* ------------------------------------------------------------------------------------------------
 * Keys. TFHE: lweKeyGen, tLweKeyGen, tGswSymEncryptInt (tGswEncryptZero + tGswAddMuIntH),
 * tLweSymEncryptZero, lweCreateKeySwitchKey, tfhe_createLweBootstrappingKey. REDsec call site:
 * client/gen_secure_keyset.cpp:94-120 (new_random_gate_bootstrapping_secret_keyset).
 * ---------------------------------------------------------------------------------------------- */
size_t ro_bk_words(const ro_params* p) {
  return (size_t)p->n * (size_t)((p->k + 1) * p->bk_l) * (size_t)(p->k + 1) * (size_t)p->N;
}
size_t ro_ksk_words(const ro_params* p) {
  return (size_t)(p->k * p->N) * (size_t)p->ks_t * ((size_t)1 << p->ks_basebit) * (size_t)(p->n + 1);
}

/* Words [first, first + count) of the SYNTHETIC key the product generates on the device for size tests of the large rings
 * (include/redsec_hip.h, rs_load_synthetic_keys): word k = high half of splitmix64(seed + k), restated here so that the
 * oracle is fed the same key words without a multi-gigabyte array crossing from the test into the product. Not a TFHE
 * routine: a test generator (SplitMix64: Steele, Lea & Flood 2014, the published constants). */
void ro_synthetic_key_words(uint64_t seed, uint64_t first, uint64_t count, int32_t* out) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)count; ++i) {
    uint64_t z = seed + (first + (uint64_t)i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    out[i] = (int32_t)(uint32_t)(z >> 32);
  }
}

/* b += s * a (negacyclic), s binary: torusPolynomialAddMulR with an IntPolynomial key. */
static void addmul_binary_key(int32_t* b, const int32_t* s, const int32_t* a, int32_t N) {
  for (int32_t i = 0; i < N; ++i) {
    if (!s[i]) continue;
    for (int32_t j = 0; j < N - i; ++j) b[i + j] = (int32_t)((uint32_t)b[i + j] + (uint32_t)a[j]);
    for (int32_t j = N - i; j < N; ++j) b[i + j - N] = (int32_t)((uint32_t)b[i + j - N] - (uint32_t)a[j]);
  }
}

void ro_keygen(const ro_params* p, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key, int32_t* bk, int32_t* ksk) {
  const int32_t n = p->n, N = p->N, k = p->k, l = p->bk_l, Bgbit = p->bk_Bgbit;
  const int32_t t = p->ks_t, basebit = p->ks_basebit, base = 1 << basebit;
  const int32_t kpl = (k + 1) * l, W = n + 1;
  ro_rng rng; ro_rng_seed(&rng, seed);
  for (int32_t i = 0; i < n; ++i) lwe_key[i] = (int32_t)(ro_rng_u32(&rng) & 1u);      /* lweKeyGen */
  for (int32_t i = 0; i < k * N; ++i) tlwe_key[i] = (int32_t)(ro_rng_u32(&rng) & 1u); /* tLweKeyGen */

  /* Bootstrapping key: BK_i = TGSW(s_i). Row p = c*l + j is a TRLWE encryption of zero
   * (tLweSymEncryptZero: a uniform, b = e + sum a_c * s'_c) plus s_i * 2^(32-(j+1)Bgbit) on the
   * constant coefficient of component c (tGswAddMuIntH). */
  for (int32_t i = 0; i < n; ++i) {
    for (int32_t row = 0; row < kpl; ++row) {
      int32_t* smp = bk + (((size_t)i * kpl + row) * (size_t)(k + 1)) * (size_t)N;
      int32_t* bpoly = smp + (size_t)k * N;
      for (int32_t j = 0; j < N; ++j) bpoly[j] = ro_gaussian32(&rng, 0, p->bk_stdev);
      for (int32_t c = 0; c < k; ++c) {
        int32_t* apoly = smp + (size_t)c * N;
        for (int32_t j = 0; j < N; ++j) apoly[j] = (int32_t)ro_rng_u32(&rng);
        addmul_binary_key(bpoly, tlwe_key + (size_t)c * N, apoly, N);
      }
      int32_t comp = row / l, dig = row % l;
      uint32_t h = (uint32_t)1 << (32 - (dig + 1) * Bgbit);
      smp[(size_t)comp * N] = (int32_t)((uint32_t)smp[(size_t)comp * N] + (uint32_t)lwe_key[i] * h);
    }
  }

  /* Keyswitch key from the extracted key (tLweExtractKey: key[c*N+j] = s'_c[j]) to the LWE key.
   * lweCreateKeySwitchKey: ks[i][j][v] encrypts v * s'_i * 2^(32-(j+1)basebit); v = 0 is the
   * noiseless trivial zero (never subtracted by lweKeySwitchTranslate_fromArray). */
  for (int32_t i = 0; i < k * N; ++i)
    for (int32_t j = 0; j < t; ++j)
      for (int32_t v = 0; v < base; ++v) {
        int32_t* row = ksk + ((((size_t)i * t + j) * (size_t)base) + v) * (size_t)W;
        if (v == 0) { ro_lwe_trivial(row, 0, n); continue; }
        uint32_t mess = ((uint32_t)tlwe_key[i] * (uint32_t)v) << (32 - (j + 1) * basebit);
        ro_lwe_encrypt(row, (int32_t)mess, p->lwe_stdev, lwe_key, n, &rng);
      }
}

/* ------------------------------------------------------------------------------------------------
 * Evaluation context
 * ---------------------------------------------------------------------------------------------- */
struct ro_ctx {
  ro_params p;
  const int32_t* bk;   /* borrowed */
  const int32_t* ksk;  /* borrowed */
  gl_tables* tb;
  uint64_t* bk_ntt;    /* [n][kpl][k+1][N] forward transforms of BK rows */
  int use_schoolbook;
  int use_fft;
  double* bk_fft;      /* [n][kpl][k+1][N/2][2] folded complex FFT of BK rows, scaled by 2/N (built on demand) */
  double* fft_tw;      /* [N/2][2] exp(2 pi i j / (N/2)); [N/2][2] twist exp(i pi j / N) after it */
  uint32_t offset;     /* gadget rounding offset */
};

ro_ctx* ro_ctx_create(const ro_params* p, const int32_t* bk, const int32_t* ksk) {
  ro_ctx* c = (ro_ctx*)calloc(1, sizeof(ro_ctx));
  c->p = *p; c->bk = bk; c->ksk = ksk;
  c->tb = gl_tables_create(p->N);
  size_t words = ro_bk_words(p);
  c->bk_ntt = (uint64_t*)malloc(sizeof(uint64_t) * words);
  size_t polys = words / (size_t)p->N;
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < (long long)polys; ++q) {
    uint64_t* dst = c->bk_ntt + (size_t)q * p->N;
    const int32_t* src = bk + (size_t)q * p->N;
    for (int32_t j = 0; j < p->N; ++j) dst[j] = gl_from_i64(src[j]);
    gl_ntt_forward(dst, c->tb);
  }
  /* TGswParams: offset = sum_{i=1..l} (Bg/2) * 2^(32 - i*Bgbit) */
  uint32_t off = 0, halfBg = (uint32_t)1 << (p->bk_Bgbit - 1);
  for (int32_t i = 1; i <= p->bk_l; ++i) off += halfBg << (32 - i * p->bk_Bgbit);
  c->offset = off;
  return c;
}
void ro_ctx_destroy(ro_ctx* c) {
  if (!c) return;
  gl_tables_destroy(c->tb); free(c->bk_ntt); free(c->bk_fft); free(c->fft_tw); free(c);
}
void ro_ctx_set_schoolbook(ro_ctx* c, int use_schoolbook) { c->use_schoolbook = use_schoolbook; }

/* ---- double-precision path (ro_ctx_set_fft) ----------------------------------------------------
 * Negacyclic product of real polynomials through an M = N/2 point complex FFT: fold z_j = a_j + i a_{j+M},
 * twist by zeta^j (zeta = exp(i pi / N)), cyclic DFT; then products are pointwise. Plain iterative
 * radix-2 (decimation in frequency forward, decimation in time inverse, so no bit reversal is needed
 * between them). Planar layout (re[M] then im[M]) and one contiguous twiddle run per stage, so that
 * the compiler vectorises the butterflies; the hot loops are cloned for AVX2+FMA hosts (run-time
 * dispatch, the baseline build stays x86-64-v2). This is the timed CPU baseline: it should not be
 * slower than it has to be. */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define RO_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define RO_CLONES
#endif
/* fft_tw: [stage s = 0 .. log2(M)-1] runs of half_s = M >> (s+1) entries (re run, then im run) at offset
 * 2 * (M - 2 * half_s) = stages laid out back to back; after them (offset 2M) the twist: re[M], im[M]. */
static void fft_tables(ro_ctx* c) {
  const int32_t M = c->p.N / 2;
  c->fft_tw = (double*)malloc(sizeof(double) * 4 * (size_t)M);
  const long double pi = 3.141592653589793238462643383279502884L;
  size_t off = 0;
  for (int32_t half = M / 2, step = 1; half >= 1; half >>= 1, step <<= 1) {
    for (int32_t j = 0; j < half; ++j) {
      c->fft_tw[off + j] = (double)cosl(2 * pi * (long double)(j * step) / M);
      c->fft_tw[off + half + j] = (double)sinl(2 * pi * (long double)(j * step) / M);
    }
    off += 2 * (size_t)half;
  }
  for (int32_t j = 0; j < M; ++j) {
    c->fft_tw[2 * M + j] = (double)cosl(pi * j / c->p.N);
    c->fft_tw[3 * M + j] = (double)sinl(pi * j / c->p.N);
  }
}
/* in-place forward DFT (e^{+2 pi i jk/M}) of re[M], im[M]; output in bit-reversed order */
RO_CLONES static void fft_dif(double* restrict re, double* restrict im, int32_t M, const double* restrict tw) {
  for (int32_t half = M / 2; half >= 1; half >>= 1) {
    const double* restrict wr = tw; const double* restrict wi = tw + half;
    for (int32_t base = 0; base < M; base += 2 * half) {
      double* restrict ar = re + base; double* restrict ai = im + base;
      double* restrict br = ar + half; double* restrict bi = ai + half;
      for (int32_t j = 0; j < half; ++j) {
        const double dr = ar[j] - br[j], di = ai[j] - bi[j];
        ar[j] += br[j]; ai[j] += bi[j];
        br[j] = dr * wr[j] - di * wi[j]; bi[j] = dr * wi[j] + di * wr[j];
      }
    }
    tw += 2 * half;
  }
}
/* in-place inverse (conjugate twiddles) from bit-reversed order back to natural order, unscaled */
RO_CLONES static void fft_dit_inv(double* restrict re, double* restrict im, int32_t M, const double* restrict tw_end) {
  const double* tw = tw_end;                    /* stage tables are walked backwards: half = 1, 2, ... */
  for (int32_t half = 1; half < M; half <<= 1) {
    tw -= 2 * half;
    const double* restrict wr = tw; const double* restrict wi = tw + half;
    for (int32_t base = 0; base < M; base += 2 * half) {
      double* restrict ar = re + base; double* restrict ai = im + base;
      double* restrict br = ar + half; double* restrict bi = ai + half;
      for (int32_t j = 0; j < half; ++j) {
        const double tr = br[j] * wr[j] + bi[j] * wi[j], ti = bi[j] * wr[j] - br[j] * wi[j];
        br[j] = ar[j] - tr; bi[j] = ai[j] - ti;
        ar[j] += tr; ai[j] += ti;
      }
    }
  }
}
/* z = planar (re[M], im[M]) transform of the folded, twisted integer polynomial */
RO_CLONES static void fft_forward_i32(double* restrict z, const int32_t* restrict poly, int32_t M, const double* restrict tw, double scale) {
  const double* restrict tr = tw + 2 * M; const double* restrict ti = tw + 3 * M;
  double* restrict re = z; double* restrict im = z + M;
  for (int32_t j = 0; j < M; ++j) {
    const double a = scale * (double)poly[j], b = scale * (double)poly[j + M];
    re[j] = a * tr[j] - b * ti[j];
    im[j] = a * ti[j] + b * tr[j];
  }
  fft_dif(re, im, M, tw);
}
/* dst += x (.) k, all planar */
RO_CLONES static void fft_mac(double* restrict dst, const double* restrict x, const double* restrict k, int32_t M) {
  const double* restrict xr = x; const double* restrict xi = x + M;
  const double* restrict kr = k; const double* restrict ki = k + M;
  double* restrict dr = dst; double* restrict di = dst + M;
  for (int32_t j = 0; j < M; ++j) {
    dr[j] += xr[j] * kr[j] - xi[j] * ki[j];
    di[j] += xr[j] * ki[j] + xi[j] * kr[j];
  }
}
/* untwist by conj(zeta^j): out_re[j] -> coefficient j, out_im[j] -> coefficient j + M (still doubles) */
RO_CLONES static void fft_untwist(double* restrict z, int32_t M, const double* restrict tw) {
  const double* restrict tr = tw + 2 * M; const double* restrict ti = tw + 3 * M;
  double* restrict re = z; double* restrict im = z + M;
  for (int32_t j = 0; j < M; ++j) {
    const double a = re[j] * tr[j] + im[j] * ti[j];
    const double b = im[j] * tr[j] - re[j] * ti[j];
    re[j] = a; im[j] = b;
  }
}
/* dst[j] += rint(z[j]) mod 2^32 for |z[j]| < 2^51: the low mantissa bits of z + 1.5 * 2^52 are rint(z)
 * (round-to-nearest-even, as llrint under the default rounding mode) -- no libm call, vectorisable */
RO_CLONES static void fft_round_add(int32_t* restrict dst, const double* restrict z, int32_t N) {
  for (int32_t j = 0; j < N; ++j) {
    union { double d; uint64_t u; } t;
    t.d = z[j] + 6755399441055744.0;
    dst[j] = (int32_t)((uint32_t)dst[j] + (uint32_t)t.u);
  }
}
void ro_ctx_set_fft(ro_ctx* c, int use_fft) {
  c->use_fft = use_fft;
  if (!use_fft || c->bk_fft) return;
  fft_tables(c);
  const size_t polys = ro_bk_words(&c->p) / (size_t)c->p.N;
  c->bk_fft = (double*)malloc(sizeof(double) * polys * (size_t)c->p.N);
#pragma omp parallel for schedule(static)
  for (long long q = 0; q < (long long)polys; ++q)
    fft_forward_i32(c->bk_fft + (size_t)q * c->p.N, c->bk + (size_t)q * c->p.N, c->p.N / 2, c->fft_tw, 2.0 / c->p.N);   /* 1/M folded into the key */
}

/* TFHE polynomials.cpp torusPolynomialMulByXaiMinusOne: result = (X^a - 1) * source, 0 <= a < 2N. */
static void mul_by_xai_minus_one(int32_t* out, int32_t a, const int32_t* in, int32_t N) {
  if (a < N) {
    for (int32_t i = 0; i < a; ++i) out[i] = (int32_t)(0u - (uint32_t)in[i - a + N] - (uint32_t)in[i]);
    for (int32_t i = a; i < N; ++i) out[i] = (int32_t)((uint32_t)in[i - a] - (uint32_t)in[i]);
  } else {
    int32_t aa = a - N;
    for (int32_t i = 0; i < aa; ++i) out[i] = (int32_t)((uint32_t)in[i - aa + N] - (uint32_t)in[i]);
    for (int32_t i = aa; i < N; ++i) out[i] = (int32_t)(0u - (uint32_t)in[i - aa] - (uint32_t)in[i]);
  }
}
/* torusPolynomialMulByXai: result = X^a * source. */
static void mul_by_xai(int32_t* out, int32_t a, const int32_t* in, int32_t N) {
  if (a < N) {
    for (int32_t i = 0; i < a; ++i) out[i] = (int32_t)(0u - (uint32_t)in[i - a + N]);
    for (int32_t i = a; i < N; ++i) out[i] = in[i - a];
  } else {
    int32_t aa = a - N;
    for (int32_t i = 0; i < aa; ++i) out[i] = in[i - aa + N];
    for (int32_t i = aa; i < N; ++i) out[i] = (int32_t)(0u - (uint32_t)in[i - aa]);
  }
}

/* TFHE tgsw-functions.cpp tGswTorus32PolynomialDecompH: digits in [-Bg/2, Bg/2). */
static void decomp_h(int32_t* digits /* [l][N] */, const int32_t* poly, const ro_ctx* c) {
  const int32_t N = c->p.N, l = c->p.bk_l, Bgbit = c->p.bk_Bgbit;
  const uint32_t mask = ((uint32_t)1 << Bgbit) - 1;
  const int32_t halfBg = 1 << (Bgbit - 1);
  for (int32_t q = 0; q < l; ++q) {
    int decal = 32 - (q + 1) * Bgbit;
    for (int32_t j = 0; j < N; ++j) {
      uint32_t u = (uint32_t)poly[j] + c->offset;
      digits[(size_t)q * N + j] = (int32_t)((u >> decal) & mask) - halfBg;
    }
  }
}

/* One CMUX step of tfhe_blindRotate_FFT / tfhe_MuxRotate_FFT:
 *   acc <- acc + BK_i (.) ((X^barai - 1) * acc)
 * with tGswFFTExternMulToTLwe evaluated exactly (row order p = c*l + j). */
static void cmux_step(const ro_ctx* c, int32_t* acc, int32_t i, int32_t barai, int32_t* scratch_i32, uint64_t* scratch_u64) {
  const int32_t N = c->p.N, k = c->p.k, l = c->p.bk_l, kpl = (k + 1) * l;
  int32_t* diff = scratch_i32;                    /* [k+1][N] */
  int32_t* digits = scratch_i32 + (size_t)(k + 1) * N; /* [kpl][N] */
  for (int32_t comp = 0; comp <= k; ++comp) {
    mul_by_xai_minus_one(diff + (size_t)comp * N, barai, acc + (size_t)comp * N, N);
    decomp_h(digits + (size_t)comp * l * N, diff + (size_t)comp * N, c);
  }
  if (c->use_schoolbook) {
    int32_t* prod = (int32_t*)scratch_u64;
    for (int32_t col = 0; col <= k; ++col)
      for (int32_t row = 0; row < kpl; ++row) {
        const int32_t* bkpoly = c->bk + ((((size_t)i * kpl + row) * (size_t)(k + 1)) + col) * (size_t)N;
        ro_negacyclic_mul_schoolbook(prod, digits + (size_t)row * N, bkpoly, N);
        int32_t* dst = acc + (size_t)col * N;
        for (int32_t j = 0; j < N; ++j) dst[j] = (int32_t)((uint32_t)dst[j] + (uint32_t)prod[j]);
      }
    return;
  }
  if (c->use_fft) {
    const int32_t M = N / 2;
    double* zd = (double*)scratch_u64;                /* [N] one digit transform (planar: re[M], im[M]) */
    double* zacc = zd + N;                            /* [k+1][N] */
    memset(zacc, 0, sizeof(double) * (size_t)(k + 1) * N);
    for (int32_t row = 0; row < kpl; ++row) {
      fft_forward_i32(zd, digits + (size_t)row * N, M, c->fft_tw, 1.0);
      for (int32_t col = 0; col <= k; ++col)
        fft_mac(zacc + (size_t)col * N, zd, c->bk_fft + ((((size_t)i * kpl + row) * (size_t)(k + 1)) + col) * (size_t)N, M);
    }
    for (int32_t col = 0; col <= k; ++col) {
      double* z = zacc + (size_t)col * N;
      fft_dit_inv(z, z + M, M, c->fft_tw + 2 * M - 2);   /* the last stage run (half = 1) ends at 2M - 2 */
      fft_untwist(z, M, c->fft_tw);
      int32_t* dst = acc + (size_t)col * N;
      fft_round_add(dst, z, N);                        /* planar re[] = coefficients 0..M-1, im[] = M..N-1 */
    }
    return;
  }
  uint64_t* fd = scratch_u64;                         /* [N] one digit transform */
  uint64_t* facc = scratch_u64 + N;                   /* [k+1][N] */
  memset(facc, 0, sizeof(uint64_t) * (size_t)(k + 1) * N);
#ifdef _OPENMP
  if (N >= 4096 && !omp_in_parallel() && omp_get_max_threads() > 1) {
    /* A single ciphertext on a large ring (the full-size parity tests): the kpl row products of a step are independent, so
     * they go to kpl threads and are summed afterwards -- the same exact values in the same field, any order. */
    uint64_t* rowprod = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)kpl * (size_t)(k + 2) * N);
    const int team = kpl < omp_get_max_threads() ? kpl : omp_get_max_threads();   /* never the whole machine: a team is woken per step */
#pragma omp parallel for schedule(static) num_threads(team)
    for (int32_t row = 0; row < kpl; ++row) {
      uint64_t* f = rowprod + (size_t)row * (size_t)(k + 2) * N;
      const int32_t* d = digits + (size_t)row * N;
      for (int32_t j = 0; j < N; ++j) f[j] = gl_from_i64(d[j]);
      gl_ntt_forward(f, c->tb);
      for (int32_t col = 0; col <= k; ++col) {
        const uint64_t* bkp = c->bk_ntt + ((((size_t)i * kpl + row) * (size_t)(k + 1)) + col) * (size_t)N;
        uint64_t* dst = f + (size_t)(col + 1) * N;
        for (int32_t j = 0; j < N; ++j) dst[j] = gl_mul(f[j], bkp[j]);
      }
    }
    for (int32_t row = 0; row < kpl; ++row)
      for (int32_t col = 0; col <= k; ++col) {
        const uint64_t* src = rowprod + (size_t)row * (size_t)(k + 2) * N + (size_t)(col + 1) * N;
        uint64_t* dst = facc + (size_t)col * N;
        for (int32_t j = 0; j < N; ++j) dst[j] = gl_add(dst[j], src[j]);
      }
    free(rowprod);
  } else
#endif
  for (int32_t row = 0; row < kpl; ++row) {
    const int32_t* d = digits + (size_t)row * N;
    for (int32_t j = 0; j < N; ++j) fd[j] = gl_from_i64(d[j]);
    gl_ntt_forward(fd, c->tb);
    for (int32_t col = 0; col <= k; ++col) {
      const uint64_t* bkp = c->bk_ntt + ((((size_t)i * kpl + row) * (size_t)(k + 1)) + col) * (size_t)N;
      uint64_t* dst = facc + (size_t)col * N;
      for (int32_t j = 0; j < N; ++j) dst[j] = gl_add(dst[j], gl_mul(fd[j], bkp[j]));
    }
  }
  for (int32_t col = 0; col <= k; ++col) {
    uint64_t* src = facc + (size_t)col * N;
    gl_ntt_inverse(src, c->tb);
    int32_t* dst = acc + (size_t)col * N;
    /* |true sum| <= (k+1) l N (Bg/2) 2^31 < 2^50 << P/2, so the centered lift is the integer. */
    for (int32_t j = 0; j < N; ++j) dst[j] = (int32_t)((uint32_t)dst[j] + (uint32_t)(uint64_t)gl_to_centered(src[j]));
  }
}

static size_t scratch_i32_words(const ro_params* p) { return (size_t)(p->k + 1) * p->N * (size_t)(1 + p->bk_l); }
static size_t scratch_u64_words(const ro_params* p) { return (size_t)(p->k + 2) * p->N; }

/* tfhe_bootstrap_woKS_FFT -> tfhe_blindRotateAndExtract_FFT -> tfhe_blindRotate_FFT. */
static void blind_rotate_tv(const ro_ctx* c, int32_t* acc, int32_t mu, const int32_t* tv, const int32_t* in, int32_t steps) {
  const int32_t N = c->p.N, k = c->p.k, n = c->p.n, Nx2 = 2 * N;
  int32_t* si = (int32_t*)malloc(sizeof(int32_t) * scratch_i32_words(&c->p));
  uint64_t* su = (uint64_t*)malloc(sizeof(uint64_t) * scratch_u64_words(&c->p));
  int32_t* testvect = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  int32_t barb = ro_modswitch_from_torus32(in[n], Nx2);
  /* tfhe_bootstrap_woKS_FFT fills the test vector with mu; tfhe_blindRotateAndExtract_FFT itself takes
   * an arbitrary test polynomial v (the programmable form: out = v[phase] on [0, 1/2), -v[phase - N] beyond) */
  for (int32_t j = 0; j < N; ++j) testvect[j] = tv ? tv[j] : mu;
  memset(acc, 0, sizeof(int32_t) * (size_t)(k + 1) * N);
  if (barb != 0) mul_by_xai(acc + (size_t)k * N, Nx2 - barb, testvect, N);
  else memcpy(acc + (size_t)k * N, testvect, sizeof(int32_t) * (size_t)N);
  if (steps < 0 || steps > n) steps = n;
  for (int32_t i = 0; i < steps; ++i) {
    int32_t barai = ro_modswitch_from_torus32(in[i], Nx2);
    if (barai == 0) continue; /* tfhe_blindRotate_FFT skips the identity CMUX */
    cmux_step(c, acc, i, barai, si, su);
  }
  free(si); free(su); free(testvect);
}

static void blind_rotate(const ro_ctx* c, int32_t* acc, int32_t mu, const int32_t* in, int32_t steps) {
  blind_rotate_tv(c, acc, mu, NULL, in, steps);
}

void ro_blind_rotate_acc(const ro_ctx* c, int32_t* acc_out, int32_t mu, const int32_t* in, int32_t steps) {
  blind_rotate(c, acc_out, mu, in, steps);
}

/* tLweExtractLweSampleIndex(index = 0). */
static void sample_extract0(const ro_ctx* c, int32_t* out, const int32_t* acc) {
  const int32_t N = c->p.N, k = c->p.k;
  for (int32_t comp = 0; comp < k; ++comp) {
    const int32_t* a = acc + (size_t)comp * N;
    out[(size_t)comp * N] = a[0];
    for (int32_t j = 1; j < N; ++j) out[(size_t)comp * N + j] = (int32_t)(0u - (uint32_t)a[N - j]);
  }
  out[(size_t)k * N] = acc[(size_t)k * N];
}

void ro_bootstrap_wo_ks(const ro_ctx* c, int32_t* out_extracted, int32_t mu, const int32_t* in) {
  int32_t* acc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->p.k + 1) * c->p.N);
  blind_rotate(c, acc, mu, in, -1);
  sample_extract0(c, out_extracted, acc);
  free(acc);
}

/* Programmable bootstrap: tfhe_blindRotateAndExtract_FFT with test polynomial `testvect` (N words), then
 * lweKeySwitch. REDsec: the corrected Quantize::relu_shift (IntFunc.cpp:934-973) evaluates
 * clamp((slope x + bias) >> slope_bits, 0, 2^shift_bits - 1) as ONE such bootstrap per neuron. */
void ro_bootstrap_lut(const ro_ctx* c, int32_t* out, const int32_t* testvect, const int32_t* in) {
  int32_t* acc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->p.k + 1) * c->p.N);
  int32_t* u = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->p.k * c->p.N + 1));
  blind_rotate_tv(c, acc, 0, testvect, in, -1);
  sample_extract0(c, u, acc);
  ro_keyswitch(c, out, u);
  free(acc); free(u);
}

/* TFHE lwe-keyswitch-functions.cpp lweKeySwitch + lweKeySwitchTranslate_fromArray. */
void ro_keyswitch(const ro_ctx* c, int32_t* out, const int32_t* in_extracted) {
  const int32_t n = c->p.n, Nk = c->p.k * c->p.N, t = c->p.ks_t, basebit = c->p.ks_basebit;
  const int32_t base = 1 << basebit, W = n + 1;
  const uint32_t prec_offset = (uint32_t)1 << (32 - (1 + basebit * t));
  const uint32_t mask = (uint32_t)base - 1;
  ro_lwe_trivial(out, in_extracted[Nk], n);
  for (int32_t i = 0; i < Nk; ++i) {
    uint32_t aibar = (uint32_t)in_extracted[i] + prec_offset;
    for (int32_t j = 0; j < t; ++j) {
      uint32_t aij = (aibar >> (32 - (j + 1) * basebit)) & mask;
      if (aij == 0) continue;
      const int32_t* row = c->ksk + ((((size_t)i * t + j) * (size_t)base) + aij) * (size_t)W;
      for (int32_t w = 0; w < W; ++w) out[w] = (int32_t)((uint32_t)out[w] - (uint32_t)row[w]);
    }
  }
}

/* tfhe_bootstrap_FFT. REDsec call sites: BinOps_enc.cpp:185 (binarize_int, mu = 1/4096),
 * :191 (unbinarize_int, mu = 1/MULTIBIT_SPACE). */
void ro_bootstrap(const ro_ctx* c, int32_t* out, int32_t mu, const int32_t* in) {
  int32_t* u = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->p.k * c->p.N + 1));
  ro_bootstrap_wo_ks(c, u, mu, in);
  ro_keyswitch(c, out, u);
  free(u);
}

/* ------------------------------------------------------------------------------------------------
 * Gates. TFHE boolean-gates.cpp bootsNAND/OR/AND/NOR/XOR/XNOR/ANDNY/ANDYN/ORNY/ORYN/MUX.
 * REDsec call sites: BinOps_enc.cpp:49-52,104-113,153-166,205; IntOps_enc.cpp:63; IntFunc.cpp:962.
 * Constants are mirrored by lib/GPU/gates.cu:246-286 (mu = 1/8, fix in {+-1/8, +-1/4}).
 * ---------------------------------------------------------------------------------------------- */
static void gate_coeffs(ro_gate_op op, int32_t* bconst, int32_t* sa, int32_t* sb) {
  const int32_t e8 = ro_modswitch_to_torus32(1, 8), e4 = ro_modswitch_to_torus32(1, 4);
  switch (op) {
    case RO_NAND:  *bconst = e8;  *sa = -1; *sb = -1; break;
    case RO_OR:    *bconst = e8;  *sa = 1;  *sb = 1;  break;
    case RO_AND:   *bconst = -e8; *sa = 1;  *sb = 1;  break;
    case RO_NOR:   *bconst = -e8; *sa = -1; *sb = -1; break;
    case RO_XOR:   *bconst = e4;  *sa = 2;  *sb = 2;  break;
    case RO_XNOR:  *bconst = -e4; *sa = -2; *sb = -2; break;
    case RO_ANDNY: *bconst = -e8; *sa = -1; *sb = 1;  break;
    case RO_ANDYN: *bconst = -e8; *sa = 1;  *sb = -1; break;
    case RO_ORNY:  *bconst = e8;  *sa = -1; *sb = 1;  break;
    case RO_ORYN:  *bconst = e8;  *sa = 1;  *sb = -1; break;
    default:       *bconst = 0;   *sa = 0;  *sb = 0;  break;
  }
}

void ro_gate_precombine(ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb, int32_t n) {
  int32_t bc, sa, sb;
  gate_coeffs(op, &bc, &sa, &sb);
  for (int32_t i = 0; i <= n; ++i)
    out[i] = (int32_t)((uint32_t)sa * (uint32_t)ca[i] + (uint32_t)sb * (uint32_t)cb[i]);
  out[n] = (int32_t)((uint32_t)out[n] + (uint32_t)bc);
}

void ro_gate(const ro_ctx* c, ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb) {
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c->p.n + 1));
  ro_gate_precombine(op, tmp, ca, cb, c->p.n);
  ro_bootstrap(c, out, ro_modswitch_to_torus32(1, 8), tmp);
  free(tmp);
}

/* bootsMUX: u1 = woKS(AND(a,b)), u2 = woKS(ANDNY(a,c)), out = KS((0,1/8) + u1 + u2). */
void ro_mux(const ro_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc) {
  const int32_t n = c->p.n, Nk = c->p.k * c->p.N, mu = ro_modswitch_to_torus32(1, 8);
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
  int32_t* u1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Nk + 1));
  int32_t* u2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Nk + 1));
  ro_gate_precombine(RO_AND, tmp, a, b, n);
  ro_bootstrap_wo_ks(c, u1, mu, tmp);
  ro_gate_precombine(RO_ANDNY, tmp, a, cc, n);
  ro_bootstrap_wo_ks(c, u2, mu, tmp);
  for (int32_t i = 0; i <= Nk; ++i) u1[i] = (int32_t)((uint32_t)u1[i] + (uint32_t)u2[i]);
  u1[Nk] = (int32_t)((uint32_t)u1[Nk] + (uint32_t)mu);
  ro_keyswitch(c, out, u1);
  free(tmp); free(u1); free(u2);
}

/* ------------------------------------------------------------------------------------------------
 * Batches: the REDsec layer loops (BinFunc.cpp:1056-1071, IntFunc.cpp:871-887, BinFunc.cpp:896-921)
 * are OpenMP loops over independent ciphertexts; so is this.
 * ---------------------------------------------------------------------------------------------- */
void ro_bootstrap_lut_batch(const ro_ctx* c, int32_t* out, const int32_t* luts, size_t lut_count, const int32_t* in, size_t B) {
  const size_t W = (size_t)c->p.n + 1, N = (size_t)c->p.N;
#pragma omp parallel for schedule(dynamic)
  for (long b = 0; b < (long)B; ++b) ro_bootstrap_lut(c, out + (size_t)b * W, luts + ((size_t)b % lut_count) * N, in + (size_t)b * W);
}

int ro_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ro_set_threads(int t) {
#ifdef _OPENMP
  if (t > 0) omp_set_num_threads(t);
#else
  (void)t;
#endif
}

void ro_bootstrap_batch(const ro_ctx* c, int32_t* out, int32_t mu, const int32_t* in, size_t B) {
  const size_t W = (size_t)c->p.n + 1;
#pragma omp parallel for schedule(dynamic)
  for (long long b = 0; b < (long long)B; ++b) ro_bootstrap(c, out + (size_t)b * W, mu, in + (size_t)b * W);
}
void ro_gate_batch(const ro_ctx* c, ro_gate_op op, int32_t* out, const int32_t* ca, const int32_t* cb, size_t B) {
  const size_t W = (size_t)c->p.n + 1;
#pragma omp parallel for schedule(dynamic)
  for (long long b = 0; b < (long long)B; ++b)
    ro_gate(c, op, out + (size_t)b * W, ca + (size_t)b * W, cb + (size_t)b * W);
}
void ro_mux_batch(const ro_ctx* c, int32_t* out, const int32_t* a, const int32_t* b, const int32_t* cc, size_t B) {
  const size_t W = (size_t)c->p.n + 1;
#pragma omp parallel for schedule(dynamic)
  for (long long i = 0; i < (long long)B; ++i)
    ro_mux(c, out + (size_t)i * W, a + (size_t)i * W, b + (size_t)i * W, cc + (size_t)i * W);
}

/* ------------------------------------------------------------------------------------------------
 * Linear stage (no bootstrap)
 * ---------------------------------------------------------------------------------------------- */
void ro_linear_fc(int32_t* out, const int32_t* in, const uint8_t* sign, const uint8_t* zero,
                  int32_t K, int32_t M, int32_t W, int32_t zero_tap_b) {
#pragma omp parallel for schedule(static)
  for (int32_t m = 0; m < M; ++m) {
    int32_t* o = out + (size_t)m * W;
    memset(o, 0, sizeof(int32_t) * (size_t)W);
    for (int32_t kk = 0; kk < K; ++kk) {
      size_t fi = (size_t)kk * M + m;
      const int32_t* x = in + (size_t)kk * W;
      if (zero && zero[fi]) { o[W - 1] = (int32_t)((uint32_t)o[W - 1] + (uint32_t)zero_tap_b); continue; }
      if (sign[fi]) for (int32_t w = 0; w < W; ++w) o[w] = (int32_t)((uint32_t)o[w] + (uint32_t)x[w]);
      else          for (int32_t w = 0; w < W; ++w) o[w] = (int32_t)((uint32_t)o[w] - (uint32_t)x[w]);
    }
  }
}

void ro_add_bias(int32_t* x, const int32_t* bias_torus, int32_t count, int32_t depth, int32_t W) {
  for (int32_t i = 0; i < count; ++i) {
    int32_t* o = x + (size_t)i * W;
    o[W - 1] = (int32_t)((uint32_t)o[W - 1] + (uint32_t)bias_torus[i % depth]);
  }
}
