// tfhe/tfhe.h -- TFHE-compatible header shim over the MI355X backend (libredsec_hip.so).
//
// REDsec's encrypted flavour includes <tfhe/tfhe.h> (lib/Layer.h:19, lib/BinOps_enc.h:6,
// client/*.cpp:1) and links libtfhe-spqlios-fma, which is neither vendored in the reference nor
// installed here. This header restates the part of TFHE v1.1's public API that REDsec touches --
// the 9 types and ~40 functions listed in SURVEY.md section 8b -- with the field names REDsec
// dereferences (bk->params->in_out_params, bk->bkFFT, bk->bk->in_out_params, key->lwe_key,
// key->cloud, sample->a / ->b), so that lib/*_enc.*, nets/*/*/{net,main}.cpp and client/*.cpp
// compile unmodified with -DENCRYPTED -I<this dir>. The implementation (tfhe_shim.cpp) keeps
// LweSample in host memory as TFHE does, runs the word-wise LWE ops there (that is what the
// reference's per-ciphertext API means), and sends every bootstrap to the GPU through the C ABI
// (rs_bootstrap / rs_gate / rs_mux with B = 1). The layer classes (layers.cpp) bypass the
// per-ciphertext route and launch one batch per stage.
#ifndef REDSEC_TFHE_SHIM_H
#define REDSEC_TFHE_SHIM_H

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int32_t Torus32;

struct LweParams {
  const int32_t n;
  const double alpha_min;
  const double alpha_max;
  LweParams(int32_t n_, double amin, double amax) : n(n_), alpha_min(amin), alpha_max(amax) {}
};

struct LweSample {
  Torus32* a;
  Torus32 b;
  double current_variance;
};

struct LweKey {
  const LweParams* params;
  int32_t* key;
};

struct TLweParams {
  const int32_t N;
  const int32_t k;
  const double alpha_min;
  const double alpha_max;
  const LweParams extracted_lweparams;
  TLweParams(int32_t N_, int32_t k_, double amin, double amax)
      : N(N_), k(k_), alpha_min(amin), alpha_max(amax), extracted_lweparams(N_ * k_, amin, amax) {}
};

struct TGswParams {
  const int32_t l;
  const int32_t Bgbit;
  const int32_t Bg;
  const int32_t halfBg;
  const uint32_t maskMod;
  const TLweParams* tlwe_params;
  const int32_t kpl;
  TGswParams(int32_t l_, int32_t Bgbit_, const TLweParams* tp)
      : l(l_), Bgbit(Bgbit_), Bg(1 << Bgbit_), halfBg((1 << Bgbit_) / 2), maskMod((1u << Bgbit_) - 1u), tlwe_params(tp),
        kpl((tp->k + 1) * l_) {}
};

struct TGswKey {
  const TGswParams* params;
  int32_t* key;  // k*N binary coefficients of the TRLWE key
};

struct TFheGateBootstrappingParameterSet {
  const int32_t ks_t;
  const int32_t ks_basebit;
  const LweParams* const in_out_params;
  const TGswParams* const tgsw_params;
  TFheGateBootstrappingParameterSet(int32_t t, int32_t basebit, const LweParams* lp, const TGswParams* gp)
      : ks_t(t), ks_basebit(basebit), in_out_params(lp), tgsw_params(gp) {}
};

// Evaluation key in the layouts of include/redsec_hip.h.
struct LweBootstrappingKey {
  const LweParams* in_out_params;
  const TGswParams* bk_params;
  int32_t* bk_words;   // [n][(k+1)l][k+1][N]
  int32_t* ksk_words;  // [k*N][t][base][n+1]
};

struct rs_ctx;
// Device-side key ("bkFFT" in TFHE): the GPU context with the transformed key loaded.
struct TFheGateBootstrappingParameterSet;
struct LweBootstrappingKeyFFT {
  const LweParams* in_out_params;
  rs_ctx* ctx;                                      // created on first bootstrap (client tools never touch the GPU)
  rs_ctx** fleet;                                   // ctx plus one more context per further device of REDSEC_DEVICES
  int fleet_size;
  const TFheGateBootstrappingParameterSet* params;  // what the lazy creation needs
  const LweBootstrappingKey* src;
};

struct TFheGateBootstrappingCloudKeySet {
  const TFheGateBootstrappingParameterSet* const params;
  const LweBootstrappingKey* const bk;
  const LweBootstrappingKeyFFT* const bkFFT;
  TFheGateBootstrappingCloudKeySet(const TFheGateBootstrappingParameterSet* p, const LweBootstrappingKey* k, const LweBootstrappingKeyFFT* f)
      : params(p), bk(k), bkFFT(f) {}
};

struct TFheGateBootstrappingSecretKeySet {
  const TFheGateBootstrappingParameterSet* params;
  const LweKey* lwe_key;
  const TGswKey* tgsw_key;
  const TFheGateBootstrappingCloudKeySet cloud;
  TFheGateBootstrappingSecretKeySet(const TFheGateBootstrappingParameterSet* p, const LweBootstrappingKey* bk, const LweBootstrappingKeyFFT* bkFFT,
                                    const LweKey* lk, const TGswKey* gk)
      : params(p), lwe_key(lk), tgsw_key(gk), cloud(p, bk, bkFFT) {}
};

// ---- parameters ----
LweParams* new_LweParams(int32_t n, double alpha_min, double alpha_max);
TLweParams* new_TLweParams(int32_t N, int32_t k, double alpha_min, double alpha_max);
TGswParams* new_TGswParams(int32_t l, int32_t Bgbit, const TLweParams* tlwe_params);
TFheGateBootstrappingParameterSet* new_default_gate_bootstrapping_parameters(int32_t minimum_lambda);

// ---- torus ----
Torus32 modSwitchToTorus32(int32_t mu, int32_t Msize);
int32_t modSwitchFromTorus32(Torus32 phase, int32_t Msize);
Torus32 dtot32(double d);

// ---- samples ----
LweSample* new_LweSample(const LweParams* params);
LweSample* new_LweSample_array(int32_t nbelts, const LweParams* params);
void delete_LweSample(LweSample* s);
void delete_LweSample_array(int32_t nbelts, LweSample* s);
LweSample* new_gate_bootstrapping_ciphertext(const TFheGateBootstrappingParameterSet* params);
LweSample* new_gate_bootstrapping_ciphertext_array(int32_t nbelems, const TFheGateBootstrappingParameterSet* params);
void delete_gate_bootstrapping_ciphertext(LweSample* sample);
void delete_gate_bootstrapping_ciphertext_array(int32_t nbelems, LweSample* samples);

void lweClear(LweSample* result, const LweParams* params);
void lweCopy(LweSample* result, const LweSample* sample, const LweParams* params);
void lweNegate(LweSample* result, const LweSample* sample, const LweParams* params);
void lweNoiselessTrivial(LweSample* result, Torus32 mu, const LweParams* params);
void lweAddTo(LweSample* result, const LweSample* sample, const LweParams* params);
void lweSubTo(LweSample* result, const LweSample* sample, const LweParams* params);
void lweAddMulTo(LweSample* result, int32_t p, const LweSample* sample, const LweParams* params);
void lweSubMulTo(LweSample* result, int32_t p, const LweSample* sample, const LweParams* params);
void lweSymEncrypt(LweSample* result, Torus32 message, double alpha, const LweKey* key);
Torus32 lwePhase(const LweSample* sample, const LweKey* key);
Torus32 lweSymDecrypt(const LweSample* sample, const LweKey* key, const int32_t Msize);

// ---- keys ----
void tfhe_random_generator_setSeed(uint32_t* values, int32_t size);
TFheGateBootstrappingSecretKeySet* new_random_gate_bootstrapping_secret_keyset(const TFheGateBootstrappingParameterSet* params);
void delete_gate_bootstrapping_secret_keyset(TFheGateBootstrappingSecretKeySet* keyset);
void delete_gate_bootstrapping_cloud_keyset(TFheGateBootstrappingCloudKeySet* keyset);
void delete_gate_bootstrapping_parameters(TFheGateBootstrappingParameterSet* params);

// ---- bootstrapped operations (GPU) ----
void tfhe_bootstrap_FFT(LweSample* result, const LweBootstrappingKeyFFT* bk, Torus32 mu, const LweSample* x);
void bootsSymEncrypt(LweSample* result, int32_t message, const TFheGateBootstrappingSecretKeySet* key);
int32_t bootsSymDecrypt(const LweSample* sample, const TFheGateBootstrappingSecretKeySet* key);
void bootsCONSTANT(LweSample* result, int32_t value, const TFheGateBootstrappingCloudKeySet* bk);
void bootsNOT(LweSample* result, const LweSample* ca, const TFheGateBootstrappingCloudKeySet* bk);
void bootsCOPY(LweSample* result, const LweSample* ca, const TFheGateBootstrappingCloudKeySet* bk);
void bootsNAND(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsOR(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsAND(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsXOR(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsXNOR(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsNOR(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsANDNY(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsANDYN(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsORNY(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsORYN(LweSample* result, const LweSample* ca, const LweSample* cb, const TFheGateBootstrappingCloudKeySet* bk);
void bootsMUX(LweSample* result, const LweSample* a, const LweSample* b, const LweSample* c, const TFheGateBootstrappingCloudKeySet* bk);

// ---- extensions used by the batched layer code (not part of TFHE) ----
// GPU context behind an evaluation key, and (un)packing between LweSample and the ABI's W-word rows.
rs_ctx* redsec_ctx_of(const TFheGateBootstrappingCloudKeySet* bk);
// REDSEC_DEVICES=0,1,...: one context (full key replica) per listed device, the layer code shards every bootstrapped
// stage across them (the reference's shape: one host thread per GPU over enc_segs[NUM_GPUS], lib/GPU/BinFunc_gpu.cu:119-137).
// Unset: the single device REDSEC_DEVICE (default 0). redsec_fleet_of returns the contexts, *count how many.
rs_ctx** redsec_fleet_of(const TFheGateBootstrappingCloudKeySet* bk, int* count);
// REDSEC_LAZY_HOST=1 leaves the host arrays of intermediate layers unfilled (layers.cpp, publish): fills one on request.
void redsec_materialize(void* host_array);
// resident slabs and cached device blocks of a context, released before the context itself (called by the keyset deleters)
void redsec_pool_release(rs_ctx* ctx);
double redsec_take_setup_seconds(void);   /* seconds of device-context creation + key upload since the last call (REDSEC_TRACE) */
void redsec_pack(int32_t* words, const LweSample* s, int32_t n);
void redsec_unpack(LweSample* s, const int32_t* words, int32_t n);

#include "tfhe_io.h"

#endif
