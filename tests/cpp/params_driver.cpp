// params_driver.cpp -- test program (tests/test_gpu_general.py): the parameter sets client/gen_secure_keyset.cpp defines
// beside the one it ships (redsec_params_small :47-68, redsec_params_medium :28-45, redsec_params_large :9-26), built with
// the TFHE constructors exactly as the client builds them, through the shim's keygen, TFHE-format key files, the
// per-ciphertext bootstrapped calls and one BinLayer. The LWE dimension of the two big rings is cut down (argv) so that
// the host-side key generation stays in seconds; ring degree, gadget and keyswitch shape are the reference's.
// Prints PASS/FAIL lines; exit code = number of failures.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lib/BinLayer.h"
#include "lib/BinOps_enc.h"

static int g_fail = 0;
static void check(const char* set, const char* what, bool ok) { printf("%s %s: %s\n", ok ? "PASS" : "FAIL", set, what); if (!ok) ++g_fail; }

static TFheGateBootstrappingParameterSet* make(int n, int N, double ks_sd, double bk_sd, double sd) {
  LweParams* lp = new_LweParams(n, ks_sd, sd);
  TLweParams* tp = new_TLweParams(N, 1, bk_sd, sd);
  TGswParams* gp = new_TGswParams(3, 10, tp);
  return new TFheGateBootstrappingParameterSet(18, 1, lp, gp);
}

static void run(const char* set, TFheGateBootstrappingParameterSet* params) {
  uint32_t seed[] = {3, 1, 4};
  tfhe_random_generator_setSeed(seed, 3);
  TFheGateBootstrappingSecretKeySet* sk = new_random_gate_bootstrapping_secret_keyset(params);
  // keys through files, as net.cpp / main.cpp read them (nets/mnist/sign1024x1/net.cpp:53-55)
  FILE* f = tmpfile();
  export_tfheGateBootstrappingCloudKeySet_toFile(f, &sk->cloud);
  rewind(f);
  TFheGateBootstrappingCloudKeySet* bk = new_tfheGateBootstrappingCloudKeySet_fromFile(f);
  fclose(f);
  const int N = bk->params->tgsw_params->tlwe_params->N, n = bk->params->in_out_params->n;
  check(set, "cloud key file round trip keeps the parameters",
        N == params->tgsw_params->tlwe_params->N && n == params->in_out_params->n && bk->params->tgsw_params->Bgbit == 10 &&
        bk->params->ks_t == 18 && bk->params->ks_basebit == 1);

  // gates in the +-1/8 encoding
  bool ok = true;
  LweSample* x = new_gate_bootstrapping_ciphertext_array(4, bk->params);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      bootsSymEncrypt(&x[0], a, sk); bootsSymEncrypt(&x[1], b, sk);
      bootsAND(&x[2], &x[0], &x[1], bk);
      bootsXOR(&x[3], &x[0], &x[1], bk);
      ok = ok && bootsSymDecrypt(&x[2], sk) == (a & b) && bootsSymDecrypt(&x[3], sk) == (a ^ b);
      bootsMUX(&x[2], &x[0], &x[1], &x[3], bk);
      ok = ok && bootsSymDecrypt(&x[2], sk) == (a ? b : (a ^ b));
    }
  check(set, "bootsAND / bootsXOR / bootsMUX truth tables", ok);

  // REDsec's sign bootstrap over the 4096-level message space (lib/BinOps_enc.cpp:182-186)
  ok = true;
  const Torus32 mu = modSwitchToTorus32(1, 4096);
  for (int m : {-1500, -300, -64, 64, 700, 1900}) {
    lweSymEncrypt(&x[0], modSwitchToTorus32(m, 4096), params->in_out_params->alpha_min, sk->lwe_key);
    BinOps::binarize_int(&x[1], &x[0], 1, bk);
    // the sign of the phase, not the nearest 1/4096: with t * basebit = 18 bits the keyswitch's own rounding noise on
    // N = 4096 / 8192 (sigma ~ 2^-14) is of the order of the 2^-13 rounding margin -- a property of the parameter set
    const Torus32 ph = lwePhase(&x[1], sk->lwe_key);
    ok = ok && (ph > 0) == (m > 0);
  }
  check(set, "BinOps::binarize_int = sign over Z/4096", ok);

  // one BinLayer(E_FC, SIGN): 24 inputs, 6 neurons, clear margins
  const int K = 24, M = 6;
  std::vector<int> w((size_t)K * M), bits(K);
  std::vector<int32_t> bias(M);
  unsigned r = 99;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r >> 8; };
  for (auto& v : w) v = (rnd() % 3 == 0) ? -1 : 1;
  for (auto& b : bits) b = (rnd() & 1) ? 1 : -1;
  for (int m = 0; m < M; ++m) {
    int pre = 0;
    for (int k = 0; k < K; ++k) pre += w[(size_t)k * M + m] * bits[k];
    bias[m] = (pre >= 0 ? 40 : -40) - pre / 2;         // pushes every neuron at least 28 steps from zero
  }
  f = tmpfile();
  fputc(2, f);
  std::vector<unsigned char> pack((w.size() * 2 + 7) / 8, 0);
  for (size_t i = 0; i < w.size(); ++i) if (w[i] > 0) pack[(2 * i) >> 3] |= 0x80 >> ((2 * i) & 7);
  fwrite(pack.data(), 1, pack.size(), f);
  fputc(4, f); fwrite(bias.data(), 4, bias.size(), f);
  rewind(f);
  tNetParams np; memset(&np, 0, sizeof np);
  np.conv.window.h = np.conv.window.w = 1; np.conv.stride.h = np.conv.stride.w = 1; np.conv.same_pad = true; np.e_bias = E_BNORM; np.version = 2;
  tDimensions d; memset(&d, 0, sizeof d);
  d.hw.h = d.hw.w = 1; d.in_dep = K; d.in_bits = 1; d.out_bits = SINGLE_BIT; d.filter_bits = SINGLE_BIT; d.bias_bits = SINGLE_BIT; d.up_bound = 1; d.scale = 1;
  BinLayer layer(E_FC, M, E_NO_POOL, E_ACTIVATION_SIGN, &np, bk);
  layer.prep(f, &d);
  fclose(f);
  tBit* in = new_gate_bootstrapping_ciphertext_array(K, bk->params);
  for (int k = 0; k < K; ++k) lweSymEncrypt(&in[k], bits[k] * mu, params->in_out_params->alpha_min, sk->lwe_key);
  tBit* out = (tBit*)layer.execute(in);
  ok = true;
  for (int m = 0; m < M; ++m) {
    int pre = bias[m];
    for (int k = 0; k < K; ++k) pre += w[(size_t)k * M + m] * bits[k];
    const Torus32 ph = lwePhase(&out[m], sk->lwe_key);
    ok = ok && (ph > 0) == (pre > 0);
  }
  check(set, "BinLayer(E_FC, SIGN) decrypts to sign(w.x + bias)", ok);
  delete_gate_bootstrapping_cloud_keyset(bk);
  delete_gate_bootstrapping_secret_keyset(sk);
}

int main(int argc, char** argv) {
  const int n_medium = argc > 1 ? atoi(argv[1]) : 24, n_large = argc > 2 ? atoi(argv[2]) : 12;
  run("redsec_params_small (n=500 N=1024)", make(500, 1024, pow(2., -25), pow(2., -36), pow(2., -11)));
  char name[96];
  snprintf(name, sizeof name, "redsec_params_medium (n=%d of 3072, N=4096)", n_medium);
  run(name, make(n_medium, 4096, pow(2., -40), pow(2., -45), pow(2., -45)));
  snprintf(name, sizeof name, "redsec_params_large (n=%d of 6144, N=8192)", n_large);
  run(name, make(n_large, 8192, pow(2., -41), pow(2., -46), pow(2., -48)));
  printf("failures: %d\n", g_fail);
  return g_fail;
}
