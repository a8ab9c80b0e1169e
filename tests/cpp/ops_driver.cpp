// ops_driver.cpp -- test program (tests/test_gpu_ops_wrappers.py): calls the per-ciphertext primitives of the layer
// mirror exactly as REDsec code would (namespace BinOps / IntOps, lib/BinOps_enc.h:8-49, lib/IntOps_enc.h:9-32;
// every bootstrapped one is a B = 1 round trip to the GPU) on fresh encryptions under a key it generates, decrypts
// with the secret key and prints PASS/FAIL per primitive. Exit code = number of failures.
#include <cstdio>
#include <initializer_list>
#include <string>
#include <vector>

#include "lib/BinOps_enc.h"
#include "lib/IntOps_enc.h"
#include "lib/Layer.h"

static TFheGateBootstrappingSecretKeySet* g_sk;
static TFheGateBootstrappingCloudKeySet* g_bk;
static int g_fail = 0;
// `ops_driver <dir>`: besides decrypting, leave the key (<dir>/secret.key, TFHE's file format) and, for the bootstrapped and
// copying primitives, the input and output CIPHERTEXTS (<dir>/vectors.ctxt, one export_gate_bootstrapping_ciphertext_toFile
// record each, in the order <dir>/vectors.txt lists: "name kind operands mu") so that the test harness can recompute every
// output with the CPU oracle and compare word for word (tests/test_gpu_ops_wrappers.py).
static FILE* g_vec = NULL;
static FILE* g_idx = NULL;
static void dump(const char* name, const char* kind, long mu, std::initializer_list<const LweSample*> in, const LweSample* out) {
  if (!g_vec) return;
  fprintf(g_idx, "%s %s %d %ld\n", name, kind, (int)in.size(), mu);
  for (const LweSample* s : in) export_gate_bootstrapping_ciphertext_toFile(g_vec, s, g_bk->params);
  export_gate_bootstrapping_ciphertext_toFile(g_vec, out, g_bk->params);
}

static void check(const char* what, bool ok) {
  printf("%s %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++g_fail;
}
static tMultiBit enc_bits(unsigned v, int bits) {   // little-endian, +-1/8 encoding
  tMultiBit m;
  m.size = (uint32_t)bits;
  m.ctxt = new_gate_bootstrapping_ciphertext_array(bits, g_bk->params);
  for (int i = 0; i < bits; ++i) bootsSymEncrypt(&m.ctxt[i], (v >> i) & 1, g_sk);
  return m;
}
static unsigned dec_bits(const tMultiBit& m, int bits) {
  unsigned v = 0;
  for (int i = 0; i < bits; ++i) v |= (unsigned)bootsSymDecrypt(&m.ctxt[i], g_sk) << i;
  return v;
}
static int dec_int(const LweSample* s, int msize) {
  const int v = modSwitchFromTorus32(lweSymDecrypt(s, g_sk->lwe_key, msize), msize);
  return v > msize / 2 ? v - msize : v;
}
static void enc_int(LweSample* s, int v) { lweSymEncrypt(s, modSwitchToTorus32(v, 4096), 1.0 / 32768, g_sk->lwe_key); }

int main(int argc, char** argv) {
  // redsec_params_small_v2, client/gen_secure_keyset.cpp:70-91
  LweParams* lp = new_LweParams(350, pow(2., -25), pow(2., -13));
  TLweParams* tp = new_TLweParams(1024, 1, pow(2., -30), pow(2., -13));
  TGswParams* gp = new_TGswParams(10, 3, tp);
  TFheGateBootstrappingParameterSet* params = new TFheGateBootstrappingParameterSet(9, 3, lp, gp);
  uint32_t seed[] = {1, 2, 3};
  tfhe_random_generator_setSeed(seed, 3);
  g_sk = new_random_gate_bootstrapping_secret_keyset(params);
  g_bk = const_cast<TFheGateBootstrappingCloudKeySet*>(&g_sk->cloud);
  const LweParams* io = params->in_out_params;
  if (argc > 1) {
    const std::string dir = argv[1];
    FILE* kf = fopen((dir + "/secret.key").c_str(), "wb");
    export_tfheGateBootstrappingSecretKeySet_toFile(kf, g_sk);
    fclose(kf);
    g_vec = fopen((dir + "/vectors.ctxt").c_str(), "wb");
    g_idx = fopen((dir + "/vectors.txt").c_str(), "w");
  }

  // a7: add (ripple carry, 5 bits - 3 gates), add_bit, inc
  for (unsigned a : {5u, 11u, 14u})
    for (unsigned b : {3u, 9u}) {
      tMultiBit x = enc_bits(a, 5), y = enc_bits(b, 5), r = enc_bits(0, 5);
      BinOps::add(&r, &x, &y, 5, g_bk);
      char what[64]; snprintf(what, sizeof what, "BinOps::add %u+%u", a, b);
      check(what, dec_bits(r, 5) == ((a + b) & 31));
    }
  {
    tMultiBit x = enc_bits(1, 1), y = enc_bits(1, 1), r;
    BinOps::add_bit(&r, &x.ctxt[0], &y.ctxt[0], g_bk);
    check("BinOps::add_bit 1+1", dec_bits(r, 2) == 2);
    tMultiBit a = enc_bits(11, 5), inc;
    BinOps::inc(&inc, &a, &x.ctxt[0], 5, g_bk);
    check("BinOps::inc 11+1", dec_bits(inc, 5) == 12);
  }
  // a8: multiply (XNOR with a plaintext bit), IntOps::invert
  {
    tMultiBit x = enc_bits(1, 1), r = enc_bits(0, 1);
    BinOps::multiply(&r.ctxt[0], &x.ctxt[0], 0, g_bk);
    check("BinOps::multiply by 0 = NOT", dec_bits(r, 1) == 0);
    dump("BinOps::multiply_by_0", "not", 0, {&x.ctxt[0]}, &r.ctxt[0]);
    BinOps::multiply(&r.ctxt[0], &x.ctxt[0], 1, g_bk);
    check("BinOps::multiply by 1 = COPY", dec_bits(r, 1) == 1);
    dump("BinOps::multiply_by_1", "copy", 0, {&x.ctxt[0]}, &r.ctxt[0]);
    tFixedPoint a = enc_bits(0b1010, 4), inv;
    uint8_t zero = 0, one = 1;
    IntOps::invert(&inv, &a, &zero, 4, g_bk);
    check("IntOps::invert b=0", dec_bits(inv, 4) == 0b0101);
    for (int i = 0; i < 4; ++i) dump("IntOps::invert_b0", "not", 0, {&a.ctxt[i]}, &inv.ctxt[i]);
    IntOps::invert(&inv, &a, &one, 4, g_bk);
    check("IntOps::invert b=1", dec_bits(inv, 4) == 0b1010);
    for (int i = 0; i < 4; ++i) dump("IntOps::invert_b1", "copy", 0, {&a.ctxt[i]}, &inv.ctxt[i]);
  }
  // a4: max = OR
  {
    tMultiBit x = enc_bits(0, 1), y = enc_bits(1, 1), r = enc_bits(0, 1);
    BinOps::max(&r.ctxt[0], &x.ctxt[0], &y.ctxt[0], g_bk);
    check("BinOps::max(0,1)", dec_bits(r, 1) == 1);
    dump("BinOps::max", "OR", 1 << 29, {&x.ctxt[0], &y.ctxt[0]}, &r.ctxt[0]);
    BinOps::max(&r.ctxt[0], &x.ctxt[0], &x.ctxt[0], g_bk);
    check("BinOps::max(0,0)", dec_bits(r, 1) == 0);
  }
  // a5: relu (AND of every bit with the top one), shift (copies with sign extension)
  {
    tMultiBit pos = enc_bits(0b10110, 5), neg = enc_bits(0b00110, 5), r = enc_bits(0, 5);
    BinOps::relu(&r, &pos, 5, g_bk);
    check("BinOps::relu top bit 1", (dec_bits(r, 5) & 15) == 0b0110);
    for (int i = 0; i < 4; ++i) dump("BinOps::relu", "AND", 1 << 29, {&pos.ctxt[i], &pos.ctxt[4]}, &r.ctxt[i]);   // bootsAND(bit i, top bit), lib/BinOps_enc.cpp:200-207
    IntOps::relu(&r, &neg, 5, g_bk);
    check("IntOps::relu top bit 0", (dec_bits(r, 5) & 15) == 0);
    for (int i = 0; i < 4; ++i) dump("IntOps::relu", "AND", 1 << 29, {&neg.ctxt[i], &neg.ctxt[4]}, &r.ctxt[i]);   // lib/IntOps_enc.cpp:58-65
    tMultiBit sh; sh.size = 0; sh.ctxt = NULL;
    BinOps::shift(&sh, &pos, 5, 2, g_bk);
    check("BinOps::shift by 2", dec_bits(sh, 5) == 0b11101);
  }
  // a2 / a3: binarize_int (mu = 1/4096), unbinarize_int (mu = 1/2048 = 1/MULTIBIT_SPACE)
  {
    LweSample* x = new_LweSample(io);
    LweSample* r = new_LweSample(io);
    bool ok = true, ok2 = true;
    for (int v : {-300, -40, 40, 300}) {
      enc_int(x, v);
      BinOps::binarize_int(r, x, 11, g_bk);
      ok = ok && dec_int(r, 4096) == (v >= 0 ? 1 : -1);
      dump("BinOps::binarize_int", "bootstrap", modSwitchToTorus32(1, 4096), {x}, r);        // lib/BinOps_enc.cpp:182-186
      BinOps::unbinarize_int(r, x, g_bk);
      ok2 = ok2 && dec_int(r, 2048) == (v >= 0 ? 1 : -1);
      dump("BinOps::unbinarize_int", "bootstrap", modSwitchToTorus32(1, 2048), {x}, r);      // :188-192, mu = 1/MULTIBIT_SPACE
    }
    check("BinOps::binarize_int", ok);
    check("BinOps::unbinarize_int (mu = 1/2048)", ok2);
  }
  // a9: word-wise integer ops
  {
    LweSample* a = new_LweSample(io); LweSample* b = new_LweSample(io); LweSample* r = new_LweSample(io);
    enc_int(a, 100); enc_int(b, -37);
    BinOps::add_int(r, a, b, g_bk);
    check("BinOps::add_int", dec_int(r, 4096) == 63);
    BinOps::add_int_inplace(r, a, g_bk);
    check("BinOps::add_int_inplace", dec_int(r, 4096) == 163);
    lweClear(r, io);
    const uint32_t mul = 7;
    BinOps::multiply_pc_ints(r, b, &mul, 8, 8, g_bk);
    check("BinOps::multiply_pc_ints", dec_int(r, 4096) == -259);
    lweClear(r, io);
    const uint16_t addend = 5;
    BinOps::add_pc_ints(r, a, &addend, 8, g_bk);                   // + 5 / MULTIBIT_SPACE = 10 / 4096
    check("BinOps::add_pc_ints", dec_int(r, 4096) == 110);
    tFixedPoint fa{a, 1}, fb{b, 1}, fr{r, 1}, fs;
    IntOps::add(&fr, &fa, &fb, 12, g_bk);
    check("IntOps::add", dec_int(fr.ctxt, 4096) == 63);
    IntOps::add_inplace(&fr, &fb, 12, g_bk);
    check("IntOps::add_inplace", dec_int(fr.ctxt, 4096) == 26);
    IntOps::subtract(&fs, &fa, &fb, 12, g_bk);
    check("IntOps::subtract", dec_int(fs.ctxt, 4096) == 137);
  }
  // a6: bootsMUX on gate-encoded bits
  {
    tMultiBit s1 = enc_bits(1, 1), s0 = enc_bits(0, 1), t = enc_bits(1, 1), e = enc_bits(0, 1), r = enc_bits(0, 1);
    bootsMUX(&r.ctxt[0], &s1.ctxt[0], &t.ctxt[0], &e.ctxt[0], g_bk);
    check("bootsMUX sel=1", dec_bits(r, 1) == 1);
    dump("bootsMUX_sel1", "MUX", 1 << 29, {&s1.ctxt[0], &t.ctxt[0], &e.ctxt[0]}, &r.ctxt[0]);
    bootsMUX(&r.ctxt[0], &s0.ctxt[0], &t.ctxt[0], &e.ctxt[0], g_bk);
    check("bootsMUX sel=0", dec_bits(r, 1) == 0);
    dump("bootsMUX_sel0", "MUX", 1 << 29, {&s0.ctxt[0], &t.ctxt[0], &e.ctxt[0]}, &r.ctxt[0]);
  }
  if (g_vec) { fclose(g_vec); fclose(g_idx); }
  printf("failures: %d\n", g_fail);
  return g_fail;
}
