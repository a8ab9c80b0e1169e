// func_driver.cpp -- test program (tests/test_gpu_ops_wrappers.py): the reference's per-stage classes BinFunc::* / IntFunc::*
// (lib/BinFunc.h:37-173, lib/IntFunc.h:27-140) against the layer classes built from them (lib/BinLayer.cpp:150-241,
// lib/IntLayer.cpp:153-235): the same weights, key and inputs through
//   BinLayer(E_FC, SIGN)                     vs  BinFunc::Convolution -> BinFunc::Quantize::execute
//   IntLayer(E_NO_CONV, E_SUMPOOL, SIGN)     vs  IntFunc::SumPooling -> IntFunc::Quantize::execute
// must give the SAME ciphertext words (a bias folded into the linear kernel or added after it is the same wrap-around sum),
// and BinFunc::MaxPooling / IntFunc::Quantize::{add_bias, relu_shift} on their own must decrypt to the plaintext function.
// Prints PASS/FAIL lines; exit code = number of failures.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lib/BinFunc.h"
#include "lib/BinLayer.h"
#include "lib/IntFunc.h"
#include "lib/IntLayer.h"

static TFheGateBootstrappingSecretKeySet* g_sk;
static TFheGateBootstrappingCloudKeySet* g_bk;
static int g_fail = 0;
static void check(const char* what, bool ok) { printf("%s %s\n", ok ? "PASS" : "FAIL", what); if (!ok) ++g_fail; }
static int dec_int(const LweSample* s, int msize) {
  const int v = modSwitchFromTorus32(lweSymDecrypt(s, g_sk->lwe_key, msize), msize);
  return v > msize / 2 ? v - msize : v;
}
static bool same(const LweSample* a, const LweSample* b, int n) { return a->b == b->b && memcmp(a->a, b->a, 4 * (size_t)n) == 0; }

// weight file records (lib/BinOps_enc.cpp:247-305): tag 2 + 2 bits per ternary weight (sign, is-zero), MSB first; tag 4 + int32[]
static void put_ternary(FILE* f, const std::vector<int>& w) {   // w in {-1, 0, +1}
  fputc(2, f);
  std::vector<unsigned char> pack((w.size() * 2 + 7) / 8, 0);
  for (size_t i = 0; i < w.size(); ++i) {
    if (w[i] > 0) pack[(2 * i) >> 3] |= 0x80 >> ((2 * i) & 7);
    if (w[i] == 0) pack[(2 * i + 1) >> 3] |= 0x80 >> ((2 * i + 1) & 7);
  }
  fwrite(pack.data(), 1, pack.size(), f);
}
static void put_ints(FILE* f, const std::vector<int32_t>& v) { fputc(4, f); fwrite(v.data(), 4, v.size(), f); }
static tDimensions dims(int h, int w, int dep) {
  tDimensions d; memset(&d, 0, sizeof d);
  d.hw.h = (int16_t)h; d.hw.w = (int16_t)w; d.in_dep = (uint32_t)dep; d.in_bits = 1; d.out_bits = SINGLE_BIT; d.filter_bits = SINGLE_BIT;
  d.bias_bits = SINGLE_BIT; d.up_bound = 1; d.scale = 1;
  return d;
}

int main() {
  LweParams* lp = new_LweParams(350, pow(2., -25), pow(2., -13));
  TLweParams* tp = new_TLweParams(1024, 1, pow(2., -30), pow(2., -13));
  TGswParams* gp = new_TGswParams(10, 3, tp);
  TFheGateBootstrappingParameterSet* params = new TFheGateBootstrappingParameterSet(9, 3, lp, gp);
  uint32_t seed[] = {4, 5, 6};
  tfhe_random_generator_setSeed(seed, 3);
  g_sk = new_random_gate_bootstrapping_secret_keyset(params);
  g_bk = const_cast<TFheGateBootstrappingCloudKeySet*>(&g_sk->cloud);
  const int n = 350;
  const Torus32 u = modSwitchToTorus32(1, 4096);

  // ---- BinLayer(E_FC, SIGN) vs Convolution + Quantize ----
  const int K = 48, M = 12;
  std::vector<int> w((size_t)K * M);
  std::vector<int32_t> bias(M);
  unsigned r = 12345;
  auto rnd = [&]() { r = r * 1664525u + 1013904223u; return r >> 8; };
  for (auto& x : w) { const unsigned t = rnd() % 10; x = t < 3 ? 0 : (t < 7 ? 1 : -1); }
  for (auto& b : bias) b = (int32_t)(rnd() % 17) - 8;
  std::vector<int> bits(K);
  for (auto& b : bits) b = (rnd() & 1) ? 1 : -1;
  FILE* f = tmpfile();
  put_ternary(f, w); put_ints(f, bias);          // the layer's records
  put_ternary(f, w); put_ints(f, bias);          // and again for the stage-by-stage instance
  rewind(f);
  auto enc_bits = [&]() {
    tBit* x = new_gate_bootstrapping_ciphertext_array(K, params);
    uint32_t s2[] = {7, 7, 7};
    tfhe_random_generator_setSeed(s2, 3);        // the same fresh encryptions both times
    for (int i = 0; i < K; ++i) lweSymEncrypt(&x[i], bits[i] * u, 1.0 / 32768, g_sk->lwe_key);
    return x;
  };
  tNetParams np; memset(&np, 0, sizeof np);
  np.conv.window.h = np.conv.window.w = 1; np.conv.stride.h = np.conv.stride.w = 1; np.conv.same_pad = true; np.e_bias = E_BNORM; np.version = 2;
  np.pool.window.h = np.pool.window.w = 2; np.pool.stride.h = np.pool.stride.w = 2;
  tDimensions d1 = dims(1, 1, K);
  BinLayer layer(E_FC, M, E_NO_POOL, E_ACTIVATION_SIGN, &np, g_bk);
  layer.prep(f, &d1);
  tBit* out_layer = (tBit*)layer.execute(enc_bits());
  tDimensions d2 = dims(1, 1, K);
  BinFunc::Convolution conv(M, &np.conv);
  tQParams q1; q1.shift_bits = 1;
  BinFunc::Quantize quant(&q1);
  std::vector<tMultiBit> pb(M);
  conv.prep(f, &d2, g_bk);
  quant.prep(f, &d2, pb.data(), NULL, g_bk);
  tBit* out_func = quant.execute(conv.execute(enc_bits()), pb.data());
  bool eq = true, right = true;
  for (int m = 0; m < M; ++m) {
    eq = eq && same(&out_layer[m], &out_func[m], n);
    int pre = bias[m];
    for (int k = 0; k < K; ++k) pre += w[(size_t)k * M + m] * bits[k];
    if (pre >= 8 || pre <= -8) right = right && dec_int(&out_func[m], 4096) == (pre >= 0 ? 1 : -1);
  }
  check("BinFunc::Convolution + Quantize::execute == BinLayer(E_FC, SIGN), word for word", eq);
  check("  ... and decrypts to sign(w.x + bias) on clear margins", right);
  check("Quantize::prep hands out the bias record", dec_int(&pb[3].ctxt[0], 4096) == bias[3]);
  fclose(f);

  // ---- IntLayer(NO_CONV, SUMPOOL, SIGN) vs IntFunc::SumPooling + IntFunc::Quantize ----
  const int H = 4;
  std::vector<int> px(H * H);
  for (auto& p : px) p = (int)(rnd() % 41) - 20;
  std::vector<int32_t> b0(1, 3);
  f = tmpfile();
  put_ints(f, b0); put_ints(f, b0);
  rewind(f);
  auto enc_px = [&]() {
    tMultiBit* x = new tMultiBit[H * H];
    uint32_t s2[] = {9, 9, 9};
    tfhe_random_generator_setSeed(s2, 3);
    for (int i = 0; i < H * H; ++i) {
      x[i].size = 1; x[i].ctxt = new_gate_bootstrapping_ciphertext_array(1, params);
      lweSymEncrypt(&x[i].ctxt[0], px[i] * u, 1.0 / 32768, g_sk->lwe_key);
    }
    return x;
  };
  tDimensions d3 = dims(H, H, 1);
  IntLayer il(E_NO_CONV, 1, E_SUMPOOL, E_ACTIVATION_SIGN, &np, g_bk);
  il.prep(f, &d3);
  tBit* o1 = (tBit*)il.execute(enc_px());
  tDimensions d4 = dims(H, H, 1);
  IntFunc::SumPooling sp(&np.pool);
  IntFunc::Quantize iq(&q1);
  std::vector<tMultiBit> pb0(1);
  sp.prep(&d4, g_bk);
  iq.prep(f, &d4, pb0.data(), NULL, g_bk);
  tBit* o2 = iq.execute(sp.execute(enc_px()), pb0.data());
  eq = true;
  for (int i = 0; i < 4; ++i) eq = eq && same(&o1[i], &o2[i], n);
  check("IntFunc::SumPooling + Quantize::execute == IntLayer(NO_CONV, SUMPOOL, SIGN), word for word", eq);
  fclose(f);

  // ---- BinFunc::MaxPooling on its own: 4x4x2 sign bits -> 2x2x2, OR of every 2x2 window ----
  {
    const int C = 2;
    std::vector<int> sb(4 * 4 * C);
    for (auto& b : sb) b = (rnd() % 3 == 0) ? 1 : -1;
    sb[0] = sb[1 * C] = sb[4 * C] = sb[5 * C] = -1;          // one all-false window (channel 0, top left)
    tBit* x = new_gate_bootstrapping_ciphertext_array(4 * 4 * C, params);
    for (size_t i = 0; i < sb.size(); ++i) lweSymEncrypt(&x[i], sb[i] * u, 1.0 / 32768, g_sk->lwe_key);
    tDimensions d5 = dims(4, 4, C);
    BinFunc::MaxPooling mp(&np.pool);
    mp.prep(&d5, g_bk);
    tBit* y = mp.execute(x);
    bool ok = d5.hw.h == 2 && d5.hw.w == 2;
    for (int oh = 0; oh < 2; ++oh) for (int ow = 0; ow < 2; ++ow) for (int c = 0; c < C; ++c) {
      int any = -1;
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (sb[((2 * oh + a) * 4 + 2 * ow + b) * C + c] > 0) any = 1;
      const int got = dec_int(&y[(oh * 2 + ow) * C + c], 4096);
      if (got != any) printf("  maxpool (%d,%d,%d): got %d want %d\n", oh, ow, c, got, any);
      ok = ok && got == any;
    }
    check("BinFunc::MaxPooling::execute = OR over each window", ok);
  }

  // ---- IntFunc::Quantize add_bias and relu_shift on their own ----
  {
    const int D = 6;
    // gentle slopes: the mod-switch moves the input by sigma ~ 8 integer steps (SURVEY.md hard part 7), i.e. by
    // slope * 8 / 64 output levels; the first and last neuron sit far inside the clamped regions
    std::vector<int32_t> bb = {-3000, 400, 500, 96, 640, 4000}, slope = {4, 4, 2, 4, 1, 3};
    std::vector<int> in = {10, 20, -30, 7, -30, 33};
    f = tmpfile();
    put_ints(f, bb);
    put_ints(f, bb); put_ints(f, slope);
    rewind(f);
    auto enc = [&]() {
      tMultiBit* x = new tMultiBit[D];
      for (int i = 0; i < D; ++i) { x[i].size = 1; x[i].ctxt = new_gate_bootstrapping_ciphertext_array(1, params); lweSymEncrypt(&x[i].ctxt[0], in[i] * u, 1.0 / 65536, g_sk->lwe_key); }
      return x;
    };
    tQParams q0; q0.shift_bits = 0;
    IntFunc::Quantize ab(&q0);
    tDimensions d6 = dims(1, 1, D);
    std::vector<tMultiBit> pbb(D);
    ab.prep(f, &d6, pbb.data(), NULL, g_bk);
    tFixedPoint* y = ab.add_bias(enc(), pbb.data());
    bool ok = true;
    for (int i = 0; i < D; ++i) {   // the message space is Z/4096: the far-out biases of the ReLU case wrap around it
      const int got = dec_int(&y[i].ctxt[0], 4096), want = in[i] + bb[i];
      if (((got - want) & 4095) != 0) printf("  add_bias %d: got %d want %d (mod 4096)\n", i, got, want);
      ok = ok && ((got - want) & 4095) == 0;
    }
    check("IntFunc::Quantize::add_bias", ok);
    tQParams q4; q4.shift_bits = 4;
    IntFunc::Quantize rq(&q4);
    tDimensions d7 = dims(1, 1, D);
    d7.scale = 4;                                      // slope_bits = 8 + 2 - 4 = 6, as in relu1024x1's first hidden layer
    std::vector<uint32_t> ps(D);
    rq.prep(f, &d7, pbb.data(), ps.data(), g_bk);
    tFixedPoint* z = rq.relu_shift(enc(), pbb.data(), ps.data());
    ok = ps[2] == (uint32_t)slope[2];                  // prep hands the raw slope record to the caller (get_intfilters_ptxt)
    if (!ok) printf("  relu_shift: p_slope[2] = %u, record holds %d\n", ps[2], slope[2]);
    for (int i = 0; i < D; ++i) {
      const long x = (long)slope[i] * in[i] + bb[i];
      const int want = x < 0 ? 0 : (int)((x >> 6) > 15 ? 15 : (x >> 6));
      const int got = dec_int(&z[i].ctxt[0], 16384);
      const bool good = i == 0 || i == D - 1 ? got == want : (got >= want - 2 && got <= want + 2);
      if (!good) printf("  relu_shift %d: got %d want %d\n", i, got, want);
      ok = ok && good;
    }
    check("IntFunc::Quantize::relu_shift = clamp((slope x + bias) >> 6, 0, 15) in units of 1/16384", ok);
    fclose(f);
  }
  // ---- two networks in one process: the value unit travels inside each chain's own tDimensions object ----
  {
    // net R: IntLayer(E_FC 6 -> 6, RELU shift_bits 4) -> IntLayer(E_FC 6 -> 3, no activation); net S: BinLayer(E_FC 6 -> 4, SIGN).
    // R's hidden activations are pinned to 0 / 15 (far inside the clamped regions), so its logits are exact integers.
    const int D = 6, C = 3, MS = 4;
    std::vector<int> wid((size_t)D * D, 0), w1((size_t)D * C), ws((size_t)D * MS);
    for (int i = 0; i < D; ++i) wid[(size_t)i * D + i] = 1;
    for (auto& x : w1) { const unsigned t = rnd() % 3; x = t == 0 ? 0 : (t == 1 ? 1 : -1); }
    for (auto& x : ws) x = (rnd() & 1) ? 1 : -1;
    std::vector<int32_t> b0 = {-3000, 4000, -3000, 4000, 4000, -3000}, sl(D, 4), b1 = {5, -7, 11}, bs = {40, -40, 40, -40};   // sign net: margins far above the mod-switch noise
    std::vector<int> in = {10, 20, -30, 7, -30, 33};
    auto make_r = [&](FILE* g) { put_ternary(g, wid); put_ints(g, b0); put_ints(g, sl); put_ternary(g, w1); put_ints(g, b1); rewind(g); };
    FILE* fr = tmpfile(); make_r(fr);
    FILE* fs = tmpfile(); put_ternary(fs, ws); put_ints(fs, bs); rewind(fs);
    tNetParams nq = np; nq.quant.shift_bits = 4;
    IntLayer r0(E_FC, D, E_NO_POOL, E_ACTIVATION_RELU, &nq, g_bk), r1(E_FC, C, E_NO_POOL, E_ACTIVATION_NONE, &np, g_bk);
    BinLayer s0(E_FC, MS, E_NO_POOL, E_ACTIVATION_SIGN, &np, g_bk);
    tDimensions dr = dims(1, 1, D), ds = dims(1, 1, D);
    dr.scale = 4;
    tDimensions* pr = &dr; tDimensions* ps = &ds;
    pr = (tDimensions*)r0.prep(fr, pr);           // interleaved on purpose: R0, S0, R1
    ps = (tDimensions*)s0.prep(fs, ps);
    pr = (tDimensions*)r1.prep(fr, pr);
    check("the dims object keeps the reference's bookkeeping: (4, 8, 15 marked) behind the integer ReLU, (1, 1, 1/2) behind the sign",
          r0.out_dim.in_bits == 4 && r0.out_dim.up_bound == 8 && r0.out_dim.scale < 15.0f && r0.out_dim.scale > 14.9999f &&
          s0.out_dim.in_bits == 1 && s0.out_dim.up_bound == 1 && s0.out_dim.scale == 0.5f);
    auto enc_in = [&](unsigned seed) {
      tMultiBit* x = new tMultiBit[D];
      uint32_t s2[] = {seed, seed, seed};
      tfhe_random_generator_setSeed(s2, 3);
      for (int i = 0; i < D; ++i) { x[i].size = 1; x[i].ctxt = new_gate_bootstrapping_ciphertext_array(1, params); lweSymEncrypt(&x[i].ctxt[0], in[i] * u, 1.0 / 65536, g_sk->lwe_key); }
      return x;
    };
    auto enc_sbits = [&]() {
      tBit* x = new_gate_bootstrapping_ciphertext_array(D, params);
      for (int i = 0; i < D; ++i) lweSymEncrypt(&x[i], (in[i] > 0 ? 1 : -1) * u, 1.0 / 65536, g_sk->lwe_key);
      return x;
    };
    auto run_r = [&](IntLayer& a, IntLayer& b) { return (tFixedPoint*)b.execute((tFixedPoint*)a.execute(enc_in(31))); };
    tFixedPoint* y1 = run_r(r0, r1);
    tBit* ys = (tBit*)s0.execute(enc_sbits());    // the other network in between
    tFixedPoint* y2 = run_r(r0, r1);
    bool okr = true, same_r = true, oks = true;
    for (int c = 0; c < C; ++c) {
      int want = b1[c];
      for (int k = 0; k < D; ++k) { const int a = b0[k] > 0 ? 15 : 0; const int w = w1[(size_t)k * C + c]; want += w * a - (w < 0 ? 1 : 0); }
      const int got = dec_int(&y1[c].ctxt[0], 16384);              // the unit the logits were summed in (the client's 4096 reads round(got / 4))
      if (got != want) printf("  relu net logit %d: got %d want %d\n", c, got, want);
      okr = okr && got == want;
      same_r = same_r && same(&y1[c].ctxt[0], &y2[c].ctxt[0], n);
    }
    for (int m = 0; m < MS; ++m) {
      int pre = bs[m];
      for (int k = 0; k < D; ++k) pre += ws[(size_t)k * MS + m] * (in[k] > 0 ? 1 : -1);
      oks = oks && dec_int(&ys[m], 4096) == (pre > 0 ? 1 : -1);
    }
    check("ReLU network: logits decrypt exactly in steps of 1/16384", okr);
    check("  ... identical words when run again after another network ran in between", same_r);
    check("sign network constructed beside it decrypts to sign(w.x)", oks);
    // a driver that re-initialises the SAME tDimensions object for a further network starts again at 1/4096
    FILE* fr2 = tmpfile(); make_r(fr2);
    IntLayer q0(E_FC, D, E_NO_POOL, E_ACTIVATION_RELU, &nq, g_bk), q1l(E_FC, C, E_NO_POOL, E_ACTIVATION_NONE, &np, g_bk);
    dr = dims(1, 1, D); dr.scale = 4;                                   // the same object, re-initialised as a driver's init() would
    pr = &dr;
    pr = (tDimensions*)q0.prep(fr2, pr);
    pr = (tDimensions*)q1l.prep(fr2, pr);
    tFixedPoint* y3 = run_r(q0, q1l);
    bool same3 = true;
    for (int c = 0; c < C; ++c) same3 = same3 && same(&y1[c].ctxt[0], &y3[c].ctxt[0], n);
    check("a re-initialised tDimensions object starts a further network at 1/4096 (same words as the first instance)", same3);
    fclose(fr); fclose(fs); fclose(fr2);
  }
  // ---- a ciphertext edited BETWEEN two stages must be seen by the second stage (lib/BinFunc.cpp:327-328,1073: a stage owns its
  //      input array; the stage API hands the intermediate array to the caller, who may touch it) ----
  {
    // Convolution (FC 40 -> 100) hands back 100 pre-activations whose device copy stays resident; the caller then adds a constant
    // to ONE of them -- row 2, which the sampled-row fingerprint of rounds 1-3 never looked at (it hashed rows 0, 1, 3, 4, 6, ...
    // of 100) -- and Quantize::execute must bootstrap the edited value, not the stale device copy.
    const int K2 = 40, M2 = 100, EDIT = 2;
    std::vector<int> w2((size_t)K2 * M2), bits2(K2);
    std::vector<int32_t> bias2(M2);
    for (auto& x : w2) { const unsigned t = rnd() % 10; x = t < 2 ? 0 : (t < 6 ? 1 : -1); }
    for (auto& b : bias2) b = (int32_t)(rnd() % 9) - 4;
    for (auto& b : bits2) b = (rnd() & 1) ? 1 : -1;
    std::vector<int> pre2(M2);
    for (int m = 0; m < M2; ++m) { pre2[m] = bias2[m]; for (int k = 0; k < K2; ++k) pre2[m] += w2[(size_t)k * M2 + m] * bits2[k]; }
    const int delta = pre2[EDIT] >= 0 ? -(pre2[EDIT] + 40) : (40 - pre2[EDIT]);     // moves the pre-activation to -40 / +40: the sign flips, far from 0
    FILE* f2 = tmpfile();
    put_ternary(f2, w2); put_ints(f2, bias2);
    put_ternary(f2, w2); put_ints(f2, bias2);
    rewind(f2);
    auto enc2 = [&]() {
      tBit* x = new_gate_bootstrapping_ciphertext_array(K2, params);
      uint32_t s2[] = {21, 22, 23};
      tfhe_random_generator_setSeed(s2, 3);
      for (int i = 0; i < K2; ++i) lweSymEncrypt(&x[i], bits2[i] * u, 1.0 / 32768, g_sk->lwe_key);
      return x;
    };
    tQParams qq; qq.shift_bits = 1;
    auto chain = [&](bool edit) {
      tDimensions dd = dims(1, 1, K2);
      BinFunc::Convolution cv(M2, &np.conv);
      BinFunc::Quantize qz(&qq);
      std::vector<tMultiBit> pbq(M2);
      cv.prep(f2, &dd, g_bk);
      qz.prep(f2, &dd, pbq.data(), NULL, g_bk);
      tMultiBit* mid = cv.execute(enc2());
      if (edit) mid[EDIT].ctxt[0].b += delta * u;              // a plain field write, as REDsec code does all over lib/*.cpp
      return qz.execute(mid, pbq.data());
    };
    tBit* plain = chain(false);
    tBit* edited = chain(true);
    bool others = true;
    for (int m = 0; m < M2; ++m) if (m != EDIT) others = others && same(&plain[m], &edited[m], n);
    const int want = pre2[EDIT] + delta >= 0 ? 1 : -1;
    check("a row edited between Convolution and Quantize is bootstrapped as edited (every word of the array is fingerprinted)",
          dec_int(&edited[EDIT], 4096) == want && dec_int(&plain[EDIT], 4096) == -want && !same(&plain[EDIT], &edited[EDIT], n));
    check("  ... and every other row is unchanged, word for word", others);
    fclose(f2);
  }

  // ---- IntFunc::Convolution with the reference's ENCRYPTED constants (lib/IntFunc.cpp:264-279; REDSEC_INTCONV=enc) ----
  {
    // 4x4x2 image, 3x3 same-padded convolution to 3 channels, ternary weights with zeros: a ternary-zero tap and a padding tap
    // each contribute the trivial sample -1/4096 (lweNoiselessTrivial(-mu_dynamic), :268 and :277), a +1 tap the input, a -1 tap
    // its negation (lweClear + lweSubTo, :219-220); Quantize::add_bias then adds trivial(bias/4096). The expected ciphertexts
    // are built here with the shim's host-side word operations and must equal the layer's output word for word.
    setenv("REDSEC_INTCONV", "enc", 1);
    const int Hh = 4, Ci = 2, Co = 3, Fh = 3;
    std::vector<int> wc((size_t)Fh * Fh * Ci * Co), px2(Hh * Hh * Ci);
    for (auto& x : wc) { const unsigned t = rnd() % 3; x = t == 0 ? 0 : (t == 1 ? 1 : -1); }
    for (auto& v : px2) v = (int)(rnd() % 61) - 30;
    std::vector<int32_t> bc = {7, -11, 3};
    FILE* f3 = tmpfile();
    put_ternary(f3, wc); put_ints(f3, bc);
    rewind(f3);
    tNetParams nc = np;
    nc.conv.window.h = nc.conv.window.w = Fh; nc.conv.stride.h = nc.conv.stride.w = 1; nc.conv.same_pad = true;
    IntLayer cl(E_CONV, Co, E_NO_POOL, E_ACTIVATION_NONE, &nc, g_bk);
    tDimensions dc = dims(Hh, Hh, Ci);
    cl.prep(f3, &dc);
    tMultiBit* xin = new tMultiBit[Hh * Hh * Ci];
    std::vector<LweSample*> keep(Hh * Hh * Ci);
    for (int i = 0; i < Hh * Hh * Ci; ++i) {
      xin[i].size = 1; xin[i].ctxt = new_gate_bootstrapping_ciphertext_array(1, params);
      lweSymEncrypt(&xin[i].ctxt[0], px2[i] * u, 1.0 / 4194304, g_sk->lwe_key);   // quiet samples: 18 of them sum per output and must still decode
      keep[i] = new_LweSample(params->in_out_params);          // execute() frees its input: the expectation needs its own copy
      lweCopy(keep[i], &xin[i].ctxt[0], params->in_out_params);
    }
    tFixedPoint* y = (tFixedPoint*)cl.execute(xin);
    unsetenv("REDSEC_INTCONV");
    bool okc = true, decs = true;
    LweSample* acc = new_LweSample(params->in_out_params);
    for (int ph = 0; ph < Hh && okc; ++ph) for (int pw = 0; pw < Hh; ++pw) for (int od = 0; od < Co; ++od) {
      lweNoiselessTrivial(acc, bc[od] * u, params->in_out_params);
      int plain = bc[od];
      for (int fh = 0; fh < Fh; ++fh) for (int fw = 0; fw < Fh; ++fw) for (int di = 0; di < Ci; ++di) {
        const int ih = ph + fh - 1, iw = pw + fw - 1;
        const int wv = wc[((size_t)(fh * Fh + fw) * Ci + di) * Co + od];                // get_filter_i, lib/IntFunc.cpp
        if (ih < 0 || ih >= Hh || iw < 0 || iw >= Hh || wv == 0) { acc->b -= u; plain -= 1; continue; }   // padding / ternary zero: trivial -1/4096
        const LweSample* in = keep[(ih * Hh + iw) * Ci + di];                            // get_input_i
        if (wv > 0) { lweAddTo(acc, in, params->in_out_params); plain += px2[(ih * Hh + iw) * Ci + di]; }
        else { lweSubTo(acc, in, params->in_out_params); plain -= px2[(ih * Hh + iw) * Ci + di]; }
      }
      const LweSample* got = &y[(ph * Hh + pw) * Co + od].ctxt[0];                       // get_output_i
      okc = okc && same(got, acc, n);
      decs = decs && dec_int(got, 4096) == plain;
    }
    check("IntLayer(E_CONV) under REDSEC_INTCONV=enc == the reference's ENCRYPTED branch restated with host word operations, word for word", okc);
    check("  ... and decrypts to sum(w x) - #(zero or padding taps) + bias", decs);
    fclose(f3);
  }
  printf("failures: %d\n", g_fail);
  return g_fail;
}
