"""ReLU path, CPU side: the plaintext restatement against the reference's own plaintext
build (tests/golden/mnist_relu1024x*.json, made by tests/golden/make_golden.py through oracle/ref_logits_driver.cpp),
the test-polynomial tables, and the oracle's programmable bootstrap."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import plain_model as pm
from redsec_amd import nets


@pytest.mark.parametrize("name", ["relu1024x1", "relu1024x2", "relu1024x3"])
def test_plain_relu_model_equals_reference_plaintext_build(name):
    gold = json.load(open(os.path.join(pm.GOLD, "mnist_%s.json" % name)))["logits"]
    labels, pixels = pm.load_images()
    net = pm.load_relu_net(name)
    hits = 0
    for i in range(100):
        logits = pm.relu_forward(net, pixels[i])
        assert list(logits) == gold[i], i
        hits += int(np.argmax(logits) == labels[i])
    assert hits >= 90


def test_relu_luts_tabulate_the_plaintext_staircase():
    slope = np.array([36, 21, 49], np.int32)
    bias = np.array([2164, -3737, 9024], np.int32)
    lut = nets.relu_luts(slope, bias, 6, 4, nets.UNIT_4096, nets.RELU_UNIT)
    assert lut.shape == (3, 1024) and lut.dtype == np.int32
    for m in range(3):
        for t in (0, 100, 511, 512, 513, 700, 1023):
            pre = (t - 512) * 2                                   # 2 integer steps of 1/4096 per mod-switched phase step
            x = int(slope[m]) * pre + int(bias[m])
            want = 0 if x < 0 else min(x >> 6, 15)
            assert lut[m, t] == want * nets.RELU_UNIT
    assert np.all(np.diff(lut.astype(np.int64), axis=1) >= 0)    # positive slopes: monotone staircases
    wide = nets.relu_luts(slope, bias, 8, 4, nets.RELU_UNIT, nets.RELU_UNIT)
    assert wide[0, 512 + 10] == min((36 * 80 + 2164) >> 8, 15) * nets.RELU_UNIT    # 8 steps of 1/16384 per phase step
    assert nets.relu_slope_bits(4.0, 4) == 6 and nets.relu_slope_bits(15.0, 4) == 8   # lib/IntFunc.cpp:812-815


def test_oracle_lut_bootstrap_constant_polynomial_is_the_sign_bootstrap(toy_redsec):
    ks, ctx = toy_redsec
    mu = ol.to_torus(1, 4096)
    ms = np.array([-900, -33, 40, 1200])
    ct = ks.encrypt([ol.to_torus(int(m), 4096) for m in ms], 2.0 ** -15, 5)
    lut = np.full((1, ks.p.N), mu, np.int32)
    assert np.array_equal(ctx.bootstrap_lut_batch(ct, lut), ctx.bootstrap_batch(ct, mu))


def test_oracle_lut_bootstrap_evaluates_the_table(toy_redsec):
    """out = lut[pbar] for the mod-switched phase pbar < N, -lut[pbar - N] beyond (tfhe_blindRotateAndExtract_FFT)."""
    ks, ctx = toy_redsec
    N = ks.p.N
    lut = (np.arange(N, dtype=np.int64) // 64 * (1 << 24)).astype(np.int32).reshape(1, N)      # 16 plateaus of 64 steps
    pbar = np.array([32, 96, 500, 1000, 1024 + 32, 2047 - 30])                                 # plateau centres
    ct = ks.encrypt([int(p) << 21 for p in pbar], 2.0 ** -20, 6)
    out = ctx.bootstrap_lut_batch(ct, lut)
    got = np.round(ks.phase(out).astype(np.float64) / (1 << 24)).astype(int)
    want = np.where(pbar < N, pbar // 64, -((pbar - N) // 64))
    assert np.array_equal(got, want)
    # two test polynomials, alternating over the batch: ciphertext b uses luts[b % 2]
    luts = np.concatenate([lut, -lut])
    out2 = ctx.bootstrap_lut_batch(ct, luts)
    got2 = np.round(ks.phase(out2).astype(np.float64) / (1 << 24)).astype(int)
    assert np.array_equal(got2, want * np.where(np.arange(len(pbar)) % 2 == 0, 1, -1))


def test_relu1024x1_on_the_oracle_decrypts_to_the_plaintext_logits():
    """The corrected encrypted ReLU chain (tests/oracle_net.py::run_relu: 1,024 programmable bootstraps) against the
    reference's plaintext logits: same argmax, logits correlated (mod-switch noise moves single activations by a few
    levels, as it moves the sign nets' weak-margin bits -- SURVEY.md hard part 7)."""
    import oracle_net
    from redsec_amd import client
    sk = client.SecretKeySet("redsec_small_v2", seed=11)

    class K:
        pass
    k = K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
    octx = ol.Ctx(k)
    octx.set_fft(True)
    net = pm.load_relu_net("relu1024x1")
    labels, pixels = pm.load_images()
    ct = sk.encrypt_image(pixels[0], seed=8, preprocess="relu")
    taps, ptaps = {}, {}
    out = oracle_net.run_relu(octx, net, ct, taps)
    plain = pm.relu_forward(net, pixels[0], ptaps)
    unit = net.LOGIT_UNIT
    dec = sk.decrypt_ints(out, msize=(1 << 32) // unit)
    assert np.abs(sk.decrypt_ints(taps["pre1"]) - 1024 - ptaps["pre1"]).max() <= 16        # fresh noise of 784 pixels, quarter turn removed
    act = sk.decrypt_ints(taps["act1"], msize=16384)
    assert act.min() >= 0 and act.max() <= 15 and np.abs(act - ptaps["act1"]).mean() < 1.0
    assert int(np.argmax(dec)) == int(np.argmax(plain)) == labels[0]
    assert np.corrcoef(dec, plain)[0, 1] > 0.97
