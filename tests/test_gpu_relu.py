"""ReLU path on the GPU: the programmable bootstrap against the oracle word for
word; nets/mnist/relu1024x1 through redsec_amd.nets.EncryptedMnistRelu against the oracle chain stage by stage and
against the reference's plaintext logits at decrypt level; the reference's UNMODIFIED relu1024x{1,2,3} drivers through
the C++ layer mirror, equal to the Python chain word for word."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import plain_model as pm
import refdrivers as rd

pytestmark = pytest.mark.gpu


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, np.int32)).cuda()


@pytest.mark.parametrize("mode", ["fft", "exact"])
@pytest.mark.parametrize("fix,name", [("toy_default", "default128"), ("toy_redsec", "redsec_small_v2"), ("full_redsec", "redsec_small_v2")])
def test_lut_bootstrap_equals_oracle(fix, name, mode, request):
    import redsec_amd
    ks, ctx = request.getfixturevalue(fix)
    be = redsec_amd.Backend(redsec_amd.params(name, n=ks.p.n), 0)
    be.load_keys(ks.bk, ks.ksk)
    be.set_mode(mode)
    rng = np.random.default_rng(9)
    N = ks.p.N
    luts = rng.integers(-2**31, 2**31, (5, N)).astype(np.int32)          # arbitrary test polynomials
    luts[0] = (np.arange(N) // 64 * (1 << 24)).astype(np.int32)          # a staircase
    cus = be.info()["num_cus"]
    sizes = (7, 2 * cus + 3, 8 * cus + 1) if fix != "full_redsec" else (33,)
    for B in sizes:                                                       # cooperative / per-wave or duo / lock-step forms
        ct = ks.encrypt(rng.integers(-2**31, 2**31, B), 2.0 ** -15, 100 + B)
        got = be.bootstrap_lut(_dev(ct), _dev(luts)).cpu().numpy()
        sample = np.r_[0:min(B, 12), max(0, B - 12):B] if B > 24 else np.arange(B)
        # ciphertext b uses luts[b % 5]: keep the sample's own table assignment
        for t in range(5):   # one oracle call per table (its batch runs under OpenMP)
            rows = sample[sample % 5 == t]
            if rows.size:
                assert np.array_equal(got[rows], ctx.bootstrap_lut_batch(ct[rows], luts[t:t + 1])), (B, t)
    if mode == "fft":
        assert be.rounding_certificate() < 0.2 and be.fft_fallbacks() == 0
    be.close()


def _oracle_ctx(sk):
    class K:
        pass
    k = K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
    c = ol.Ctx(k)
    c.set_fft(True)        # the fast CPU path; bit-equal to the exact paths (tests/test_oracle_kat.py)
    return c


def test_relu1024x1_equals_oracle_chain_word_for_word():
    import torch
    import redsec_amd
    import oracle_net
    from redsec_amd import client, nets
    sk = client.SecretKeySet("redsec_small_v2", seed=11)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    net = pm.load_relu_net("relu1024x1")
    labels, pixels = pm.load_images()
    ct = sk.encrypt_image(pixels[2], seed=5, preprocess="relu")
    cpu_taps, gpu_taps = {}, {}
    cpu = oracle_net.run_relu(_oracle_ctx(sk), net, ct, cpu_taps)
    gpu = nets.EncryptedMnistRelu(be, net).run(torch.from_numpy(ct).cuda(), gpu_taps)
    for name in ("in0", "pre1", "act1"):
        assert np.array_equal(gpu_taps[name].cpu().numpy().reshape(cpu_taps[name].shape), cpu_taps[name]), name
    assert np.array_equal(gpu.cpu().numpy(), cpu)
    assert be.rounding_certificate() < 0.2 and be.fft_fallbacks() == 0
    be.close()


@pytest.mark.parametrize("name", ["relu1024x1", "relu1024x2", "relu1024x3"])
def test_relu_nets_decrypt_to_the_reference_plaintext_logits(name):
    """Decrypt level against tests/golden/mnist_relu1024x*.json (the reference's own plaintext build): every stage equals
    the plaintext stage applied to what the previous stage actually produced, up to the mod-switch noise of one
    programmable bootstrap; argmax and logits follow the plaintext ones."""
    import torch
    import redsec_amd
    from redsec_amd import client, nets
    sk = client.SecretKeySet("redsec_small_v2", seed=21)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    net = pm.load_relu_net(name)
    enc = nets.EncryptedMnistRelu(be, net)
    gold = json.load(open(os.path.join(pm.GOLD, "mnist_%s.json" % name)))["logits"]
    labels, pixels = pm.load_images()
    stages, _ = net.stages()
    unit = net.LOGIT_UNIT
    agree, corr = 0, []
    for i in range(12):
        ct = torch.from_numpy(sk.encrypt_image(pixels[i], seed=40 + i, preprocess="relu")).cuda()
        taps = {}
        out = enc.run(ct, taps)
        dec = sk.decrypt_ints(out.cpu().numpy(), msize=(1 << 32) // unit)
        # stage-wise: activations are 4-bit values, and the next pre-activation is the ternary sum of the produced ones
        v = sk.decrypt_ints(taps["in0"].cpu().numpy())
        for li, ((sign, zero, bias, slope), (sb, u_in, u_out)) in enumerate(zip(net.fc, stages)):
            w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
            want_pre = v @ w - net.neg_taps(sign, zero)
            assert np.abs(want_pre).max() < (1 << 30) // u_in           # inside the quarter-turn window
            pre = sk.decrypt_ints(taps["pre%d" % (li + 1)].cpu().numpy(), msize=(1 << 32) // u_in) - (1 << 30) // u_in
            assert np.abs(pre - want_pre).max() <= (16 if li == 0 else 40), (li, np.abs(pre - want_pre).max())
            act = sk.decrypt_ints(taps["act%d" % (li + 1)].cpu().numpy(), msize=(1 << 32) // u_out)
            assert act.min() >= 0 and act.max() <= 15
            xb = slope.astype(np.int64) * want_pre + bias
            ideal = np.where(xb < 0, 0, np.minimum(xb >> sb, 15))
            assert np.abs(act - ideal).mean() < 1.5, (li, np.abs(act - ideal).mean())   # mod-switch noise: a few levels on steep neurons
            v = act
        sign, zero, bias = net.final
        w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
        assert np.abs(dec - (v @ w - net.neg_taps(sign, zero) + bias)).max() <= 12   # final layer of the produced activations
        agree += int(np.argmax(dec) == np.argmax(gold[i]))
        corr.append(np.corrcoef(dec, gold[i])[0, 1])
    assert agree >= 10 and np.mean(corr) > 0.95, (agree, corr)
    assert be.rounding_certificate() < 0.2 and be.fft_fallbacks() == 0
    be.close()


@pytest.mark.parametrize("name", ["relu1024x1", "relu1024x3"])
def test_unmodified_relu_driver_equals_python_chain(tmp_path, name):
    """nets/mnist/relu1024x*/{net,main}.cpp, compiled unmodified against the layer mirror: keys from the reference's
    own keygen tool (TFHE-format files), the image encrypted by redsec_amd.client with the ReLU nets' own input map
    (the reference's encrypt_image only knows 2p - 255), the driver's network_output.ctxt equal WORD FOR WORD to
    redsec_amd.nets.EncryptedMnistRelu on the same key and image, and decrypting to the plaintext class."""
    import torch
    import redsec_amd
    from redsec_amd import client, nets
    if not os.path.exists(os.path.join(rd.REFNETS, "mnist_%s_enc.out" % name)):
        pytest.skip("build/refnets not shipped")
    cdir, netdir = rd.make_tree(str(tmp_path), name)
    assert rd.run("client_gen_secure_keyset.out", cdir).returncode == 0
    keys = client.read_tfhe_keyset(open(os.path.join(cdir, "secret.key"), "rb"), secret=True)
    sk = client.SecretKeySet("redsec_small_v2", seed=1)            # shell for encrypt/decrypt, re-keyed from the file
    sk.lwe_key, sk.tlwe_key, sk.bk, sk.ksk = keys["lwe_key"], keys["tlwe_key"], keys["bk"], keys["ksk"]
    labels, pixels = pm.load_images()
    net = pm.load_relu_net(name)
    i = 1
    ct = sk.encrypt_image(pixels[i], seed=77, preprocess="relu")
    with open(os.path.join(cdir, "image.ctxt"), "wb") as f:
        client.write_ciphertexts(f, ct)
    r = rd.run("mnist_%s_enc.out" % name, netdir)
    assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout + r.stderr
    driver = rd.read_ciphertexts(os.path.join(cdir, "network_output.ctxt"), 350, 10)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    chain = nets.EncryptedMnistRelu(be, net).run(torch.from_numpy(ct).cuda()).cpu().numpy()
    assert np.array_equal(driver, chain)
    unit = net.LOGIT_UNIT
    dec = sk.decrypt_ints(driver, msize=(1 << 32) // unit)
    plain = pm.relu_forward(net, pixels[i])
    assert int(np.argmax(dec)) == int(np.argmax(plain)) == labels[i]
    assert np.corrcoef(dec, plain)[0, 1] > 0.95
    be.close()
    # the reference's own, unmodified client/decrypt_image.cpp (message space 4096) reads the same file: it sees round(logit / 4)
    # of logits that travel in 1/16384 steps, i.e. the same class unless two logits are within a few units of each other
    import re
    r = rd.run("client_decrypt_image.out", cdir, "MNIST")
    m = re.search(r"Classification Result: (\d)", r.stdout)
    assert r.returncode == 0 and m and int(m.group(1)) == labels[i], r.stdout + r.stderr
