"""`make cpu-encrypt` of the reference, on the MI355X backend: the reference's own net.cpp/main.cpp
and client tools -- compiled UNMODIFIED (build/refnets) and linked to libredsec_layers.so -- run
keygen -> encrypt image -> encrypted inference on the GPU -> decrypt, end to end through files."""
import os
import re

import numpy as np
import pytest

import plain_model as pm
import refdrivers as rd

pytestmark = pytest.mark.gpu


def test_unmodified_sign1024x1_driver_end_to_end(tmp_path):
    if not rd.available():
        pytest.skip("build/refnets not shipped")
    client, netdir = rd.make_tree(str(tmp_path))
    assert rd.run("client_gen_secure_keyset.out", client).returncode == 0
    labels, pixels = pm.load_images()
    net = pm.load_net("sign1024x1")
    params, lwe_key = rd.read_secret_key(os.path.join(client, "secret.key"))
    ok = 0
    for i in (1, 3, 8):                                          # clear-margin images
        rd.write_image_csv(os.path.join(client, "img.csv"), labels[i], pixels[i])
        assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
        r = rd.run("mnist_sign1024x1_enc.out", netdir)
        assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout + r.stderr
        logits_ct = rd.read_ciphertexts(os.path.join(client, "network_output.ctxt"), 350, 10)
        r = rd.run("client_decrypt_image.out", client, "MNIST")
        m = re.search(r"Classification Result: (\d)", r.stdout)
        assert r.returncode == 0 and m, r.stdout + r.stderr
        # our own decryption of the same file agrees with the reference's decrypt tool
        phase = (logits_ct[:, 350].astype(np.int64) - (logits_ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
        dec = ((phase + (1 << 19)) >> 20) & 0xFFF
        dec = np.where(dec > 2048, dec - 4096, dec)
        assert int(np.argmax(dec)) == int(m.group(1))
        # logits stay close to the plaintext ones (hidden-unit sign flips move each by a few units)
        assert np.abs(dec - pm.forward(net, pixels[i])).max() < 200
        ok += int(int(m.group(1)) == labels[i])
    assert ok >= 2


@pytest.mark.parametrize("maxpool", ["fused", "chain"])
def test_unmodified_cifar_binarynet_small_driver(tmp_path, maxpool, monkeypatch):
    """BASELINE configs[3] shape through the C++ mirror: the reference's nets/cifar/binarynet_small
    {net,main}.cpp, unmodified -- IntLayer(NO_CONV) + 6 ternary 3x3 convolutions with three 2x2
    max-pools + 3 FC layers -- on one clear-margin image, with both max-pool forms of the layer mirror
    (DESIGN.md "Max-pool semantics": one bootstrap per window / bootsOR chain)."""
    import shutil
    if not os.path.exists(os.path.join(rd.REFNETS, "cifar_binarynet_small_enc.out")):
        pytest.skip("build/refnets not shipped")
    monkeypatch.setenv("REDSEC_MAXPOOL", maxpool)   # one-bootstrap OR of the window (default) / bootsOR chain
    client = str(tmp_path / "client")
    netdir = str(tmp_path / "nets" / "cifar" / "binarynet_small")
    os.makedirs(client); os.makedirs(netdir)
    shutil.copyfile(os.path.join(rd.GOLD, "cifar_binarynet_small_var_prep.dat"), os.path.join(netdir, "var_prep.dat"))
    assert rd.run("client_gen_secure_keyset.out", client).returncode == 0
    labels, pix = pm.load_cifar_images()
    net = pm.CifarNet("binarynet_small")
    i = 1
    with open(os.path.join(client, "img.csv"), "w") as f:
        f.write(",".join(str(int(v)) for v in [labels[i], 32, 32, 3] + list(pix[i])) + ",\n")
    assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
    r = rd.run("cifar_binarynet_small_enc.out", netdir)
    assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout + r.stderr
    r = rd.run("client_decrypt_image.out", client, "CIFAR-10")
    m = re.search(r"Classification Result: (\d)", r.stdout)
    plain = pm.cifar_forward(net, pix[i])
    assert m and int(m.group(1)) == int(np.argmax(plain)) == labels[i]       # margin 342 vs 36 in plaintext
    params, lwe_key = rd.read_secret_key(os.path.join(client, "secret.key"))
    ct = rd.read_ciphertexts(os.path.join(client, "network_output.ctxt"), 350, 10)
    phase = (ct[:, 350].astype(np.int64) - (ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
    dec = ((phase + (1 << 19)) >> 20) & 0xFFF
    dec = np.where(dec > 2048, dec - 4096, dec)
    assert np.corrcoef(dec, plain)[0, 1] > 0.8


def test_unmodified_cifar_binarynet_full_driver(tmp_path):
    """BASELINE configs[3] itself: the reference's nets/cifar/binarynet {net,main}.cpp, unmodified, one encrypted image
    on one MI355X (521,216 bootstraps with the fused max-pool, largest batch 131,072). Pinned against the plaintext
    model (itself pinned to the reference's plaintext build, tests/golden/cifar_binarynet.json): same class, logits
    correlated -- weak-margin units flip under mod-switch noise in the reference too (SURVEY.md hard part 7)."""
    import shutil
    import time
    if not os.path.exists(os.path.join(rd.REFNETS, "cifar_binarynet_enc.out")):
        pytest.skip("build/refnets not shipped")
    client = str(tmp_path / "client")
    netdir = str(tmp_path / "nets" / "cifar" / "binarynet")
    os.makedirs(client); os.makedirs(netdir)
    shutil.copyfile(os.path.join(rd.GOLD, "cifar_binarynet_var_prep.dat"), os.path.join(netdir, "var_prep.dat"))
    assert rd.run("client_gen_secure_keyset.out", client).returncode == 0
    labels, pix = pm.load_cifar_images()
    net = pm.CifarNet("binarynet")
    plain = [pm.cifar_forward(net, pix[i]) for i in range(8)]
    margins = [np.sort(p)[-1] - np.sort(p)[-2] if int(np.argmax(p)) == int(labels[i]) else -1 for i, p in enumerate(plain)]
    i = int(np.argmax(margins))                                   # the clearest correctly classified image of the first 8
    with open(os.path.join(client, "img.csv"), "w") as f:
        f.write(",".join(str(int(v)) for v in [labels[i], 32, 32, 3] + list(pix[i])) + ",\n")
    assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
    t0 = time.time()
    r = rd.run("cifar_binarynet_enc.out", netdir)
    wall = time.time() - t0
    assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout + r.stderr
    assert wall < 60, wall                                        # 4.4 s on an idle MI355X incl. reading the 218 MB key file
    params, lwe_key = rd.read_secret_key(os.path.join(client, "secret.key"))
    ct = rd.read_ciphertexts(os.path.join(client, "network_output.ctxt"), 350, 10)
    phase = (ct[:, 350].astype(np.int64) - (ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
    dec = ((phase + (1 << 19)) >> 20) & 0xFFF
    dec = np.where(dec > 2048, dec - 4096, dec)
    assert int(np.argmax(dec)) == int(np.argmax(plain[i])) == int(labels[i])
    assert np.corrcoef(dec, plain[i])[0, 1] > 0.5


@pytest.mark.parametrize("family,net,devices,lazy,staged", [("mnist", "sign1024x1", "0,0", False, False), ("mnist", "relu1024x1", "0,0,0", False, False),
                                                            ("mnist", "sign1024x1", "0,0,0,0", True, False), ("cifar", "binarynet_small", "0,0", False, False),
                                                            ("cifar", "binarynet_small", "0,0,0,0", True, False),
                                                            ("mnist", "relu1024x1", "0,0,0", False, True)])
def test_driver_sharded_over_several_contexts_equals_single_device(tmp_path, monkeypatch, family, net, devices, lazy, staged):
    """Gate-parallel evaluation of ONE image inside the C++ layer mirror (the reference's shape: enc_segs[NUM_GPUS], one host
    thread per GPU, lib/GPU/BinFunc_gpu.cu:119-137): REDSEC_DEVICES lists the devices, every bootstrapped stage is split
    contiguously across one context per entry and the slices are exchanged device to device before the next linear stage.
    On a one-GPU box the same device is listed several times -- several contexts, the same code path, ragged slices with
    three -- and the unmodified driver's network_output.ctxt must equal the single-device run BYTE FOR BYTE. The exchange is
    rs_allgather_rows: one host thread per context, event-ordered asynchronous copies, no device-wide waits. `lazy` adds
    REDSEC_LAZY_HOST=1 to the sharded run: intermediate host arrays stay unfilled, only the logits come down. `staged` adds
    RS_FORCE_HOST_STAGED=1: every slice travels through pinned host memory (source D2H once, destinations H2D) -- the path a
    device pair without peer access takes, which a one-GPU box cannot reach otherwise."""
    import shutil
    from redsec_amd import client
    exe = "%s_%s_enc.out" % (family, net)
    if not os.path.exists(os.path.join(rd.REFNETS, exe)):
        pytest.skip("build/refnets not shipped")
    cdir = str(tmp_path / "client")
    netdir = str(tmp_path / "nets" / family / net)
    os.makedirs(cdir); os.makedirs(netdir)
    shutil.copyfile(os.path.join(rd.GOLD, "%s_%s_var_prep.dat" % (family, net)), os.path.join(netdir, "var_prep.dat"))
    assert rd.run("client_gen_secure_keyset.out", cdir).returncode == 0
    keys = client.read_tfhe_keyset(open(os.path.join(cdir, "secret.key"), "rb"), secret=True)
    sk = client.SecretKeySet("redsec_small_v2", seed=1)
    sk.lwe_key = keys["lwe_key"]
    if family == "mnist":
        labels, pixels = pm.load_images()
        ct = sk.encrypt_image(pixels[4], seed=9, preprocess="relu" if net.startswith("relu") else "sign")
    else:
        labels, pixels = pm.load_cifar_images()
        ct = sk.encrypt_image(pixels[1], seed=9)
    with open(os.path.join(cdir, "image.ctxt"), "wb") as f:
        client.write_ciphertexts(f, ct)
    outs = []
    for dev in (None, devices):
        if dev is None:
            monkeypatch.delenv("REDSEC_DEVICES", raising=False)
        else:
            monkeypatch.setenv("REDSEC_DEVICES", dev)
            if lazy:
                monkeypatch.setenv("REDSEC_LAZY_HOST", "1")
            if staged:
                monkeypatch.setenv("RS_FORCE_HOST_STAGED", "1")
        r = rd.run(exe, netdir)
        assert r.returncode == 0 and "Result ctxts loaded" in r.stdout, r.stdout + r.stderr
        outs.append(open(os.path.join(cdir, "network_output.ctxt"), "rb").read())
    assert len(outs[0]) == 10 * (4 * 350 + 16) and outs[0] == outs[1]
