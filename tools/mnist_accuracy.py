#!/usr/bin/env python3
"""Encrypted vs plaintext classification of the bundled MNIST rows (tests/golden/*_images.json are the reference's own
plaintext logits): how many of the first K images the encrypted run classifies like the label / like the plaintext run.
The difference is the reference's parameter choice, not this backend (a 1/4096 message step against a
mod-switch noise of sigma ~ 7.7 steps at n = 350 flips weak-margin hidden units in any TFHE implementation).

  python tools/mnist_accuracy.py [net] [K]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import redsec_amd
from redsec_amd import client, nets
import plain_model as pm

name = sys.argv[1] if len(sys.argv) > 1 else "sign1024x1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sk = client.SecretKeySet("redsec_small_v2", seed=7)
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
be.load_keys(sk.bk, sk.ksk)
net = pm.load_net(name)
enc = nets.EncryptedMnist(be, net)
labels, pixels = pm.load_images()
hit_enc = hit_plain = agree = 0
for i in range(K):
    ct = torch.from_numpy(sk.encrypt_image(pixels[i], seed=100 + i)).cuda()
    cls = sk.classify(enc.run(ct).cpu().numpy())
    plain = int(np.argmax(pm.forward(net, pixels[i])))
    hit_enc += int(cls == labels[i]); hit_plain += int(plain == labels[i]); agree += int(cls == plain)
print("%s, %d images: plaintext %d correct, encrypted %d correct, encrypted == plaintext class on %d" % (name, K, hit_plain, hit_enc, agree))
