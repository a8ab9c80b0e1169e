#!/usr/bin/env python3
"""Network-level agreement of encrypted CIFAR binarynet with the plaintext model (which is pinned to the reference's own
plaintext build, tests/golden/cifar_binarynet.json): K bundled images x S encryption-noise seeds, fused max-pool, one MI355X.
Kernel-level parity is exact (tests/test_gpu_cifar.py); what this measures is the reference's own property that weak-margin
units flip under the 2N = 2048 mod-switch (SURVEY.md hard part 7), image by image.
  python tools/cifar_agreement.py [images=4] [seeds=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import redsec_amd
from redsec_amd import client, nets
import plain_model as pm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sk = client.SecretKeySet("redsec_small_v2", seed=7)
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
be.load_keys(sk.bk, sk.ksk)
net = pm.CifarNet("binarynet")
enc = nets.EncryptedCifar(be, net)
labels, pix = pm.load_cifar_images()
plain = [pm.cifar_forward(net, pix[i]) for i in range(len(labels))]
margin = [np.sort(p)[-1] - np.sort(p)[-2] if int(np.argmax(p)) == int(labels[i]) else -1 for i, p in enumerate(plain)]
order = np.argsort(margin)[::-1][:K]                       # the K clearest correctly classified images
same_class = right = runs = 0
for i in order:
    for s in range(S):
        t0 = time.time()
        out = enc.run(torch.from_numpy(sk.encrypt_image(pix[i], seed=1000 * int(i) + s)).cuda())
        logits = sk.decrypt_ints(out.cpu().numpy())
        c = int(np.argmax(logits))
        runs += 1; same_class += c == int(np.argmax(plain[i])); right += c == int(labels[i])
        print("image %2d (label %d, plaintext margin %3d) seed %d: encrypted class %d, corr(logits, plaintext) %.2f, %.2f s"
              % (i, labels[i], margin[i], s, c, float(np.corrcoef(logits, plain[i])[0, 1]), time.time() - t0), flush=True)
print("encrypted class = plaintext class in %d of %d runs; = label in %d" % (same_class, runs, right))
be.close()
