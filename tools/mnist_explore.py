"""Exploratory: encrypted sign1024x1 on the GPU, layer-wise comparison with the plaintext checker."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import redsec_amd
from redsec_amd import client, nets
import plain_model as pm

name = sys.argv[1] if len(sys.argv) > 1 else "sign1024x1"
sk = client.SecretKeySet("redsec_small_v2", seed=7)
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
be.load_keys(sk.bk, sk.ksk)
net = pm.load_net(name)
enc = nets.EncryptedMnist(be, net)
labels, pixels = pm.load_images()
gold = json.load(open(os.path.join(pm.GOLD, "mnist_%s.json" % name)))
agree = 0
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    ct = torch.from_numpy(sk.encrypt_image(pixels[i], seed=100 + i)).cuda()
    taps = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = enc.run(ct, taps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ptaps = {}
    plog = pm.forward(net, pixels[i], ptaps)
    dec = sk.decrypt_ints(out.cpu().numpy())
    pre0 = sk.decrypt_ints(taps["pre0"].cpu().numpy())
    b0 = np.where(sk.phase(taps["bits0"].cpu().numpy()) > 0, 1, -1)
    flips0 = int((b0 != ptaps["bits0"]).sum()); strong0 = np.abs(ptaps["pre0"]) >= 32
    sflips0 = int((b0[strong0] != ptaps["bits0"][strong0]).sum())
    # layer 1 consistency given the ACTUAL bits0
    s, z, b = net.fc[0]
    w = np.where(z == 1, 0, np.where(s == 1, 1, -1)).astype(np.int64)
    pre1_expect = b0 @ w + b
    pre1 = sk.decrypt_ints(taps["pre1"].cpu().numpy())
    b1 = np.where(sk.phase(taps["bits1"].cpu().numpy()) > 0, 1, -1)
    strong1 = np.abs(pre1_expect) >= 32
    flips1 = int((b1 != np.where(pre1_expect >= 0, 1, -1)).sum()); sflips1 = int((b1[strong1] != np.where(pre1_expect >= 0, 1, -1)[strong1]).sum())
    last_bits = np.where(sk.phase(taps["bits%d" % len(net.fc)].cpu().numpy()) > 0, 1, -1)
    s, z, b = net.final
    w = np.where(z == 1, 0, np.where(s == 1, 1, -1)).astype(np.int64)
    final_expect = last_bits @ w + b
    # noise of bootstrapped outputs in units of 1/4096
    ph_bits1 = sk.phase(taps["bits1"].cpu().numpy()).astype(np.float64) / 2**20
    print(f"img {i} label {labels[i]} enc_argmax {int(np.argmax(dec))} plain_argmax {int(np.argmax(plog))} ms {dt*1e3:.1f} "
          f"max|pre0-plain| {int(np.abs(pre0 - ptaps['pre0']).max())} flips0 {flips0} (strong {sflips0}) "
          f"max|pre1-expect| {int(np.abs(pre1 - pre1_expect).max())} flips1 {flips1}/{len(b1)} (strong {sflips1}) "
          f"max|logit-expect| {int(np.abs(dec - final_expect).max())} max|logit-plain| {int(np.abs(dec - plog).max())} "
          f"bits1 noise std {np.std(np.abs(ph_bits1) - 1):.4f}")
    agree += int(np.argmax(dec) == np.argmax(plog))
print("argmax agreement with plaintext:", agree)
