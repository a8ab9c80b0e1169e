#!/usr/bin/env python3
"""One encrypted CIFAR image through the WHOLE network on the CPU oracle, stage by stage, and every stage's slab -- and the ten logit
ciphertexts -- compared word for word with what the GPU chain (redsec_amd.nets.EncryptedCifar, fused max-pool) produced from the same
ciphertext under the same key. The `-m gpu` suite checks sampled rows of every stage against the oracle (tests/test_gpu_cifar.py);
this is the unsampled statement, too slow for a suite (binarynet_small: 262,144 bootstraps at ~800 per second on a GPU box's 16
host cores): network-level parity at CIFAR scale, and with it the claim that the class of an encrypted image is whatever ANY exact
implementation of these parameters produces from this ciphertext.

The oracle side is test infrastructure: bootstraps by oracle/redsec_oracle.c (its FP64-FFT path, cross-checked against its exact paths
by tests/test_oracle_kat.py), linear stages by tests/linear_check.py (numpy). Nothing of it is in the product path.

  python tools/cifar_oracle_end_to_end.py [binarynet_small|binarynet] [image_index]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
import linear_check as lc
import oracle_lib as ol
import plain_model as pm
import redsec_amd
from redsec_amd import client, nets

net_name = sys.argv[1] if len(sys.argv) > 1 else "binarynet_small"
img = int(sys.argv[2]) if len(sys.argv) > 2 else 1
MU = 1 << 20
sk = client.SecretKeySet("redsec_small_v2", seed=13)
net = pm.CifarNet(net_name)
labels, pix = pm.load_cifar_images()
ct = sk.encrypt_image(pix[img], seed=4)

# ---- GPU ----
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
be.load_keys(sk.bk, sk.ksk)
taps = []
t0 = time.perf_counter()
out_gpu = nets.EncryptedCifar(be, net).run(torch.from_numpy(ct).cuda(), taps=taps)
torch.cuda.synchronize()
gpu_s = time.perf_counter() - t0
gpu = {r["name"]: (r["inputs"][0].cpu().numpy(), r["out"].cpu().numpy()) for r in taps}
out_gpu = out_gpu.cpu().numpy()
be.close()

# ---- oracle ----
class _K:
    pass
k = _K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
cores = bench.host_cpu_share()
ol.lib().ro_set_threads(cores)
octx = ol.Ctx(k)
octx.set_fft(True)
tor = nets.MnistSignNet.bias_to_torus
W = ct.shape[1]
report, boots, t_boot = [], 0, 0.0


def stage(name, pre, mu):
    global boots, t_boot
    t1 = time.perf_counter()
    bits = octx.bootstrap_batch(pre, mu)
    t_boot += time.perf_counter() - t1
    boots += pre.shape[0]
    g_in, g_out = gpu[name]
    rec = {"stage": name, "ciphertexts": int(pre.shape[0]), "input_slab_equal": bool(np.array_equal(pre, g_in)), "output_slab_equal": bool(np.array_equal(bits, g_out))}
    report.append(rec)
    print(json.dumps(rec), flush=True)
    return bits


t_all = time.perf_counter()
pre = ct.astype(np.int64).reshape(32, 32, 3, W).copy()
pre[..., -1] += tor(net.bias0).astype(np.int64)[None, None, :]
bits = stage("quantize0", lc.wrap32(pre).reshape(-1, W), MU)
H, C = 32, 3
for li, (sign, zero, bias) in enumerate(net.convs):
    Cout = sign.shape[3]
    shape = dict(H=H, Wd=H, Cin=C, Cout=Cout, fh=3, fw=3, stride_h=1, stride_w=1, off_h=1, off_w=1, Ho=H, Wo=H)
    pre = lc.conv_full(bits.reshape(H, H, C, W), shape, sign, zero, tor(bias)).reshape(-1, W)
    C = Cout
    pooled = li % 2 == 1
    bits = stage("conv%d" % (li + 1), pre, (1 << 28) if pooled else MU)
    if pooled:   # the OR of a 2x2 window as ONE bootstrap of the windowed sum + 3/16
        s = bits.astype(np.int64).reshape(H // 2, 2, H // 2, 2, C, W).sum(axis=(1, 3))
        s[..., -1] += 3 << 28
        bits = stage("maxpool%d" % (li + 1), lc.wrap32(s).reshape(-1, W), MU)
        H //= 2
v = bits
for i, (sign, zero, bias) in enumerate(net.fcs):
    w = lc.ternary_weights(sign, zero).astype(np.float64)
    assert w.shape[0] * 2.0 ** 31 < 2.0 ** 53
    pre = np.rint(w.T @ v.astype(np.float64)).astype(np.int64)
    pre[:, -1] += tor(bias).astype(np.int64)
    pre = lc.wrap32(pre)
    if i == len(net.fcs) - 1:
        out_cpu = pre
        break
    v = stage("fc%d" % (i + 1), pre, MU)
cpu_s = time.perf_counter() - t_all
logits_c, logits_g = sk.decrypt_ints(out_cpu), sk.decrypt_ints(out_gpu)
plain = pm.cifar_forward(net, pix[img])
print(json.dumps({"net": net_name, "image": img, "label": int(labels[img]), "bootstraps": boots, "stages": len(report),
                  "every_stage_input_and_output_slab_equal": all(r["input_slab_equal"] and r["output_slab_equal"] for r in report),
                  "logit_ciphertexts_equal": bool(np.array_equal(out_cpu, out_gpu)),
                  "class": {"oracle": int(np.argmax(logits_c)), "gpu": int(np.argmax(logits_g)), "plaintext": int(np.argmax(plain))},
                  "cpu_s": round(cpu_s, 1), "cpu_bootstrap_s": round(t_boot, 1), "cpu_cores": int(cores), "cpu_bootstraps_per_s": round(boots / t_boot, 1),
                  "gpu_s": round(gpu_s, 3), "speedup": round(cpu_s / gpu_s, 1)}))
