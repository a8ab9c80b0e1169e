"""Plaintext (numpy) model of nets/mnist/sign1024x<K>: the CHECKER for the encrypted chain.

Restates the reference's plaintext flavour (lib/IntFunc.cpp SumPooling/Quantize, lib/BinFunc.cpp
Convolution/Quantize with `p_window = bit==0 ? -1 : 1`, lib/BinOps.cpp:207-217 binarize: x >= 0 -> 1)
and is itself pinned against the logits the reference's own `make ptxt` build prints
(tests/golden/mnist_sign1024x1.json). Test infrastructure only."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_images():
    d = json.load(open(os.path.join(GOLD, "mnist_images.json")))
    return np.array(d["labels"]), np.array(d["pixels"], dtype=np.int64)


def load_net(name):
    from redsec_amd.nets import MnistSignNet
    k = int(name[-1])
    blob = open(os.path.join(GOLD, "mnist_%s_var_prep.dat" % name), "rb").read()
    return MnistSignNet(blob, hidden_layers=k)


def forward(net, pixels, taps=None):
    """pixels [784] -> integer logits [10] (units of 1/4096 in the encrypted domain)."""
    x = (2 * pixels - 255).reshape(28, 28)                      # main.cpp:155 / encrypt_image.cpp:76
    pooled = x.reshape(14, 2, 14, 2).sum(axis=(1, 3)).reshape(196)
    pre = pooled + int(net.bias0[0])
    bits = np.where(pre >= 0, 1, -1)
    if taps is not None:
        taps["pre0"], taps["bits0"] = pre.copy(), bits.copy()
    for li, (sign, zero, bias) in enumerate(net.fc):
        w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
        pre = bits @ w + bias.astype(np.int64)
        bits = np.where(pre >= 0, 1, -1)
        if taps is not None:
            taps["pre%d" % (li + 1)], taps["bits%d" % (li + 1)] = pre.copy(), bits.copy()
    sign, zero, bias = net.final
    w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
    return bits @ w + bias.astype(np.int64)
