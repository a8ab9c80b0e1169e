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


# ---- nets/mnist/relu1024x<K>: the reference's PLAINTEXT flavour, restated -----------------------------------
def load_relu_net(name):
    from redsec_amd.nets import MnistReluNet
    blob = open(os.path.join(GOLD, "mnist_%s_var_prep.dat" % name), "rb").read()
    return MnistReluNet(blob, hidden_layers=int(name[-1]))


def relu_forward(net, pixels, taps=None):
    """pixels [784] -> integer logits [10], bit for bit what the reference's plaintext build computes when run
    on ONE thread (tests/golden/mnist_relu1024x*.json; its relu_shift loop shares a scratch value between
    OpenMP threads, lib/IntFunc.cpp:953): IntFunc::Convolution multiplies by -1 as ~x (IntOps::invert), so a
    negative tap contributes -x - 1; relu_shift is x = slope * pre + bias, y = x >> slope_bits,
    out = 0 if x < 0 else min(y, 15) (lib/IntFunc.cpp:964-967, lib/IntOps.cpp relu/shift)."""
    from redsec_amd import nets
    x = nets.relu_preprocess(pixels).reshape(28, 28)
    v = x.reshape(14, 2, 14, 2).sum(axis=(1, 3)).reshape(196) + int(net.bias0[0])
    stages, _ = net.stages()
    for li, ((sign, zero, bias, slope), (sb, _, _)) in enumerate(zip(net.fc, stages)):
        w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
        pre = v @ w - net.neg_taps(sign, zero)
        xb = slope.astype(np.int64) * pre + bias.astype(np.int64)
        v = np.where(xb < 0, 0, np.minimum(xb >> sb, (1 << net.SHIFT_BITS) - 1))
        if taps is not None:
            taps["pre%d" % (li + 1)], taps["act%d" % (li + 1)] = pre.copy(), v.copy()
    sign, zero, bias = net.final
    w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
    return v @ w - net.neg_taps(sign, zero) + bias.astype(np.int64)


# ---- CIFAR binarynet / binarynet_small (nets/cifar/*/net.cpp:96-209) ---------------------------------
CIFAR_WIDTHS = {"binarynet": ([128, 128, 256, 256, 512, 512], [1024, 1024]),
                "binarynet_small": ([64, 64, 128, 128, 256, 256], [512, 512])}


def load_cifar_images():
    d = json.load(open(os.path.join(GOLD, "cifar_images.json")))
    return np.array(d["labels"]), np.array(d["pixels"], dtype=np.int64)


class CifarNet:
    """IntLayer(NO_CONV, SIGN) ; 6 x BinLayer(CONV 3x3 same, SIGN, max-pool after every second) ;
    2 x BinLayer(FC, SIGN) ; BinLayer(FC_FINAL, NONE). Weight layouts as lib/BinFunc.cpp:388."""

    def __init__(self, name):
        from redsec_amd.nets import WeightReader
        convs, fcs = CIFAR_WIDTHS[name]
        r = WeightReader(open(os.path.join(GOLD, "cifar_%s_var_prep.dat" % name), "rb").read())
        self.bias0 = r.ints(3)
        self.convs = []
        cin = 3
        for cout in convs:
            sign, zero = r.ternary(3 * 3 * cin * cout)
            self.convs.append((sign.reshape(3, 3, cin, cout), zero.reshape(3, 3, cin, cout), r.ints(cout)))
            cin = cout
        k = 4 * 4 * cin
        self.fcs = []
        for m in fcs + [10]:
            sign, zero = r.ternary(k * m)
            self.fcs.append((sign.reshape(k, m), zero.reshape(k, m), r.ints(m)))
            k = m
        assert r.done()


def cifar_forward(net, pixels, taps=None):
    """-> integer logits [10]. `taps` (dict) receives, under the stage names of redsec_amd.nets.EncryptedCifar ("quantize0",
    "conv<k>", "maxpool<k>", "fc<k>"), (pre-activations or None, +-1 bits) flattened in (row, column, channel) order, and the
    pre-activation maps as "pre<k>"."""
    x = (2 * pixels - 255).reshape(32, 32, 3)
    pre = x + net.bias0.astype(np.int64)[None, None, :]
    bits = np.where(pre >= 0, 1, -1)
    if taps is not None:
        taps["quantize0"] = (pre.reshape(-1), bits.reshape(-1))
    for li, (sign, zero, bias) in enumerate(net.convs):
        H, W, C = bits.shape
        w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64).reshape(9 * C, -1)
        pad = np.zeros((H + 2, W + 2, C), np.int64)
        pad[1:-1, 1:-1] = bits                                   # out-of-image taps contribute 0 (BinFunc.cpp:266-289)
        cols = np.stack([pad[fh:fh + H, fw:fw + W] for fh in range(3) for fw in range(3)], axis=2).reshape(H * W, 9 * C)
        pre = (cols @ w + bias.astype(np.int64)).reshape(H, W, -1)
        bits = np.where(pre >= 0, 1, -1)
        if taps is not None:
            taps["pre%d" % (li + 1)] = pre
            taps["conv%d" % (li + 1)] = (pre.reshape(-1), bits.reshape(-1))
        if li % 2 == 1:                                          # E_MAXPOOL 2x2 on layers 2, 4, 6
            bits = bits.reshape(H // 2, 2, W // 2, 2, -1).max(axis=(1, 3))
            if taps is not None:
                taps["maxpool%d" % (li + 1)] = (None, bits.reshape(-1))
    v = bits.reshape(-1)
    for i, (sign, zero, bias) in enumerate(net.fcs):
        w = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
        pre = v @ w + bias.astype(np.int64)
        if i == len(net.fcs) - 1:
            return pre
        v = np.where(pre >= 0, 1, -1)
        if taps is not None:
            taps["fc%d" % (i + 1)] = (pre, v)
