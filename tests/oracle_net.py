"""Encrypted sign1024xK inference on the CPU oracle (tests/oracle_lib.py): the same layer chain as
redsec_amd/nets.py::EncryptedMnist, stage by stage (IntLayer sum-pool + bias -> sign bootstrap ->
ternary FC + bias -> sign bootstrap -> ... -> final FC + bias), with numpy for the word-wise linear
stages. Test infrastructure only (BASELINE configs[0]: the CPU plumbing baseline; and the checker of
the ciphertext-exact end-to-end GPU test)."""
import numpy as np

import oracle_lib as ol

MU_SIGN = 1 << 20   # modSwitchToTorus32(1, 4096), BinOps_enc.cpp:184


def _bias_words(b, W):
    """trivial samples of b/4096: only the last word (b) is non-zero."""
    out = np.zeros((len(b), W), np.int64)
    out[:, W - 1] = (np.asarray(b, np.int64) << 20)
    return out


def _wrap(x):
    return (x & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def run(ctx, net, image_ct, taps=None):
    """ctx: oracle_lib.Ctx for the REDsec parameter set; net: redsec_amd.nets.MnistSignNet;
    image_ct int32 [784][W]. Returns int32 [10][W]."""
    W = image_ct.shape[1]
    x = image_ct.astype(np.int64).reshape(14, 2, 14, 2, W)
    pre0 = _wrap(x.sum(axis=(1, 3)).reshape(196, W) + _bias_words(np.resize(net.bias0, 196), W))   # SumPooling + add_int(bias[i % depth])
    bits = ctx.bootstrap_batch(pre0, MU_SIGN)
    if taps is not None:
        taps["pre0"], taps["bits0"] = pre0, bits
    for li, (sign, zero, bias) in enumerate(net.fc):
        pre = _wrap(ol.linear_fc(bits, sign, zero).astype(np.int64) + _bias_words(bias, W))
        bits = ctx.bootstrap_batch(pre, MU_SIGN)
        if taps is not None:
            taps["pre%d" % (li + 1)], taps["bits%d" % (li + 1)] = pre, bits
    sign, zero, bias = net.final
    return _wrap(ol.linear_fc(bits, sign, zero).astype(np.int64) + _bias_words(bias, W))
