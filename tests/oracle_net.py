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


def run_relu(ctx, net, image_ct, taps=None):
    """relu1024xK on the oracle: the chain of redsec_amd/nets.py::EncryptedMnistRelu stage by stage -- sum-pool
    + bias, then per hidden layer a ternary FC (+ quarter turn, -1 per negative tap) and ONE programmable
    bootstrap per neuron (oracle_lib.Ctx.bootstrap_lut_batch = tfhe_blindRotateAndExtract_FFT + keyswitch),
    then the final FC + bias. net: redsec_amd.nets.MnistReluNet. The test polynomials are the product's own
    tables (nets.relu_luts): what is being checked is the bootstrap, not the table."""
    from redsec_amd import nets
    W = image_ct.shape[1]

    def words(v):
        out = np.zeros((len(v), W), np.int64)
        out[:, W - 1] = np.asarray(v, np.int64)
        return out
    x = image_ct.astype(np.int64).reshape(14, 2, 14, 2, W)
    v = _wrap(x.sum(axis=(1, 3)).reshape(196, W) + words(np.resize(net.bias0.astype(np.int64) * nets.UNIT_4096, 196)))
    if taps is not None:
        taps["in0"] = v
    stages, logit_unit = net.stages()
    for li, ((sign, zero, bias, slope), (sb, u_in, u_out)) in enumerate(zip(net.fc, stages)):
        pre = _wrap(ol.linear_fc(v, sign, zero).astype(np.int64) + words(nets.QUARTER - net.neg_taps(sign, zero) * u_in))
        v = ctx.bootstrap_lut_batch(pre, nets.relu_luts(slope, bias, sb, net.SHIFT_BITS, u_in, u_out))
        if taps is not None:
            taps["pre%d" % (li + 1)], taps["act%d" % (li + 1)] = pre, v
    sign, zero, bias = net.final
    out = _wrap(ol.linear_fc(v, sign, zero).astype(np.int64) + words((bias.astype(np.int64) - net.neg_taps(sign, zero)) * logit_unit))
    return _wrap(out.astype(np.int64) * (net.LOGIT_UNIT // logit_unit))      # handed back in the client's 1/4096 steps (exact multiple)
