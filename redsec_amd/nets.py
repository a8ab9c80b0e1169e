"""Encrypted inference of the reference's MNIST sign networks on the GPU, device-resident.

Host-side mirror of the layer chain `nets/mnist/sign1024x{1,2,3}/net.cpp:87-112` builds:

    IntLayer(E_NO_CONV, 1, E_SUMPOOL, SIGN)            28x28x1 --sumpool 2x2--> 14x14x1 --+bias, sign-->
    BinLayer(E_FC, 1024, NO_POOL, SIGN)  x K           196|1024 -> 1024 ternary FC --+bias, sign-->
    BinLayer(E_FC_FINAL, 10, NO_POOL, NONE)            1024 -> 10 ternary FC --+bias--> logits

in the fixed order conv -> sumpool -> quantize of `BinLayer::run` / `IntLayer::run`
(lib/BinLayer.cpp:150-241, lib/IntLayer.cpp:153-235). Every stage is ONE launch over all output
ciphertexts (the reference runs an OpenMP loop of per-ciphertext TFHE calls, e.g.
lib/BinFunc.cpp:1056-1071); ciphertexts stay in HBM between stages: the image goes up, ten logit
ciphertexts come back.

Weight file format (lib/BinOps_enc.cpp:247-305): per tensor a 1-byte tag (1 BIN, 2 TERN, 3 UINT32,
4 INT32) followed by MSB-first bit-packed weights (TERN: 2 bits per weight = sign, is-zero) or
int32[len].
"""
import numpy as np

FMT_BIN, FMT_TERN, FMT_UINT32, FMT_INT32 = 1, 2, 3, 4
MU_SIGN = 1 << 20  # modSwitchToTorus32(1, 4096), BinOps_enc.cpp:184


class WeightReader:
    def __init__(self, blob):
        self.b = memoryview(blob)
        self.pos = 0

    def ternary(self, length):
        """get_ternfilters -> (sign uint8[len] (1 => +1, 0 => -1), zero uint8[len])."""
        tag = self.b[self.pos]
        self.pos += 1
        assert tag in (FMT_BIN, FMT_TERN), "bad filter tag %d" % tag
        nbits = 1 if tag == FMT_BIN else 2
        nbytes = (length * nbits + 7) // 8
        raw = np.frombuffer(self.b[self.pos:self.pos + nbytes], dtype=np.uint8)
        self.pos += nbytes
        bits = np.unpackbits(raw)  # MSB first, as (pack >> (7 - j)) & 1
        if nbits == 1:
            return bits[:length].copy(), np.zeros(length, np.uint8)
        pairs = bits[:2 * length].reshape(length, 2)
        return pairs[:, 0].copy(), pairs[:, 1].copy()

    def ints(self, length):
        """get_intfilters / get_intfilters_ptxt."""
        tag = self.b[self.pos]
        self.pos += 1
        assert tag in (FMT_UINT32, FMT_INT32), "bad int tag %d" % tag
        v = np.frombuffer(self.b[self.pos:self.pos + 4 * length], dtype=np.int32).copy()
        self.pos += 4 * length
        return v

    def done(self):
        return self.pos == len(self.b)


class MnistSignNet:
    """Weights of nets/mnist/sign1024x<K> as plain arrays (layouts of lib/BinFunc.cpp:388,402)."""

    def __init__(self, blob, hidden_layers, hidden=1024, classes=10):
        r = WeightReader(blob)
        self.bias0 = r.ints(1)                          # IntLayer quantize bias (depth 1)
        self.fc = []
        k = 14 * 14
        for _ in range(hidden_layers):
            sign, zero = r.ternary(k * hidden)          # [K][M]: ((fh*fw_+fw)*Cin+di)*Cout+od with 1x1 window
            bias = r.ints(hidden)
            self.fc.append((sign.reshape(k, hidden), zero.reshape(k, hidden), bias))
            k = hidden
        sign, zero = r.ternary(k * classes)
        bias = r.ints(classes)
        self.final = (sign.reshape(k, classes), zero.reshape(k, classes), bias)
        assert r.done(), "trailing bytes in weight file"

    @staticmethod
    def bias_to_torus(b):
        """get_intfilters: modSwitchToTorus32(b, 4096) on the b word of a trivial sample."""
        return (b.astype(np.int64) << 20).astype(np.uint64).astype(np.uint32).view(np.int32)


class EncryptedMnist:
    """Device-resident evaluation on a redsec_amd.Backend whose keys are loaded."""

    def __init__(self, backend, net):
        import torch
        self.be = backend
        self.net = net
        dev = "cuda:%d" % backend.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.bias0 = t(MnistSignNet.bias_to_torus(net.bias0))
        self.fc = [(t(s), t(z), t(MnistSignNet.bias_to_torus(b))) for s, z, b in net.fc]
        s, z, b = net.final
        self.final = (t(s), t(z), t(MnistSignNet.bias_to_torus(b)))
        self.pool = dict(H=28, Wd=28, C=1, win_h=2, win_w=2, stride_h=2, stride_w=2, off_h=0, off_w=0, Ho=14, Wo=14)

    def run(self, image_ct, taps=None):
        """image_ct: int32 CUDA tensor [784][W] (encrypt_image.cpp order: row-major pixels).
        Returns int32 [10][W]. `taps` (dict) receives the intermediate ciphertext tensors."""
        be = self.be
        x = image_ct.view(28, 28, 1, be.W)
        # IntLayer: SumPooling::execute + Quantize::execute (bias folded into the pooling kernel)
        pre0 = be.sumpool(x, self.pool, bias_b=self.bias0).view(196, be.W)
        bits = be.bootstrap(pre0, MU_SIGN)
        if taps is not None:
            taps["pre0"], taps["bits0"] = pre0, bits
        for li, (sign, zero, bias) in enumerate(self.fc):
            pre = be.linear_fc(bits, sign, zero, zero_tap_b=0, bias_b=bias)   # BinFunc: zero taps add nothing
            bits = be.bootstrap(pre, MU_SIGN)
            if taps is not None:
                taps["pre%d" % (li + 1)], taps["bits%d" % (li + 1)] = pre, bits
        sign, zero, bias = self.final
        return be.linear_fc(bits, sign, zero, zero_tap_b=0, bias_b=bias)      # Quantize::add_bias, no bootstrap
