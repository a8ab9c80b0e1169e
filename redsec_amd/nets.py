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

    def run(self, image_ct, taps=None, shard=False):
        """image_ct: int32 CUDA tensor [784][W] (encrypt_image.cpp order: row-major pixels).
        Returns int32 [10][W]. `taps` (dict) receives the intermediate ciphertext tensors.
        shard=True (inside an initialised torch.distributed job, one process per GPU, keys loaded on
        every rank): gate-parallel evaluation of ONE image -- every rank bootstraps a contiguous slice
        of each stage's ciphertexts and the slices are all-gathered before the next linear stage, which
        needs the whole bit vector (partitioning 2). The linear stages are
        recomputed on every rank (< 2 % of the image). Results are identical to the unsharded run word
        for word: a bootstrap's output depends on its own input ciphertext only."""
        be = self.be
        if shard:
            from . import sharding
            boot = lambda pre: sharding.sharded_stage(lambda rows: be.bootstrap(rows, MU_SIGN), pre)
        else:
            boot = lambda pre: be.bootstrap(pre, MU_SIGN)
        x = image_ct.view(28, 28, 1, be.W)
        # IntLayer: SumPooling::execute + Quantize::execute (bias folded into the pooling kernel)
        pre0 = be.sumpool(x, self.pool, bias_b=self.bias0).view(196, be.W)
        bits = boot(pre0)
        if taps is not None:
            taps["pre0"], taps["bits0"] = pre0, bits
        for li, (sign, zero, bias) in enumerate(self.fc):
            pre = be.linear_fc(bits, sign, zero, zero_tap_b=0, bias_b=bias)   # BinFunc: zero taps add nothing
            bits = boot(pre)
            if taps is not None:
                taps["pre%d" % (li + 1)], taps["bits%d" % (li + 1)] = pre, bits
        sign, zero, bias = self.final
        return be.linear_fc(bits, sign, zero, zero_tap_b=0, bias_b=bias)      # Quantize::add_bias, no bootstrap


class EncryptedCifar:
    """nets/cifar/binarynet{,_small}/net.cpp:96-209 on a redsec_amd.Backend, device-resident:
    IntLayer(NO_CONV, SIGN); 6 x BinLayer(CONV 3x3 same, SIGN), a 2x2 max-pool after every second one;
    2 x BinLayer(FC, SIGN); BinLayer(FC_FINAL). Same stage order and the same max-pool semantics as the
    C++ layer mirror (redsec_amd/host/layers.cpp, DESIGN.md "Max-pool semantics"). maxpool="fused" (the
    default): the sign bootstrap ahead of a max-pool emits +-1/16 and ONE bootstrap of the windowed sum
    + 3/16 is the OR of the 2x2 window; maxpool="chain": the bits are emitted as +-1/8 and OR-ed by
    bootsOR gates in (fh, fw) order starting from a copy of the first tap, the last OR re-encoding to
    +-1/4096.

    `net`: weights as plain arrays -- bias0 int32[3]; convs [(sign, zero uint8[3][3][Cin][Cout], bias
    int32[Cout])]; fcs [(sign, zero uint8[K][M], bias int32[M])], the last one being the logits layer
    (layouts of lib/BinFunc.cpp:388). shard=True: gate-parallel over the ranks of an initialised
    torch.distributed job as in EncryptedMnist.run (693,248 bootstraps per binarynet image, the
    largest all_gather 131,072 x 351 words)."""

    MU8 = 1 << 29   # modSwitchToTorus32(1, 8): the encoding bootsOR assumes

    def __init__(self, backend, net, maxpool="fused"):
        import torch
        assert maxpool in ("fused", "chain")
        self.be = backend
        self.maxpool = maxpool
        dev = "cuda:%d" % backend.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        tor = MnistSignNet.bias_to_torus
        self.bias0 = t(tor(net.bias0))
        self.pool_bias = t(np.array([3 << 28], np.int64).astype(np.uint32).view(np.int32))   # (w - 1) / (4 w), w = 4
        self.convs = [(t(s), t(z), t(tor(b))) for s, z, b in net.convs]
        self.fcs = [(t(s), t(z), t(tor(b))) for s, z, b in net.fcs]
        self._pool_index = {}
        self._t = t

    def _pool(self, H, Wd, C):
        """[tap][out] input rows of a 2x2 stride-2 window over [H][Wd][C] (MaxPooling::prep index math)."""
        key = (H, Wd, C)
        if key not in self._pool_index:
            Ho, Wo = H // 2, Wd // 2
            oh, ow, c = np.meshgrid(np.arange(Ho), np.arange(Wo), np.arange(C), indexing="ij")
            taps = [((2 * oh + fh) * Wd + (2 * ow + fw)) * C + c for fh in range(2) for fw in range(2)]
            self._pool_index[key] = self._t(np.stack([a.reshape(-1) for a in taps]).astype(np.int32))
        return self._pool_index[key]

    def run(self, image_ct, shard=False):
        """image_ct: int32 CUDA tensor [32*32*3][W] in (row, column, channel) order. Returns int32 [10][W]."""
        be = self.be
        if shard:
            from . import sharding
            stage = lambda fn, *xs: sharding.sharded_stage(lambda rows: fn(*[rows[k] for k in range(len(xs))]), _Rows(xs))
        else:
            stage = lambda fn, *xs: fn(*xs)
        W = be.W
        H = Wd = 32
        one = dict(H=H, Wd=Wd, C=3, win_h=1, win_w=1, stride_h=1, stride_w=1, off_h=0, off_w=0, Ho=H, Wo=Wd)
        pre = be.sumpool(image_ct.view(H, Wd, 3, W), one, bias_b=self.bias0).view(-1, W)     # Quantize: x + bias[i % depth]
        bits = stage(lambda r: be.bootstrap(r, MU_SIGN), pre)
        C = 3
        for li, (sign, zero, bias) in enumerate(self.convs):
            Cout = sign.shape[3]
            shape = dict(H=H, Wd=Wd, Cin=C, Cout=Cout, fh=3, fw=3, stride_h=1, stride_w=1, off_h=1, off_w=1, Ho=H, Wo=Wd)
            pre = be.conv_ternary(bits.view(H, Wd, C, W), sign, zero, shape, zero_tap_b=0, pad_tap_b=0, bias_b=bias).view(-1, W)
            C = Cout
            pooled = li % 2 == 1
            fused = pooled and self.maxpool == "fused"
            mu = MU_SIGN if not pooled else ((1 << 28) if fused else self.MU8)
            bits = stage(lambda r: be.bootstrap(r, mu), pre)
            if fused:
                win = dict(H=H, Wd=Wd, C=C, win_h=2, win_w=2, stride_h=2, stride_w=2, off_h=0, off_w=0, Ho=H // 2, Wo=Wd // 2)
                pre = be.sumpool(bits.view(H, Wd, C, W), win, bias_b=self.pool_bias).view(-1, W)
                bits = stage(lambda r: be.bootstrap(r, MU_SIGN), pre)
                H //= 2; Wd //= 2
            elif pooled:
                idx = self._pool(H, Wd, C)
                acc = be.gather_rows(bits, idx[0])
                for tp in range(1, 4):
                    tap = be.gather_rows(bits, idx[tp])
                    mu = MU_SIGN if tp == 3 else self.MU8
                    acc = stage(lambda a, b: be.gate_mu("OR", a, b, mu), acc, tap)
                bits = acc
                H //= 2; Wd //= 2
        v = bits
        for i, (sign, zero, bias) in enumerate(self.fcs):
            pre = be.linear_fc(v, sign, zero, zero_tap_b=0, bias_b=bias)
            if i == len(self.fcs) - 1:
                return pre
            v = stage(lambda r: be.bootstrap(r, MU_SIGN), pre)


class _Rows:
    """Several equally long row batches sliced together (the operands of a two-input gate stage)."""

    def __init__(self, xs):
        self.xs = xs
        self.shape = xs[0].shape

    def __getitem__(self, sl):
        return _Rows([x[sl] for x in self.xs]) if isinstance(sl, slice) else self.xs[sl]

    def contiguous(self):
        return _Rows([x.contiguous() for x in self.xs])
