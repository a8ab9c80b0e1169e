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

    def run_many(self, image_cts):
        """K images at once: int32 CUDA tensor [K][784][W] -> [K][10][W]. Every bootstrap stage is ONE launch over the
        neurons of all K images (the gates of different images are as independent as the gates of one: the batch moves
        from the latency forms into the throughput form), the linear stages run per image. Equal word for word to K
        calls of run()."""
        import torch
        be = self.be
        K = image_cts.shape[0]
        pre = torch.cat([be.sumpool(image_cts[k].view(28, 28, 1, be.W), self.pool, bias_b=self.bias0).view(196, be.W) for k in range(K)])
        bits = be.bootstrap(pre, MU_SIGN)
        for sign, zero, bias in self.fc:
            width = bits.shape[0] // K
            pre = torch.cat([be.linear_fc(bits[k * width:(k + 1) * width], sign, zero, zero_tap_b=0, bias_b=bias) for k in range(K)])
            bits = be.bootstrap(pre, MU_SIGN)
        sign, zero, bias = self.final
        width = bits.shape[0] // K
        return torch.stack([be.linear_fc(bits[k * width:(k + 1) * width], sign, zero, zero_tap_b=0, bias_b=bias) for k in range(K)])


# ---- ReLU networks (nets/mnist/relu1024x{1,2,3}) -------------------------------------------------------
# Quantize::relu_shift (lib/IntFunc.cpp:934-973), corrected semantics -- DESIGN.md "ReLU semantics".
# The reference's ENCRYPTED branch is not a function of the layer input (shared scratch across OpenMP
# threads, accumulation into an uncleared ciphertext, +-1/4096 operands fed to a +-1/8 MUX, and
# slope*x + bias wrapping around the torus); its plaintext branch is: x = slope*pre + bias;
# y = x >> slope_bits; out = 0 if x < 0 else min(y, 2^shift_bits - 1). That staircase is evaluated here as
# ONE programmable bootstrap per neuron (rs_bootstrap_lut_dev) whose test polynomial tabulates it over
# the mod-switched phase of pre, with two encoding choices the reference leaves open:
#   * the phase is shifted by 1/4 so that pre in [-N/2, N/2) LUT steps lands on [0, 1/2), where a
#     negacyclic test polynomial is unconstrained;
#   * ReLU outputs are emitted in units of RELU_UNIT = 2^-14 (not 2^-12): the next layer sums up to 1,024 of
#     them (|pre| up to 2,690 over the 100 bundled images, 15,360 worst case) and must stay inside a
#     quarter turn. Each layer therefore carries its input unit; biases are added in that unit.
SLOPE_BITS_INT = 8            # lib/IntFunc.cpp:45
UNIT_4096 = 1 << 20           # one integer step of a client-encrypted / sign-layer value: 1/4096
RELU_UNIT = 1 << 18           # one integer step of a ReLU output: 1/16384
QUARTER = 1 << 30
LUT_STEP = 1 << 21            # torus32 width of one mod-switched phase step (2^32 / 2N, N = 1024)


def relu_luts(slope, bias, slope_bits, shift_bits, unit_in, unit_out, N=1024):
    """Test polynomials of a ReLU layer: int32 [len(slope)][N]. Index t <-> pre = (t - N/2) * (LUT_STEP / unit_in)
    (the centre of the phase bucket the mod-switch rounds to, after the quarter-turn shift)."""
    upi = LUT_STEP // unit_in
    assert upi * unit_in == LUT_STEP and upi >= 1
    pre = (np.arange(N, dtype=np.int64) - N // 2) * upi
    x = slope.astype(np.int64)[:, None] * pre[None, :] + bias.astype(np.int64)[:, None]
    y = np.where(x < 0, 0, np.minimum(x >> slope_bits, (1 << shift_bits) - 1))       # IntOps::shift + IntOps::relu
    return np.ascontiguousarray((y * unit_out).astype(np.uint64).astype(np.uint32).view(np.int32))


def relu_slope_bits(scale, shift_bits):
    """IntFunc::Quantize::prep, lib/IntFunc.cpp:812-815: SLOPE_BITS + ceil(log2(scale)) - shift_bits."""
    sc_b = 0
    while (1 << sc_b) < scale:
        sc_b += 1
    return SLOPE_BITS_INT + sc_b - shift_bits


class MnistReluNet:
    """Weights of nets/mnist/relu1024x<K>: IntLayer(NO_CONV, SUMPOOL, NONE); K x IntLayer(FC 1024, RELU
    shift_bits 4); IntLayer(FC 10, NONE) (net.cpp:125-167). Records: bias0[1]; per hidden layer ternary
    [K][1024], bias[1024], slope[1024]; final ternary [1024][10], bias[10]."""

    SHIFT_BITS = 4

    def __init__(self, blob, hidden_layers, hidden=1024, classes=10):
        r = WeightReader(blob)
        self.bias0 = r.ints(1)
        self.fc = []
        k = 14 * 14
        for _ in range(hidden_layers):
            sign, zero = r.ternary(k * hidden)
            bias = r.ints(hidden)
            slope = r.ints(hidden)
            self.fc.append((sign.reshape(k, hidden), zero.reshape(k, hidden), bias, slope))
            k = hidden
        sign, zero = r.ternary(k * classes)
        self.final = (sign.reshape(k, classes), zero.reshape(k, classes), r.ints(classes))
        assert r.done(), "trailing bytes in weight file"

    @staticmethod
    def neg_taps(sign, zero):
        """IntFunc::Convolution's plaintext branch multiplies by -1 as the one's complement ~x = -x - 1
        (IntOps::invert, lib/IntOps.cpp): every NEGATIVE tap also contributes -1. The trained biases fold
        that constant, so the encrypted chain adds it too (the ENCRYPTED branch's own constant, -1/4096 per
        ZERO tap at lib/IntFunc.cpp:268,277, classifies 5 % of relu1024x2's bundled images; this one 100 %)."""
        return ((zero == 0) & (sign == 0)).sum(axis=0).astype(np.int64)

    # what the network hands back: logits in the unit the last layer summed them in (1/16384 behind a ReLU). The reference's
    # client decodes with message space 4096 and so reads round(logit / 4); multiplying by 4 instead (layers.cpp,
    # REDSEC_RESCALE_LOGITS=1) would wrap around that client's +-2048 range on relu1024x3, whose logits reach +-2000.
    LOGIT_UNIT = RELU_UNIT

    def stages(self):
        """Per hidden layer: (slope_bits, unit_in, unit_out), and the unit the final layer SUMS in; the scale chain of
        IntFunc::*::prep."""
        out = []
        scale, unit = 4.0, UNIT_4096          # input scale 1 x 2x2 sum-pool (lib/IntFunc.cpp:629)
        for _ in self.fc:
            out.append((relu_slope_bits(scale, self.SHIFT_BITS), unit, RELU_UNIT))
            scale, unit = float((1 << self.SHIFT_BITS) - 1), RELU_UNIT
        return out, unit


def relu_preprocess(pixels):
    """nets/mnist/relu1024x1/main.cpp:203: v / 100 - 1 (0..99 -> -1, 100..199 -> 0, 200..255 -> 1)."""
    return np.asarray(pixels, dtype=np.int64) // 100 - 1


class EncryptedMnistRelu:
    """relu1024x<K> on a redsec_amd.Backend, device-resident. image_ct: [784][W] encryptions of
    relu_preprocess(pixels) / 4096. Returns the 10 logit ciphertexts in units of `self.logit_unit`."""

    def __init__(self, backend, net):
        import torch
        self.be = backend
        dev = "cuda:%d" % backend.device
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        w32 = lambda a: np.asarray(a, np.int64).astype(np.uint64).astype(np.uint32).view(np.int32)
        self.bias0 = t(w32(net.bias0.astype(np.int64) * UNIT_4096))
        self.pool = dict(H=28, Wd=28, C=1, win_h=2, win_w=2, stride_h=2, stride_w=2, off_h=0, off_w=0, Ho=14, Wo=14)
        stages, self.sum_unit = net.stages()
        self.logit_unit = net.LOGIT_UNIT
        self.fc = []
        for (sign, zero, bias, slope), (sb, u_in, u_out) in zip(net.fc, stages):
            lin_bias = w32(QUARTER - net.neg_taps(sign, zero) * u_in)          # quarter-turn shift + the -1 per negative tap
            self.fc.append((t(sign), t(zero), t(lin_bias), t(relu_luts(slope, bias, sb, net.SHIFT_BITS, u_in, u_out))))
        sign, zero, bias = net.final
        self.final = (t(sign), t(zero), t(w32((bias.astype(np.int64) - net.neg_taps(sign, zero)) * self.sum_unit)))

    def run(self, image_ct, taps=None):
        be = self.be
        v = be.sumpool(image_ct.view(28, 28, 1, be.W), self.pool, bias_b=self.bias0).view(196, be.W)   # Quantize::add_bias
        if taps is not None:
            taps["in0"] = v
        for li, (sign, zero, lin_bias, luts) in enumerate(self.fc):
            pre = be.linear_fc(v, sign, zero, zero_tap_b=0, bias_b=lin_bias)
            v = be.bootstrap_lut(pre, luts)
            if taps is not None:
                taps["pre%d" % (li + 1)], taps["act%d" % (li + 1)] = pre, v
        sign, zero, bias = self.final
        out = be.linear_fc(v, sign, zero, zero_tap_b=0, bias_b=bias)
        k = self.logit_unit // self.sum_unit
        return out if k == 1 else be.lincomb(out, k)      # 1/16384 -> 1/4096 steps: the same integers, exactly


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

    def run(self, image_ct, shard=False, taps=None):
        """image_ct: int32 CUDA tensor [32*32*3][W] in (row, column, channel) order. Returns int32 [10][W].
        `taps` (list) receives one record per BOOTSTRAPPED stage, in execution order: dict(name, kind = "sign" | "or",
        mu, inputs = the stage's input slab(s) [B][W], out = its output slab) -- what the stage-level parity test
        (tests/test_gpu_cifar.py) compares with the oracle row by row."""
        be = self.be
        if shard:
            from . import sharding
            run_stage = lambda fn, *xs: sharding.sharded_stage(lambda rows: fn(*[rows[k] for k in range(len(xs))]), _Rows(xs))
        else:
            run_stage = lambda fn, *xs: fn(*xs)

        def stage(fn, *xs, name=None, kind="sign", mu=MU_SIGN):
            out = run_stage(fn, *xs)
            if taps is not None:
                taps.append(dict(name=name, kind=kind, mu=int(mu), inputs=xs, out=out))
            return out
        W = be.W
        H = Wd = 32
        one = dict(H=H, Wd=Wd, C=3, win_h=1, win_w=1, stride_h=1, stride_w=1, off_h=0, off_w=0, Ho=H, Wo=Wd)
        pre = be.sumpool(image_ct.view(H, Wd, 3, W), one, bias_b=self.bias0).view(-1, W)     # Quantize: x + bias[i % depth]
        bits = stage(lambda r: be.bootstrap(r, MU_SIGN), pre, name="quantize0")
        C = 3
        for li, (sign, zero, bias) in enumerate(self.convs):
            Cout = sign.shape[3]
            shape = dict(H=H, Wd=Wd, Cin=C, Cout=Cout, fh=3, fw=3, stride_h=1, stride_w=1, off_h=1, off_w=1, Ho=H, Wo=Wd)
            pre = be.conv_ternary(bits.view(H, Wd, C, W), sign, zero, shape, zero_tap_b=0, pad_tap_b=0, bias_b=bias).view(-1, W)
            C = Cout
            pooled = li % 2 == 1
            fused = pooled and self.maxpool == "fused"
            mu = MU_SIGN if not pooled else ((1 << 28) if fused else self.MU8)
            bits = stage(lambda r: be.bootstrap(r, mu), pre, name="conv%d" % (li + 1), mu=mu)
            if fused:
                win = dict(H=H, Wd=Wd, C=C, win_h=2, win_w=2, stride_h=2, stride_w=2, off_h=0, off_w=0, Ho=H // 2, Wo=Wd // 2)
                pre = be.sumpool(bits.view(H, Wd, C, W), win, bias_b=self.pool_bias).view(-1, W)
                bits = stage(lambda r: be.bootstrap(r, MU_SIGN), pre, name="maxpool%d" % (li + 1))
                H //= 2; Wd //= 2
            elif pooled:
                idx = self._pool(H, Wd, C)
                acc = be.gather_rows(bits, idx[0])
                for tp in range(1, 4):
                    tap = be.gather_rows(bits, idx[tp])
                    mu = MU_SIGN if tp == 3 else self.MU8
                    acc = stage(lambda a, b: be.gate_mu("OR", a, b, mu), acc, tap, name="maxpool%d_or%d" % (li + 1, tp), kind="or", mu=mu)
                bits = acc
                H //= 2; Wd //= 2
        v = bits
        for i, (sign, zero, bias) in enumerate(self.fcs):
            pre = be.linear_fc(v, sign, zero, zero_tap_b=0, bias_b=bias)
            if i == len(self.fcs) - 1:
                return pre
            v = stage(lambda r: be.bootstrap(r, MU_SIGN), pre, name="fc%d" % (i + 1))


class _Rows:
    """Several equally long row batches sliced together (the operands of a two-input gate stage)."""

    def __init__(self, xs):
        self.xs = xs
        self.shape = xs[0].shape

    def __getitem__(self, sl):
        return _Rows([x[sl] for x in self.xs]) if isinstance(sl, slice) else self.xs[sl]

    def contiguous(self):
        return _Rows([x.contiguous() for x in self.xs])
