"""numpy restatement of the reference's ENCRYPTED linear stage on LWE words: the CHECKER for rs_conv_ternary_dev,
rs_linear_fc_dev, rs_sumpool_dev and rs_gather_rows_dev at the shapes of nets/cifar/binarynet/net.cpp:114-209.
Test infrastructure only (nothing under redsec_amd/ imports it).

Follows /root/reference/lib/BinFunc.cpp:
  :217-320  Convolution::execute, ENCRYPTED branch: per output (ph, pw, od) a window of partsum_ops samples -- the input
            (filter bit 1) or its negation (filter bit 0: p_inputs_bar = lweClear + lweSubTo, :207-208), lweClear for a
            ternary-zero tap (:262-269) and for an out-of-image tap under same padding (:271-295, "always set to 0") --
            summed by a tree of lweAddTo (:297-310); int32 wrap-around, so any order gives the same words;
  :344-362  retrieve_dims: tap wi -> (di, fh, fw), out of bounds when (fh + ph*stride - offset) falls outside the image;
  :373-402  get_input_i = (h*W + w)*Cin + di; get_filter_i = ((fh*fw_ + fw)*Cin + di)*Cout + od; get_output_i = (ph*Wo + pw)*Cout + od;
  :677-732  SumPooling::execute: windowed lweAddTo; :1056-1071 Quantize::execute: + bias[i % depth] on the b word
            (the bias is a trivial sample, lib/BinOps_enc.cpp:274-297), then the bootstrap (checked elsewhere).
lib/IntFunc.cpp:268,277: the IntFunc variant adds the trivial constant -1/4096 for ternary-zero AND padding taps
(zero_tap_b / pad_tap_b here).

Everything is int64 arithmetic wrapped to 32 bits at the end; `rows(...)` fetches only the ciphertexts a check needs
from a (device) slab, so full-size CIFAR stages cost a few MB each."""
import numpy as np


def wrap32(v):
    return (np.asarray(v, np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def rows(slab, idx):
    """slab: torch tensor or ndarray [R][W]; idx: int sequence -> int64 ndarray [len(idx)][W]."""
    idx = np.asarray(idx, np.int64)
    if isinstance(slab, np.ndarray):
        return slab[idx].astype(np.int64)
    import torch
    return slab[torch.from_numpy(idx).to(slab.device)].cpu().numpy().astype(np.int64)


def ternary_weights(sign, zero):
    """+1 where the filter bit is 1, -1 where 0, 0 where the ternary mask is set (get_ternfilters, lib/BinOps_enc.cpp:247-272)."""
    sign, zero = np.asarray(sign), np.asarray(zero)
    return np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)


def conv_outputs(prev, shape, sign, zero, bias_torus, outs, zero_tap_b=0, pad_tap_b=0):
    """Expected ciphertexts of Convolution::execute at outs = [(ph, pw, od), ...] -> (flat output indices, int32 [len][W]).
    prev: the input slab [H*Wd*Cin][W]; shape: the rs_conv_shape dict; bias_torus: int32[Cout] added on the b word."""
    H, Wd, Cin, Cout = shape["H"], shape["Wd"], shape["Cin"], shape["Cout"]
    fh_, fw_ = shape["fh"], shape["fw"]
    w = ternary_weights(sign, zero).reshape(fh_, fw_, Cin, Cout)
    zmask = np.asarray(zero).reshape(fh_, fw_, Cin, Cout)
    by_pixel = {}
    for ph, pw, od in outs:
        by_pixel.setdefault((ph, pw), []).append(od)
    flat, want = [], []
    for (ph, pw), ods in by_pixel.items():
        taps, inb = [], []
        for fh in range(fh_):
            for fw in range(fw_):
                ih, iw = fh + ph * shape["stride_h"] - shape["off_h"], fw + pw * shape["stride_w"] - shape["off_w"]
                ok = 0 <= ih < H and 0 <= iw < Wd
                inb.append(ok)
                if ok:
                    taps.append(((ih * Wd + iw) * Cin + np.arange(Cin), fh, fw))
        patch = rows(prev, np.concatenate([t[0] for t in taps]))                      # [(in-image taps) * Cin][W]
        for od in ods:
            col = np.concatenate([w[fh, fw, :, od] for _, fh, fw in taps])
            acc = col @ patch
            n_zero_inb = sum(int(zmask[fh, fw, :, od].sum()) for _, fh, fw in taps)
            n_oob = (fh_ * fw_ - len(taps)) * Cin
            n_zero_oob = int(zmask[:, :, :, od].sum()) - n_zero_inb
            # the reference tests the ternary mask before the bounds (":262" before ":271"): a zero tap is a zero tap wherever it lies
            acc[-1] += zero_tap_b * (n_zero_inb + n_zero_oob) + pad_tap_b * (n_oob - n_zero_oob)
            if bias_torus is not None:
                acc[-1] += int(bias_torus[od % len(bias_torus)])
            flat.append((ph * shape["Wo"] + pw) * Cout + od)
            want.append(wrap32(acc))
    return np.array(flat, np.int64), np.stack(want)


def conv_full(x, shape, sign, zero, bias_torus, zero_tap_b=0, pad_tap_b=0):
    """The whole output map [Ho][Wo][Cout][W] of one convolution, by im2col and ONE float64 matrix product: every partial
    sum is below K * 2^31 < 2^53, so the BLAS result is the exact integer (asserted)."""
    H, Wd, Cin, Cout = shape["H"], shape["Wd"], shape["Cin"], shape["Cout"]
    fh_, fw_, Ho, Wo = shape["fh"], shape["fw"], shape["Ho"], shape["Wo"]
    K = fh_ * fw_ * Cin
    assert K * 2.0 ** 31 < 2.0 ** 53
    Wn = x.shape[-1]
    x = np.asarray(x).reshape(H, Wd, Cin, Wn)
    oob = np.zeros((Ho, Wo, fh_, fw_), np.int64)
    src = np.zeros((Ho, Wo, fh_, fw_, 2), np.int64)
    for ph in range(Ho):
        for pw in range(Wo):
            for fh in range(fh_):
                for fw in range(fw_):
                    ih, iw = fh + ph * shape["stride_h"] - shape["off_h"], fw + pw * shape["stride_w"] - shape["off_w"]
                    if 0 <= ih < H and 0 <= iw < Wd:
                        src[ph, pw, fh, fw] = ih, iw
                    else:
                        oob[ph, pw, fh, fw] = 1
    w = ternary_weights(sign, zero).reshape(K, Cout).astype(np.float64)
    out = np.zeros((Ho * Wo, Cout, Wn), np.int64)
    for w0 in range(0, Wn, 64):                                                       # word chunks bound the im2col buffer
        xs = x[..., w0:w0 + 64].astype(np.float64)
        cols = xs[src[..., 0], src[..., 1]] * (1 - oob)[..., None, None]              # [Ho][Wo][fh][fw][Cin][chunk]
        # out[p, od, :] = sum_k w[k, od] * cols[p, k, :]
        part = np.einsum("ko,pkw->pow", w, cols.reshape(Ho * Wo, K, -1), optimize=True)
        out[:, :, w0:w0 + 64] = np.rint(part).astype(np.int64)
    z = np.asarray(zero).reshape(fh_, fw_, Cin, Cout).astype(np.int64)
    n_zero = z.sum(axis=(0, 1, 2))                                                    # [Cout]
    z_tap = z.sum(axis=2)                                                             # [fh][fw][Cout]
    n_zero_oob = np.einsum("pqab,abo->pqo", oob, z_tap).reshape(Ho * Wo, Cout)
    n_oob = oob.sum(axis=(2, 3)).reshape(Ho * Wo, 1) * Cin
    out[:, :, -1] += zero_tap_b * n_zero[None, :] + pad_tap_b * (n_oob - n_zero_oob)
    if bias_torus is not None:
        out[:, :, -1] += np.asarray(bias_torus, np.int64)[np.arange(Cout) % len(bias_torus)][None, :]
    return wrap32(out).reshape(Ho, Wo, Cout, Wn)


def fc_outputs(prev, sign, zero, bias_torus, ms, zero_tap_b=0):
    """Fully-connected form (1x1 window over a 1x1xK map, lib/BinLayer.cpp E_FC): out[m] = sum_k w[k][m] * in[k] + bias[m]."""
    w = ternary_weights(sign, zero)
    K = w.shape[0]
    x = rows(prev, np.arange(K))
    want = []
    for m in ms:
        acc = w[:, m] @ x
        acc[-1] += zero_tap_b * int(np.asarray(zero)[:, m].sum())
        if bias_torus is not None:
            acc[-1] += int(bias_torus[m % len(bias_torus)])
        want.append(wrap32(acc))
    return np.array(list(ms), np.int64), np.stack(want)


def sumpool_outputs(prev, shape, bias_torus, outs):
    """SumPooling::execute (+ the constant of the stage behind it) at outs = [(oh, ow, c), ...]; taps outside the image are skipped."""
    H, Wd, C = shape["H"], shape["Wd"], shape["C"]
    flat, want = [], []
    for oh, ow, c in outs:
        idx = []
        for a in range(shape["win_h"]):
            for b in range(shape["win_w"]):
                ih, iw = oh * shape["stride_h"] - shape["off_h"] + a, ow * shape["stride_w"] - shape["off_w"] + b
                if 0 <= ih < H and 0 <= iw < Wd:
                    idx.append((ih * Wd + iw) * C + c)
        acc = rows(prev, idx).sum(axis=0)
        if bias_torus is not None:
            acc[-1] += int(bias_torus[c % len(bias_torus)])
        flat.append((oh * shape["Wo"] + ow) * C + c)
        want.append(wrap32(acc))
    return np.array(flat, np.int64), np.stack(want)


def spread_outputs(Ho, Wo, Cout, rng, extra=24):
    """(ph, pw, od) triples that exercise a convolution's index math: the four corners, the four edge midpoints, two
    interior pixels -- each with the first and last channel, the channels either side of every 32-channel tile boundary
    the kernel's register tiling has -- and `extra` seeded triples anywhere."""
    px = {(0, 0), (0, Wo - 1), (Ho - 1, 0), (Ho - 1, Wo - 1), (0, Wo // 2), (Ho - 1, Wo // 2), (Ho // 2, 0), (Ho // 2, Wo - 1),
          (Ho // 2, Wo // 2), (1, Wo - 2)}
    ch = {0, Cout - 1} | {c for c in (31, 32, 33, Cout // 2 - 1, Cout // 2, Cout - 32, Cout - 33) if 0 <= c < Cout}
    outs = {(ph, pw, od) for (ph, pw) in px for od in ch}
    while len(outs) < len(px) * len(ch) + extra:
        outs.add((int(rng.integers(Ho)), int(rng.integers(Wo)), int(rng.integers(Cout))))
    return sorted(outs)
