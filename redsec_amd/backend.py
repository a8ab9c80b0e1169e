"""ctypes binding of libredsec_hip.so (the C ABI in include/redsec_hip.h).

PyTorch is used only as plumbing: device buffers are int32 CUDA (HIP) tensors and launches go on
torch's current stream. All arithmetic happens in the HIP kernels; there is no CPU or eager
fallback -- if the library or a gfx950 device is missing, calls raise RedsecHipError.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

# every symbol include/redsec_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "rs_last_error", "rs_version", "rs_params_default128", "rs_params_redsec_small_v2", "rs_create", "rs_destroy",
    "rs_load_keys", "rs_reserve", "rs_bootstrap_dev", "rs_bootstrap", "rs_gate_dev", "rs_gate", "rs_mux_dev", "rs_mux",
    "rs_gate_mu_dev", "rs_gather_rows_dev", "rs_bootstrap_wo_ks_dev", "rs_keyswitch_dev", "rs_debug_polymul", "rs_debug_cohort_table", "rs_debug_fp64_rate", "rs_linear_fc_dev", "rs_conv_ternary_dev",
    "rs_sumpool_dev", "rs_lincomb_dev", "rs_dev_alloc", "rs_dev_free", "rs_copy_to_dev", "rs_copy_to_host", "rs_sync",
    "rs_set_timing", "rs_last_kernel_ms", "rs_info", "rs_set_mode", "rs_get_mode", "rs_rounding_certificate", "rs_fft_fallbacks",
    "rs_bootstrap_lut_dev", "rs_set_certificate_limit", "rs_certify", "rs_reserve_stream", "rs_last_kernel_ms_stream", "rs_last_launch", "rs_copy_dev_to_dev",
    "rs_params_redsec_small", "rs_params_redsec_medium", "rs_params_redsec_large", "rs_split_bound",
    "rs_allgather_rows", "rs_release_stream", "rs_load_synthetic_keys",
]

GATES = {"NAND": 0, "OR": 1, "AND": 2, "NOR": 3, "XOR": 4, "XNOR": 5, "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9}


class RedsecHipError(RuntimeError):
    pass


class RsParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("k", C.c_int32), ("bk_l", C.c_int32), ("bk_Bgbit", C.c_int32),
                ("ks_t", C.c_int32), ("ks_basebit", C.c_int32)]


class RsConvShape(C.Structure):
    _fields_ = [(f, C.c_int32) for f in
                ("H", "Wd", "Cin", "Cout", "fh", "fw", "stride_h", "stride_w", "off_h", "off_w", "Ho", "Wo")]


class RsPoolShape(C.Structure):
    _fields_ = [(f, C.c_int32) for f in
                ("H", "Wd", "C", "win_h", "win_w", "stride_h", "stride_w", "off_h", "off_w", "Ho", "Wo")]


_lib = None


def load_library(path=None):
    """dlopen libredsec_hip.so (building it in-tree first if the sources are newer)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("REDSEC_HIP_LIB")   # timing experiments may point at a variant build
    so = path or _build.build_hip()
    # torch's HIP runtime must be initialised BEFORE this library's code object registers with its own copy of the runtime in the
    # same process: the other order leaves one of the two without devices ("No HIP GPUs are available" / RS_ERR_NO_DEVICE;
    # measured on the MI355X boxes, ROCm 7.2 + torch 2.10). Loading is the first thing every user of the package does.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    if not os.path.exists(so):
        raise RedsecHipError("libredsec_hip.so is missing: run `python -m redsec_amd.build`")
    try:
        L = C.CDLL(so)
    except OSError as e:  # pragma: no cover
        raise RedsecHipError("cannot load %s: %s" % (so, e))
    L.rs_last_error.restype = C.c_char_p
    L.rs_version.restype = C.c_char_p
    P = C.POINTER(RsParams)
    vp = C.c_void_p
    L.rs_params_default128.argtypes = [P]
    L.rs_params_redsec_small_v2.argtypes = [P]
    L.rs_params_redsec_small.argtypes = [P]
    L.rs_params_redsec_medium.argtypes = [P]
    L.rs_params_redsec_large.argtypes = [P]
    L.rs_split_bound.argtypes = [vp, C.POINTER(C.c_double)]
    L.rs_create.argtypes = [C.POINTER(vp), P, C.c_int]
    L.rs_destroy.argtypes = [vp]
    L.rs_load_keys.argtypes = [vp, _i32p, _i32p]
    L.rs_load_synthetic_keys.argtypes = [vp, C.c_uint64]
    L.rs_reserve.argtypes = [vp, C.c_size_t]
    L.rs_bootstrap_dev.argtypes = [vp, vp, vp, C.c_int32, C.c_size_t, vp]
    L.rs_bootstrap.argtypes = [vp, _i32p, _i32p, C.c_int32, C.c_size_t]
    L.rs_gate_dev.argtypes = [vp, C.c_int, vp, vp, vp, C.c_size_t, vp]
    L.rs_gate.argtypes = [vp, C.c_int, _i32p, _i32p, _i32p, C.c_size_t]
    L.rs_gate_mu_dev.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int32, C.c_size_t, vp]
    L.rs_gather_rows_dev.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    L.rs_mux_dev.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.rs_mux.argtypes = [vp, _i32p, _i32p, _i32p, _i32p, C.c_size_t]
    L.rs_bootstrap_wo_ks_dev.argtypes = [vp, vp, vp, C.c_int32, C.c_size_t, vp]
    L.rs_keyswitch_dev.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.rs_debug_polymul.argtypes = [vp, _i32p, _i32p, _i32p, C.c_size_t]
    L.rs_debug_cohort_table.argtypes = [vp, vp, _i32p]
    L.rs_debug_fp64_rate.argtypes = [vp, C.POINTER(C.c_double)]
    L.rs_linear_fc_dev.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp]
    L.rs_conv_ternary_dev.argtypes = [vp, vp, vp, vp, vp, C.POINTER(RsConvShape), C.c_int32, C.c_int32, vp, C.c_int32, vp]
    L.rs_sumpool_dev.argtypes = [vp, vp, vp, C.POINTER(RsPoolShape), vp, C.c_int32, vp]
    L.rs_lincomb_dev.argtypes = [vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, C.c_size_t, vp]
    L.rs_dev_alloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.rs_dev_free.argtypes = [vp, vp]
    L.rs_copy_to_dev.argtypes = [vp, vp, vp, C.c_size_t]
    L.rs_copy_to_host.argtypes = [vp, vp, vp, C.c_size_t]
    L.rs_sync.argtypes = [vp]
    L.rs_set_timing.argtypes = [vp, C.c_int]
    L.rs_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rs_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rs_set_mode.argtypes = [vp, C.c_int]
    L.rs_get_mode.argtypes = [vp, C.POINTER(C.c_int)]
    L.rs_rounding_certificate.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    L.rs_fft_fallbacks.argtypes = [vp, C.POINTER(C.c_int64)]
    L.rs_bootstrap_lut_dev.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    L.rs_copy_dev_to_dev.argtypes = [vp, vp, vp, vp, C.c_size_t]
    L.rs_allgather_rows.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp), C.c_size_t, C.c_size_t]
    L.rs_release_stream.argtypes = [vp, vp]
    L.rs_set_certificate_limit.argtypes = [vp, C.c_double]
    L.rs_certify.argtypes = [vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.rs_reserve_stream.argtypes = [vp, C.c_size_t, vp]
    L.rs_last_kernel_ms_stream.argtypes = [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rs_last_launch.argtypes = [vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    if path is None:
        _lib = L
    return L


def _check(L, rc):
    if rc != 0:
        raise RedsecHipError("redsec_hip error %d: %s" % (rc, (L.rs_last_error() or b"").decode()))


def params(name, n=None):
    """'default128' | 'redsec_small_v2' | 'redsec_small' | 'redsec_medium' | 'redsec_large'; `n` overrides the LWE
    dimension (reduced-size test keys)."""
    L = load_library()
    p = RsParams()
    if name == "default128":
        _check(L, L.rs_params_default128(C.byref(p)))
    elif name == "redsec_small_v2":
        _check(L, L.rs_params_redsec_small_v2(C.byref(p)))
    elif name in ("redsec_small", "redsec_medium", "redsec_large"):
        _check(L, getattr(L, "rs_params_" + name)(C.byref(p)))
    else:
        raise KeyError(name)
    if n is not None:
        p.n = int(n)
    return p


def _np_i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


class Backend:
    """One context = one GPU + one evaluation key (mirrors TFheGateBootstrappingCloudKeySet usage,
    /root/reference/lib/BinOps_enc.cpp: every primitive takes `bk` as its last argument)."""

    def __init__(self, p, device=0):
        # torch's HIP runtime must be initialised BEFORE this library initialises its own in the same process: the
        # other order leaves torch with "No HIP GPUs are available" (measured on the MI355X boxes, ROCm 7.2 + torch 2.10)
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        self.L = load_library()
        self.p = p
        self.W = p.n + 1
        self.device = device
        h = C.c_void_p()
        _check(self.L, self.L.rs_create(C.byref(h), C.byref(p), device))
        self.h = h

    @property
    def closed(self):
        """True once rs_destroy has run: every other method would hand the library a null context."""
        return not getattr(self, "h", None)

    def close(self, check=None):
        """rs_destroy; raises if the enforced split-mode certificate of some stream's last call had failed (nothing else would
        look at it any more) -- unless another exception is already on its way out (a close() in a `finally:` block must not
        mask the error that brought the caller there) or check=False says the caller has dealt with the context's state. A
        failure that is not raised is never dropped: it is reported as a RuntimeWarning carrying rs_last_error()."""
        if getattr(self, "h", None):
            h, self.h = self.h, None
            rc = self.L.rs_destroy(h)
            if check is None:
                import sys
                check = sys.exc_info()[0] is None
            if check:
                _check(self.L, rc)
            elif rc != 0:
                import warnings
                warnings.warn("rs_destroy reported error %d while another error was being handled: %s"
                              % (rc, (self.L.rs_last_error() or b"").decode()), RuntimeWarning, stacklevel=2)

    def __del__(self):
        try:
            self.close(check=False)
        except Exception:
            pass

    # ---- keys ----
    def load_keys(self, bk, ksk):
        bk, pbk = _np_i32(bk)
        ksk, pksk = _np_i32(ksk)
        p = self.p
        assert bk.size == p.n * 2 * p.bk_l * 2 * p.N, "bk has the wrong size"
        assert ksk.size == p.N * p.ks_t * (1 << p.ks_basebit) * (p.n + 1), "ksk has the wrong size"
        _check(self.L, self.L.rs_load_keys(self.h, pbk, pksk))

    def load_synthetic_keys(self, seed):
        """A key of pseudo-random words generated ON THE DEVICE (rs_load_synthetic_keys; client.synthetic_key_words restates the
        generator): benchmarks and parity tests of the large rings, whose real keys are gigabytes on the host."""
        _check(self.L, self.L.rs_load_synthetic_keys(self.h, C.c_uint64(int(seed) & (2**64 - 1))))

    def release_stream(self, stream):
        _check(self.L, self.L.rs_release_stream(self.h, C.c_void_p(int(stream))))

    def reserve(self, max_batch):
        """Pre-size the workspace of torch's current stream."""
        _check(self.L, self.L.rs_reserve_stream(self.h, int(max_batch), self._stream()))

    # ---- torch plumbing ----
    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _ck_dev(self, t, cols=None):
        import torch
        assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), "need contiguous int32 CUDA tensors"
        assert t.device.index == self.device, "tensor is on another device than the context"
        if cols is not None:
            assert t.shape[-1] == cols, "last dimension must be %d words" % cols
        return C.c_void_p(t.data_ptr())

    def empty(self, *shape):
        import torch
        return torch.empty(*shape, dtype=torch.int32, device="cuda:%d" % self.device)

    # ---- bootstraps on device tensors [B][W] ----
    def bootstrap(self, x, mu, out=None):
        B = x.shape[0]
        out = self.empty(B, self.W) if out is None else out
        _check(self.L, self.L.rs_bootstrap_dev(self.h, self._ck_dev(out, self.W), self._ck_dev(x, self.W), int(mu), B, self._stream()))
        return out

    def bootstrap_lut(self, x, lut, out=None, first=0):
        """Programmable bootstrap: ciphertext b uses the test polynomial lut[(first + b) % len(lut)] (int32 CUDA [L][N])."""
        B = x.shape[0]
        out = self.empty(B, self.W) if out is None else out
        _check(self.L, self.L.rs_bootstrap_lut_dev(self.h, self._ck_dev(out, self.W), self._ck_dev(x, self.W),
                                                    self._ck_dev(lut, self.p.N), lut.shape[0], int(first), B, self._stream()))
        return out

    def gate(self, op, a, b, out=None):
        B = a.shape[0]
        out = self.empty(B, self.W) if out is None else out
        _check(self.L, self.L.rs_gate_dev(self.h, GATES[op], self._ck_dev(out, self.W), self._ck_dev(a, self.W),
                                           self._ck_dev(b, self.W), B, self._stream()))
        return out

    def gate_mu(self, op, a, b, mu, out=None):
        B = a.shape[0]
        out = self.empty(B, self.W) if out is None else out
        _check(self.L, self.L.rs_gate_mu_dev(self.h, GATES[op], self._ck_dev(out, self.W), self._ck_dev(a, self.W),
                                              self._ck_dev(b, self.W), int(mu), B, self._stream()))
        return out

    def gather_rows(self, x, index):
        """out[i] = x[index[i]] (index: int32 CUDA tensor; negative -> zero sample)."""
        B = index.numel()
        out = self.empty(B, self.W)
        _check(self.L, self.L.rs_gather_rows_dev(self.h, self._ck_dev(out), self._ck_dev(x, self.W), C.c_void_p(index.data_ptr()), B,
                                                  self._stream()))
        return out

    def mux(self, a, b, c, out=None):
        B = a.shape[0]
        out = self.empty(B, self.W) if out is None else out
        _check(self.L, self.L.rs_mux_dev(self.h, self._ck_dev(out, self.W), self._ck_dev(a, self.W), self._ck_dev(b, self.W),
                                          self._ck_dev(c, self.W), B, self._stream()))
        return out

    def bootstrap_wo_ks(self, x, mu):
        B = x.shape[0]
        u = self.empty(B, self.p.N + 1)
        _check(self.L, self.L.rs_bootstrap_wo_ks_dev(self.h, self._ck_dev(u), self._ck_dev(x, self.W), int(mu), B, self._stream()))
        return u

    def keyswitch(self, u):
        B = u.shape[0]
        out = self.empty(B, self.W)
        _check(self.L, self.L.rs_keyswitch_dev(self.h, self._ck_dev(out), self._ck_dev(u, self.p.N + 1), B, self._stream()))
        return out

    # ---- linear stage ----
    def lincomb(self, a, ca, b=None, cb=0, bconst=0):
        B = a.numel() // self.W
        out = self.empty(*a.shape)
        pb = self._ck_dev(b, self.W) if b is not None else None
        _check(self.L, self.L.rs_lincomb_dev(self.h, self._ck_dev(out), self._ck_dev(a, self.W), int(ca), pb, int(cb),
                                              int(bconst), B, self._stream()))
        return out

    def _u8(self, t):
        import torch
        if t is None:
            return None
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
        return C.c_void_p(t.data_ptr())

    def linear_fc(self, x, sign, zero=None, zero_tap_b=0, bias_b=None):
        """x [K][W]; sign/zero uint8 [K][M] -> [M][W]."""
        K = x.shape[0]
        M = sign.shape[1]
        out = self.empty(M, self.W)
        pbias = self._ck_dev(bias_b) if bias_b is not None else None
        depth = int(bias_b.numel()) if bias_b is not None else 0
        _check(self.L, self.L.rs_linear_fc_dev(self.h, self._ck_dev(out), self._ck_dev(x, self.W), self._u8(sign), self._u8(zero),
                                                K, M, int(zero_tap_b), pbias, depth, self._stream()))
        return out

    def conv_ternary(self, x, sign, zero, shape, zero_tap_b=0, pad_tap_b=0, bias_b=None):
        """x [H][Wd][Cin][W] -> [Ho][Wo][Cout][W]; shape: dict of rs_conv_shape fields."""
        s = RsConvShape(**shape)
        out = self.empty(s.Ho, s.Wo, s.Cout, self.W)
        pbias = self._ck_dev(bias_b) if bias_b is not None else None
        depth = int(bias_b.numel()) if bias_b is not None else 0
        _check(self.L, self.L.rs_conv_ternary_dev(self.h, self._ck_dev(out), self._ck_dev(x, self.W), self._u8(sign),
                                                   self._u8(zero), C.byref(s), int(zero_tap_b), int(pad_tap_b), pbias, depth,
                                                   self._stream()))
        return out

    def sumpool(self, x, shape, bias_b=None):
        s = RsPoolShape(**shape)
        out = self.empty(s.Ho, s.Wo, s.C, self.W)
        pbias = self._ck_dev(bias_b) if bias_b is not None else None
        depth = int(bias_b.numel()) if bias_b is not None else 0
        _check(self.L, self.L.rs_sumpool_dev(self.h, self._ck_dev(out), self._ck_dev(x, self.W), C.byref(s), pbias, depth,
                                              self._stream()))
        return out

    # ---- host (numpy) conveniences: synchronous H2D -> kernels -> D2H inside the library ----
    def bootstrap_host(self, x, mu):
        x, px = _np_i32(x)
        out = np.empty_like(x)
        _check(self.L, self.L.rs_bootstrap(self.h, out.ctypes.data_as(_i32p), px, int(mu), x.shape[0]))
        return out

    def gate_host(self, op, a, b):
        a, pa = _np_i32(a)
        b, pb = _np_i32(b)
        out = np.empty_like(a)
        _check(self.L, self.L.rs_gate(self.h, GATES[op], out.ctypes.data_as(_i32p), pa, pb, a.shape[0]))
        return out

    def mux_host(self, a, b, c):
        a, pa = _np_i32(a)
        b, pb = _np_i32(b)
        c, pc = _np_i32(c)
        out = np.empty_like(a)
        _check(self.L, self.L.rs_mux(self.h, out.ctypes.data_as(_i32p), pa, pb, pc, a.shape[0]))
        return out

    def polymul_host(self, a_small, b_torus):
        a, pa = _np_i32(a_small)
        b, pb = _np_i32(b_torus)
        out = np.empty_like(b)
        _check(self.L, self.L.rs_debug_polymul(self.h, out.ctypes.data_as(_i32p), pa, pb, a.size // self.p.N))
        return out

    # ---- arithmetic mode ----
    def set_mode(self, mode):
        """'fft' (default), 'exact' (exact NTT) or 'split' (split-key FFT, exact by an a-priori bound; the only mode of
        the parameter sets outside the specialised N = 1024 kernels)."""
        _check(self.L, self.L.rs_set_mode(self.h, {"exact": 0, "ntt": 0, "fft": 1, "split": 2}[mode]))

    def mode(self):
        m = C.c_int()
        _check(self.L, self.L.rs_get_mode(self.h, C.byref(m)))
        return {0: "exact", 1: "fft", 2: "split"}[m.value]

    def split_bound(self):
        """A-priori bound on the rounding distance of the split-key product for this parameter set."""
        d = C.c_double()
        _check(self.L, self.L.rs_split_bound(self.h, C.byref(d)))
        return d.value

    def rounding_certificate(self, reset=True):
        d = C.c_double()
        _check(self.L, self.L.rs_rounding_certificate(self.h, C.byref(d), 1 if reset else 0))
        return d.value

    def fft_fallbacks(self):
        """Calls (all streams) whose FFT result was recomputed exactly on the device: certificate >= the limit."""
        n = C.c_int64()
        _check(self.L, self.L.rs_fft_fallbacks(self.h, C.byref(n)))
        return n.value

    def set_certificate_limit(self, limit):
        """Rounding distance at which a call is recomputed exactly (default 0.25; 0 forces every call)."""
        _check(self.L, self.L.rs_set_certificate_limit(self.h, float(limit)))

    def certify(self, reset=True):
        """Synchronise torch's current stream -> (largest rounding distance, calls recomputed exactly) on it."""
        d, n = C.c_double(), C.c_int64()
        _check(self.L, self.L.rs_certify(self.h, self._stream(), C.byref(d), C.byref(n), 1 if reset else 0))
        return d.value, n.value

    def last_launch(self):
        """(form, waves per workgroup, ciphertexts per key sweep) of the last blind rotation on the current stream."""
        f, w, r = C.c_int32(), C.c_int32(), C.c_int64()
        _check(self.L, self.L.rs_last_launch(self.h, self._stream(), C.byref(f), C.byref(w), C.byref(r)))
        return {"form": ["per_wave", "workgroup", "duo", "coop2", "coop4", "general", "split_workgroup", "split_coop", "split_duo", "coop8", "coop8_listed"][f.value], "waves_per_block": w.value, "resident": r.value}

    def fp64_rate(self):
        """FP64 FMA lane-operations per second this device sustains right now (box calibration, see include/redsec_hip.h)."""
        v = C.c_double()
        _check(self.L, self.L.rs_debug_fp64_rate(self.h, C.byref(v)))
        return v.value

    def cohort_table(self):
        """Progress table [8 XCDs][64 slots] of the current stream's last lock-step launch with XCD cohorts (debug tap)."""
        out = np.empty((8, 64), dtype=np.int32)
        _check(self.L, self.L.rs_debug_cohort_table(self.h, self._stream(), out.ctypes.data_as(_i32p)))
        return out

    # ---- timing / facts ----
    def set_timing(self, on=True):
        _check(self.L, self.L.rs_set_timing(self.h, 1 if on else 0))

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        _check(self.L, self.L.rs_last_kernel_ms_stream(self.h, self._stream(), C.byref(a), C.byref(b)))
        return a.value, b.value

    def info(self):
        bk, ksk, wpb, cus = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _check(self.L, self.L.rs_info(self.h, C.byref(bk), C.byref(ksk), C.byref(wpb), C.byref(cus)))
        return {"bk_device_bytes": bk.value, "ksk_device_bytes": ksk.value, "waves_per_block": wpb.value, "num_cus": cus.value}

    def sync(self):
        _check(self.L, self.L.rs_sync(self.h))
