"""ctypes binding of the CPU oracle (oracle/libredsec_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. The product package (redsec_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "libredsec_oracle.so")


class RoParams(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("N", C.c_int32), ("k", C.c_int32),
        ("bk_l", C.c_int32), ("bk_Bgbit", C.c_int32),
        ("ks_t", C.c_int32), ("ks_basebit", C.c_int32),
        ("lwe_stdev", C.c_double), ("bk_stdev", C.c_double),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class RoRng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("has_spare", C.c_int), ("spare", C.c_double)]


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "redsec_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "redsec_oracle.h")
    from redsec_amd import build as _b      # its content-fingerprint helpers only (a pushed snapshot never recompiles)
    deps = [src, hdr, os.path.join(ORACLE_DIR, "Makefile")]
    if force or _b.is_stale(_SO, deps):
        subprocess.check_call(["make", "-s", "-B", "-C", ORACLE_DIR, os.path.join(ORACLE_DIR, "libredsec_oracle.so")])
        _b.write_stamp(_SO, deps)
    return _SO


_lib = None
_i32p = C.POINTER(C.c_int32)


def _cpu_share():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    # OpenMP teams no larger than the CPUs this process may actually use: a GPU box shows 256 hardware threads and grants a
    # cgroup quota of 16, and a 256-thread team spinning on 16 CPUs makes every parallel region slower than a serial one
    L.ro_set_threads(C.c_int(_cpu_share()))
    P = C.POINTER(RoParams)
    L.ro_params_default128.argtypes = [P]
    L.ro_params_redsec_small_v2.argtypes = [P]
    for f in (L.ro_params_redsec_small, L.ro_params_redsec_medium, L.ro_params_redsec_large):
        f.argtypes = [P]
    for f in (L.ro_modswitch_to_torus32, L.ro_modswitch_from_torus32, L.ro_approx_phase):
        f.argtypes = [C.c_int32, C.c_int32]
        f.restype = C.c_int32
    L.ro_rng_seed.argtypes = [C.POINTER(RoRng), C.c_uint64]
    L.ro_bk_words.argtypes = [P]; L.ro_bk_words.restype = C.c_size_t
    L.ro_ksk_words.argtypes = [P]; L.ro_ksk_words.restype = C.c_size_t
    L.ro_keygen.argtypes = [P, C.c_uint64, _i32p, _i32p, _i32p, _i32p]
    L.ro_synthetic_key_words.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _i32p]
    L.ro_lwe_encrypt.argtypes = [_i32p, C.c_int32, C.c_double, _i32p, C.c_int32, C.POINTER(RoRng)]
    L.ro_lwe_phase.argtypes = [_i32p, _i32p, C.c_int32]; L.ro_lwe_phase.restype = C.c_int32
    L.ro_lwe_decrypt.argtypes = [_i32p, _i32p, C.c_int32, C.c_int32]; L.ro_lwe_decrypt.restype = C.c_int32
    L.ro_negacyclic_mul_schoolbook.argtypes = [_i32p, _i32p, _i32p, C.c_int32]
    L.ro_negacyclic_mul_ntt.argtypes = [_i32p, _i32p, _i32p, C.c_int32]
    L.ro_ctx_create.argtypes = [P, _i32p, _i32p]; L.ro_ctx_create.restype = C.c_void_p
    L.ro_ctx_destroy.argtypes = [C.c_void_p]
    L.ro_ctx_set_schoolbook.argtypes = [C.c_void_p, C.c_int]
    L.ro_ctx_set_fft.argtypes = [C.c_void_p, C.c_int]
    L.ro_bootstrap_wo_ks.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p]
    L.ro_keyswitch.argtypes = [C.c_void_p, _i32p, _i32p]
    L.ro_bootstrap.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p]
    L.ro_blind_rotate_acc.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p, C.c_int32]
    L.ro_gate_precombine.argtypes = [C.c_int, _i32p, _i32p, _i32p, C.c_int32]
    L.ro_gate.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p]
    L.ro_mux.argtypes = [C.c_void_p, _i32p, _i32p, _i32p, _i32p]
    L.ro_bootstrap_batch.argtypes = [C.c_void_p, _i32p, C.c_int32, _i32p, C.c_size_t]
    L.ro_gate_batch.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, _i32p, C.c_size_t]
    L.ro_mux_batch.argtypes = [C.c_void_p, _i32p, _i32p, _i32p, _i32p, C.c_size_t]
    L.ro_bootstrap_lut.argtypes = [C.c_void_p, _i32p, _i32p, _i32p]
    L.ro_bootstrap_lut_batch.argtypes = [C.c_void_p, _i32p, _i32p, C.c_size_t, _i32p, C.c_size_t]
    L.ro_max_threads.restype = C.c_int
    L.ro_set_threads.argtypes = [C.c_int]
    L.ro_linear_fc.argtypes = [_i32p, _i32p, _u8p, _u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.ro_add_bias.argtypes = [_i32p, _i32p, C.c_int32, C.c_int32, C.c_int32]
    _lib = L
    return L


def _p(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i32p)


def _pu8(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


GATES = {"NAND": 0, "OR": 1, "AND": 2, "NOR": 3, "XOR": 4, "XNOR": 5, "ANDNY": 6, "ANDYN": 7, "ORNY": 8, "ORYN": 9}


def params(name):
    p = RoParams()
    if name == "default128":
        lib().ro_params_default128(C.byref(p))
    elif name == "redsec_small_v2":
        lib().ro_params_redsec_small_v2(C.byref(p))
    elif name == "toy":
        # reduced-n variant of default128 for fast tests: same ring, gadget and keyswitch shape
        lib().ro_params_default128(C.byref(p))
        p.n = 24
    elif name == "toy_ks6":
        # non-standard keyswitch shape: exercises the generic (gather) keyswitch kernel
        lib().ro_params_default128(C.byref(p))
        p.n = 24
        p.ks_t = 6
    elif name == "toy_redsec":
        lib().ro_params_redsec_small_v2(C.byref(p))
        p.n = 20
    elif name in ("redsec_small", "redsec_medium", "redsec_large"):
        getattr(lib(), "ro_params_" + name)(C.byref(p))
    elif name in ("toy_small", "toy_medium", "toy_large"):
        # the ring, gadget and keyswitch shape of the set, LWE dimension cut down so that keys generate in seconds
        getattr(lib(), "ro_params_redsec_" + name[4:])(C.byref(p))
        p.n = {"toy_small": 16, "toy_medium": 10, "toy_large": 6}[name]
    elif name == "toy_n2048":
        # a ring degree no shipped set uses: default-128 gadget on N = 2048
        lib().ro_params_default128(C.byref(p))
        p.n = 12; p.N = 2048
    else:
        raise KeyError(name)
    return p


def to_torus(mu, msize):
    return lib().ro_modswitch_to_torus32(int(mu), int(msize))


class KeySet:
    """Secret + evaluation keys from the oracle's deterministic generator."""

    def __init__(self, p, seed=0):
        self.p = p
        self.lwe_key = np.zeros(p.n, np.int32)
        self.tlwe_key = np.zeros(p.k * p.N, np.int32)
        self.bk = np.zeros(lib().ro_bk_words(C.byref(p)), np.int32)
        self.ksk = np.zeros(lib().ro_ksk_words(C.byref(p)), np.int32)
        lib().ro_keygen(C.byref(p), seed, _p(self.lwe_key), _p(self.tlwe_key), _p(self.bk), _p(self.ksk))

    @property
    def W(self):
        return self.p.n + 1

    def encrypt(self, mus, alpha, seed):
        """Fresh LWE encryptions of torus32 messages `mus` -> int32 [B][n+1]."""
        mus = np.asarray(mus, dtype=np.int64).ravel()
        rng = RoRng()
        lib().ro_rng_seed(C.byref(rng), seed)
        out = np.zeros((len(mus), self.W), np.int32)
        for i, mu in enumerate(mus):
            mu32 = int(np.int64(mu).astype(np.int32)) if not (-2**31 <= mu < 2**31) else int(mu)
            lib().ro_lwe_encrypt(_p(out[i]), mu32, alpha, _p(self.lwe_key), self.p.n, C.byref(rng))
        return out

    def phase(self, samples):
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, self.W)
        return np.array([lib().ro_lwe_phase(_p(s), _p(self.lwe_key), self.p.n) for s in samples], np.int32)

    def phase_extracted(self, samples):
        Nk = self.p.k * self.p.N
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, Nk + 1)
        return np.array([lib().ro_lwe_phase(_p(s), _p(self.tlwe_key), Nk) for s in samples], np.int32)

    def decrypt(self, samples, msize):
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, self.W)
        return np.array([lib().ro_lwe_decrypt(_p(s), _p(self.lwe_key), self.p.n, msize) for s in samples], np.int32)


class Ctx:
    def __init__(self, keys):
        self.keys = keys
        self.p = keys.p
        self.h = lib().ro_ctx_create(C.byref(keys.p), _p(keys.bk), _p(keys.ksk))

    def close(self):
        if self.h:
            lib().ro_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_schoolbook(self, flag):
        lib().ro_ctx_set_schoolbook(self.h, int(flag))

    def set_fft(self, flag):
        """Double-precision FFT products (fast; the cpu_baseline path). Parity checks use the exact default."""
        lib().ro_ctx_set_fft(self.h, int(flag))

    def bootstrap_batch(self, x, mu):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros_like(x)
        lib().ro_bootstrap_batch(self.h, _p(out), int(mu), _p(x), x.shape[0])
        return out

    def bootstrap_lut_batch(self, x, luts):
        """Programmable bootstrap: ciphertext b uses the test polynomial luts[b % len(luts)] ([L][N] int32)."""
        x = np.ascontiguousarray(x, np.int32)
        luts = np.ascontiguousarray(luts, np.int32).reshape(-1, self.p.N)
        out = np.zeros_like(x)
        lib().ro_bootstrap_lut_batch(self.h, _p(out), _p(luts), luts.shape[0], _p(x), x.shape[0])
        return out

    def bootstrap_wo_ks(self, x, mu):
        x = np.ascontiguousarray(x, np.int32)
        Nk = self.p.k * self.p.N
        out = np.zeros((x.shape[0], Nk + 1), np.int32)
        for i in range(x.shape[0]):
            lib().ro_bootstrap_wo_ks(self.h, _p(out[i]), int(mu), _p(x[i]))
        return out

    def keyswitch(self, u):
        u = np.ascontiguousarray(u, np.int32)
        out = np.zeros((u.shape[0], self.p.n + 1), np.int32)
        for i in range(u.shape[0]):
            lib().ro_keyswitch(self.h, _p(out[i]), _p(u[i]))
        return out

    def blind_rotate_acc(self, x, mu, steps=-1):
        x = np.ascontiguousarray(x, np.int32)
        out = np.zeros((x.shape[0], (self.p.k + 1) * self.p.N), np.int32)
        for i in range(x.shape[0]):
            lib().ro_blind_rotate_acc(self.h, _p(out[i]), int(mu), _p(x[i]), steps)
        return out

    def gate_batch(self, op, a, b):
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
        out = np.zeros_like(a)
        lib().ro_gate_batch(self.h, GATES[op], _p(out), _p(a), _p(b), a.shape[0])
        return out

    def mux_batch(self, a, b, c):
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
        c = np.ascontiguousarray(c, np.int32)
        out = np.zeros_like(a)
        lib().ro_mux_batch(self.h, _p(out), _p(a), _p(b), _p(c), a.shape[0])
        return out


def gate_precombine(op, a, b):
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.zeros_like(a)
    for i in range(a.shape[0]):
        lib().ro_gate_precombine(GATES[op], _p(out[i]), _p(a[i]), _p(b[i]), a.shape[1] - 1)
    return out


def negacyclic_mul(a_small, b_torus, method="ntt"):
    a_small = np.ascontiguousarray(a_small, np.int32); b_torus = np.ascontiguousarray(b_torus, np.int32)
    out = np.zeros_like(b_torus)
    f = lib().ro_negacyclic_mul_ntt if method == "ntt" else lib().ro_negacyclic_mul_schoolbook
    f(_p(out), _p(a_small), _p(b_torus), len(a_small))
    return out


def linear_fc(x, sign, zero, zero_tap_b=0):
    """x: [K][W] int32; sign/zero: [K][M] uint8 -> [M][W]."""
    x = np.ascontiguousarray(x, np.int32)
    sign = np.ascontiguousarray(sign, np.uint8)
    zero = None if zero is None else np.ascontiguousarray(zero, np.uint8)
    K, W = x.shape
    M = sign.shape[1]
    out = np.zeros((M, W), np.int32)
    lib().ro_linear_fc(_p(out), _p(x), _pu8(sign), _pu8(zero), K, M, W, int(zero_tap_b))
    return out


def synthetic_key_words(seed, count, first=0):
    """Words [first, first + count) of the synthetic key of `seed` (the oracle's restatement of rs_load_synthetic_keys' generator)."""
    out = np.empty(int(count), np.int32)
    lib().ro_synthetic_key_words(int(seed) & (2**64 - 1), int(first), int(count), _p(out))
    return out
