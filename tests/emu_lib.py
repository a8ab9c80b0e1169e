"""ctypes binding of the TEST-ONLY lane emulator (redsec_amd/librs_emulate.so)."""
import ctypes as C

import numpy as np

from redsec_amd import build as _build

_i32p = C.POINTER(C.c_int32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build_emulator())
        L.rs_emu_prime.restype = C.c_uint64
        L.rs_emu_forward.argtypes = [C.c_int, _i32p, C.POINTER(C.c_double)]
        L.rs_emu_digit_mismatches.restype = C.c_long
        L.rs_emu_digit_mismatches.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_long]
        L.rs_emu_gen_layout_violations.restype = C.c_long
        L.rs_emu_plane_layout_violations.restype = C.c_long
        L.rs_emu_gen_error_bound.restype = C.c_double
        L.rs_emu_gen_error_bound.argtypes = [C.c_int, C.c_int, C.c_int]
        L.rs_emu_gen_transform_errors.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rs_emu_exchange_schedule.restype = C.c_long
        L.rs_emu_exchange_schedule.argtypes = [C.c_long, C.c_int, C.POINTER(C.c_long)]
        L.rs_emu_gen_digit_mismatches.restype = C.c_long
        L.rs_emu_gen_digit_mismatches.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_long]
        L.rs_emu_gen_polymul.argtypes = [C.c_int, _i32p, _i32p, _i32p, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def gen_polymul(logn, a, b):
    """Split-key product of the general ring path (rs_general.h) -> (product mod 2^32, largest rounding distance)."""
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.zeros(1 << logn, np.int32)
    dev = C.c_double(0)
    assert lib().rs_emu_gen_polymul(logn, _p(a), _p(b), _p(out), C.byref(dev)) == 0
    return out, dev.value


def _p(a):
    return None if a is None else a.ctypes.data_as(_i32p)


def polymul(cfg, a, b):
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.zeros(1024, np.int32)
    assert lib().rs_emu_polymul(cfg, _p(a), _p(b), _p(out)) == 0
    return out


def blind_rotate(cfg, n, in0, in1, c0, c1, bconst, mu, bk, steps=-1):
    in0 = np.ascontiguousarray(in0, np.int32)
    in1 = None if in1 is None else np.ascontiguousarray(in1, np.int32)
    u = np.zeros(1025, np.int32); acc = np.zeros(2048, np.int32)
    rc = lib().rs_emu_blind_rotate(cfg, n, _p(in0), _p(in1), int(c0), int(c1), int(bconst), int(mu), _p(bk), _p(u), _p(acc), steps)
    assert rc == 0
    return u, acc


def forward(cfg, poly):
    poly = np.ascontiguousarray(poly, np.int32)
    out = np.zeros(1024, np.float64)
    assert lib().rs_emu_forward(cfg, _p(poly), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return out


def set_planar(on):
    """FFT entry points emulate the planar LDS exchange of the workgroup kernel (True) or the interleaved one."""
    lib().rs_emu_set_planar(1 if on else 0)


def polymul_fft(a, b):
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    out = np.zeros(1024, np.int32)
    dev = C.c_double(0)
    assert lib().rs_emu_polymul_fft(_p(a), _p(b), _p(out), C.byref(dev)) == 0
    return out, dev.value


def blind_rotate_fft(cfg, n, in0, in1, c0, c1, bconst, mu, bk, steps=-1):
    in0 = np.ascontiguousarray(in0, np.int32)
    in1 = None if in1 is None else np.ascontiguousarray(in1, np.int32)
    u = np.zeros(1025, np.int32); acc = np.zeros(2048, np.int32)
    dev = C.c_double(0)
    rc = lib().rs_emu_blind_rotate_fft(cfg, n, _p(in0), _p(in1), int(c0), int(c1), int(bconst), int(mu), _p(bk), _p(u), _p(acc), steps,
                                       C.byref(dev))
    assert rc == 0
    return u, acc, dev.value


def forward_digits(cfg, coef, q):
    coef = np.ascontiguousarray(coef, np.int32)
    out = np.zeros(1024, np.float64)
    assert lib().rs_emu_forward_digits(cfg, _p(coef), int(q), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    return out


def validate(cfg):
    msg = C.create_string_buffer(256)
    rc = lib().rs_emu_validate(cfg, msg, 256)
    return rc, msg.value.decode()


def digit_mismatches(cfg, start, step, count):
    return lib().rs_emu_digit_mismatches(int(cfg), int(start) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF, int(count))
