"""Host-side client (redsec_amd/client.py): keys it generates are valid CGGI keys in the C-ABI
layout -- checked by evaluating gates on them with the CPU oracle -- and its encrypt/decrypt follow
client/encrypt_image.cpp / client/decrypt_image.cpp."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from redsec_amd import client


def _oracle_ctx(sk):
    p = ol.params(sk.name)
    p.n = sk.n

    class K:  # duck-typed KeySet for ol.Ctx
        pass
    k = K(); k.p = p; k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
    return ol.Ctx(k)


@pytest.mark.parametrize("name,n", [("default128", 40), ("redsec_small_v2", 32), ("redsec_small", 24), ("redsec_medium", 6)])
def test_generated_keys_evaluate_gates(name, n):
    sk = client.SecretKeySet(name, seed=11, n=n)
    ctx = _oracle_ctx(sk)
    A = np.array([0, 0, 1, 1]); B = np.array([0, 1, 0, 1])
    ca, cb = sk.encrypt_bits(A, seed=1), sk.encrypt_bits(B, seed=2)
    assert np.array_equal(sk.decrypt_bits(ca), A)
    for op, fn in (("NAND", lambda a, b: 1 - (a & b)), ("XOR", lambda a, b: a ^ b), ("OR", lambda a, b: a | b)):
        out = ctx.gate_batch(op, ca, cb)
        assert np.array_equal(sk.decrypt_bits(out), fn(A, B)), op


def test_image_encoding_roundtrip():
    sk = client.SecretKeySet("redsec_small_v2", seed=3, n=16)
    pixels = np.array([0, 1, 17, 128, 254, 255])
    ct = sk.encrypt_image(pixels, seed=5)
    assert np.array_equal(sk.decrypt_ints(ct), 2 * pixels - 255)   # encrypt_image.cpp:76
    # agrees with the oracle's lweSymDecrypt
    dec = np.array([ol.lib().ro_lwe_decrypt(ol._p(c), ol._p(sk.lwe_key), sk.n, 4096) for c in ct])
    assert np.array_equal(dec >> 20, 2 * pixels - 255)


def test_modswitch_to_torus_matches_oracle():
    for mu, m in [(1, 4096), (-1, 8), (255, 4096), (-2047, 4096), (3, 2048)]:
        assert int(client.modswitch_to_torus32([mu], m)[0]) == ol.to_torus(mu, m)
