"""Known-answer tests that pin the CPU oracle (oracle/redsec_oracle.c).

The reference holds no test vectors for the TFHE boundary ("parity
unpinned"), so the oracle is pinned by self-validating KATs: exact arithmetic identities and
decrypt-level truth tables under seeded keys and seeded fresh encryptions.
"""
import numpy as np
import pytest

import oracle_lib as ol

ALPHA = 2.0 ** -15  # client/encrypt_image.cpp:10 SECALPHA


def test_modswitch_constants():
    # modSwitchToTorus32 values used by REDsec: BinOps_enc.cpp:184 (1/4096), :190 (1/2048), gates 1/8, 1/4
    assert ol.to_torus(1, 4096) == 1 << 20
    assert ol.to_torus(1, 2048) == 1 << 21
    assert ol.to_torus(1, 8) == 1 << 29
    assert ol.to_torus(1, 4) == 1 << 30
    assert ol.to_torus(-1, 8) == -(1 << 29)
    assert ol.to_torus(-255, 4096) == -255 * (1 << 20)


def test_modswitch_from_torus_matches_gates_cu_formula():
    # lib/GPU/gates.cu:39-42 ModSwitch2048: ((a << 32) + 2^52) >> 53 on 64-bit wrap-around
    L = ol.lib()
    rng = np.random.default_rng(0)
    vals = list(rng.integers(-2**31, 2**31, 2000)) + [0, -1, 2**31 - 1, -2**31, (1 << 20) - 1, 1 << 20, -(1 << 20)]
    for a in vals:
        a = int(a)
        expect = ((((a & 0xFFFFFFFF) << 32) + (1 << 52)) & 0xFFFFFFFFFFFFFFFF) >> 53
        assert L.ro_modswitch_from_torus32(a, 2048) == expect
        assert 0 <= expect < 2048


def test_encrypt_decrypt_roundtrip_message_space(toy_redsec):
    ks, _ = toy_redsec
    ms = np.arange(-2048, 2048, 37)
    ct = ks.encrypt([ol.to_torus(int(m), 4096) for m in ms], ALPHA, 1)
    dec = ks.decrypt(ct, 4096)
    got = np.array([ol.lib().ro_modswitch_from_torus32(int(d), 4096) for d in dec])
    got = np.where(got >= 2048, got - 4096, got)
    # client/decrypt_image.cpp:52-58 maps (msg_space/2, msg_space) to negatives; -2048 aliases +2048
    assert np.array_equal(got[1:], ms[1:])


@pytest.mark.parametrize("half,seed", [(64, 0), (4, 1)])
def test_schoolbook_equals_ntt_product(half, seed):
    rng = np.random.default_rng(seed)
    for trial in range(4):
        a = rng.integers(-half, half, 1024).astype(np.int32)
        b = rng.integers(-2**31, 2**31, 1024).astype(np.int32)
        if trial == 3:  # extreme magnitudes
            a[:] = -half
            b[:] = -2**31
        assert np.array_equal(ol.negacyclic_mul(a, b, "ntt"), ol.negacyclic_mul(a, b, "schoolbook"))


def test_negacyclic_wraparound_sign():
    a = np.zeros(1024, np.int32); a[1] = 1          # X
    b = np.zeros(1024, np.int32); b[1023] = 5       # 5 X^1023
    out = ol.negacyclic_mul(a, b, "ntt")            # 5 X^1024 = -5
    assert out[0] == -5 and np.count_nonzero(out) == 1


def test_bootstrap_ntt_path_equals_schoolbook_path(toy_default, toy_redsec):
    for ks, ctx in (toy_default, toy_redsec):
        mu = ol.to_torus(1, 8)
        ct = ks.encrypt([mu, -mu], ALPHA, 5)
        fast = ctx.bootstrap_batch(ct, mu)
        ctx.set_schoolbook(True)
        slow = ctx.bootstrap_batch(ct, mu)
        ctx.set_schoolbook(False)
        assert np.array_equal(fast, slow)


def test_bootstrap_fft_path_equals_exact_path(toy_default, toy_redsec):
    """ro_ctx_set_fft: the double-precision folded FFT (the class of arithmetic TFHE's CPU library uses,
    timed by bench.py as cpu_baseline) rounds to exactly the integer products of the exact NTT path."""
    for ks, ctx in (toy_default, toy_redsec):
        mu = ol.to_torus(1, 8)
        ct = ks.encrypt([mu, -mu, mu, -mu, mu], ALPHA, 6)
        exact = ctx.gate_batch("XOR", ct, ct[::-1].copy())
        ctx.set_fft(True)
        try:
            fast = ctx.gate_batch("XOR", ct, ct[::-1].copy())
        finally:
            ctx.set_fft(False)
        assert np.array_equal(fast, exact)


TRUTH = {
    "NAND": lambda a, b: 1 - (a & b), "AND": lambda a, b: a & b, "OR": lambda a, b: a | b,
    "NOR": lambda a, b: 1 - (a | b), "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b),
    "ANDNY": lambda a, b: (1 - a) & b, "ANDYN": lambda a, b: a & (1 - b),
    "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b),
}


@pytest.mark.parametrize("fixture", ["full_default", "full_redsec"])
def test_gate_truth_tables(fixture, request):
    ks, ctx = request.getfixturevalue(fixture)
    e8 = ol.to_torus(1, 8)
    A = np.array([0, 0, 1, 1]); B = np.array([0, 1, 0, 1])
    ca = ks.encrypt(np.where(A, e8, -e8), ALPHA, 7)
    cb = ks.encrypt(np.where(B, e8, -e8), ALPHA, 8)
    for op, fn in TRUTH.items():
        ph = ks.phase(ctx.gate_batch(op, ca, cb))
        assert np.array_equal((ph > 0).astype(int), fn(A, B)), op
        # outputs are fresh +-1/8 encodings: within a quarter of the margin
        assert np.all(np.abs(np.abs(ph / 2.0**29) - 1.0) < 0.25), op


def test_mux_truth_table(full_redsec):
    ks, ctx = full_redsec
    e8 = ol.to_torus(1, 8)
    bits = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)])
    enc = [ks.encrypt(np.where(bits[:, i], e8, -e8), ALPHA, 20 + i) for i in range(3)]
    ph = ks.phase(ctx.mux_batch(*enc))
    expect = np.where(bits[:, 0] == 1, bits[:, 1], bits[:, 2])
    assert np.array_equal((ph > 0).astype(int), expect)


def test_sign_bootstrap_redsec_params(full_redsec):
    """BinOps::binarize_int (BinOps_enc.cpp:182-186): +-1/4096 by the sign of the phase, for
    |m| >= 32 where mod-switch noise cannot flip the decision (SURVEY.md hard part 7)."""
    ks, ctx = full_redsec
    mu = ol.to_torus(1, 4096)
    ms = np.array([-2000, -1000, -300, -64, -32, 32, 64, 300, 1000, 2000])
    ct = ks.encrypt([ol.to_torus(int(m), 4096) for m in ms], ALPHA, 9)
    ph = ks.phase(ctx.bootstrap_batch(ct, mu)) / float(mu)
    assert np.array_equal(np.sign(ph), np.sign(ms))
    assert np.all(np.abs(np.abs(ph) - 1.0) < 0.3)


def test_trivial_input_gives_exact_mu(full_redsec):
    """Noiseless trivial inputs skip every CMUX (all bara = 0): output is the exact trivial +-mu."""
    ks, ctx = full_redsec
    mu = ol.to_torus(1, 4096)
    x = np.zeros((2, ks.W), np.int32)
    x[0, -1] = ol.to_torus(5, 4096)
    x[1, -1] = ol.to_torus(-5, 4096)
    out = ctx.bootstrap_batch(x, mu)
    assert np.all(out[:, :-1] == 0)
    assert out[0, -1] == mu and out[1, -1] == -mu


def test_keyswitch_preserves_phase(full_default):
    ks, ctx = full_default
    mu = ol.to_torus(1, 8)
    ct = ks.encrypt([mu, -mu, mu], ALPHA, 13)
    u = ctx.bootstrap_wo_ks(ct, mu)
    before = ks.phase_extracted(u).astype(np.int64)
    after = ks.phase(ctx.keyswitch(u)).astype(np.int64)
    # keyswitch noise: N*t rows of stdev 2^-15 plus rounding 2^-(t*basebit+1): far below 1/16
    assert np.all(np.abs(before - after) < 2**27)


def test_sample_extract_matches_accumulator(toy_default):
    ks, ctx = toy_default
    mu = ol.to_torus(1, 8)
    ct = ks.encrypt([mu], ALPHA, 17)
    acc = ctx.blind_rotate_acc(ct, mu)[0]
    u = ctx.bootstrap_wo_ks(ct, mu)[0]
    N = 1024
    assert u[0] == acc[0] and u[N] == acc[N]
    assert np.array_equal(u[1:N], (-acc[N - 1:0:-1].astype(np.int64)).astype(np.int32))


def test_linear_fc_matches_numpy():
    rng = np.random.default_rng(3)
    K, M, W = 37, 11, 25
    x = rng.integers(-2**31, 2**31, (K, W)).astype(np.int32)
    sign = rng.integers(0, 2, (K, M)).astype(np.uint8)
    zero = (rng.random((K, M)) < 0.2).astype(np.uint8)
    zb = -(1 << 20)
    out = ol.linear_fc(x, sign, zero, zb)
    s = np.where(zero == 1, 0, np.where(sign == 1, 1, -1)).astype(np.int64)
    ref = (s.T @ x.astype(np.int64))
    ref[:, -1] += zero.sum(axis=0).astype(np.int64) * zb
    assert np.array_equal(out, ref.astype(np.int32) if False else (ref & 0xFFFFFFFF).astype(np.uint32).view(np.int32))
