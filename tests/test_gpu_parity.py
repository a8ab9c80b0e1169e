"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact, on identical keys and
identical seeded inputs. Integer work => the bar is equality of every ciphertext word."""
import contextlib
import os
import numpy as np
import pytest

import oracle_lib as ol
from backend_pool import BackendPool

pytestmark = pytest.mark.gpu

ALPHA = 2.0 ** -15
ALL_GATES = ["NAND", "OR", "AND", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]


POOL = BackendPool()      # contexts this module keeps alive; a closed one is never called again (tests/backend_pool.py)


def _make(ks, name, **fields):
    import torch
    import redsec_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    p = redsec_amd.params(name, n=ks.p.n)
    for k, v in fields.items():
        setattr(p, k, v)
    be = redsec_amd.Backend(p, device=0)
    be.load_keys(ks.bk, ks.ksk)
    return be


def _backend(ks, name):
    """A context for the module's lifetime (the module-scoped fixtures below)."""
    return POOL.add(_make(ks, name))


def _scratch(ks, name, **fields):
    """`with _scratch(...) as be:` -- a context for one test body, closed and forgotten by the pool."""
    return POOL.scratch(lambda: _make(ks, name, **fields))


@pytest.fixture(scope="module", autouse=True)
def _close_module_contexts():
    yield
    POOL.close_all()


@pytest.fixture(autouse=True, params=["fft", "exact"])
def arith_mode(request):
    """Every parity test runs twice: in the FFT mode (default; exact after rounding, checked against the
    SAME exact oracle) and in the guaranteed-exact NTT mode. After an FFT-mode test the rounding
    certificate must be far below 1/2."""
    POOL.enter_mode(request.param)
    yield request.param
    POOL.leave_mode(request.param)


@pytest.fixture(scope="module")
def be_toy_default(toy_default):
    return _backend(toy_default[0], "default128")


@pytest.fixture(scope="module")
def be_toy_redsec(toy_redsec):
    return _backend(toy_redsec[0], "redsec_small_v2")


@pytest.fixture(scope="module")
def be_full_default(full_default):
    return _backend(full_default[0], "default128")


@pytest.fixture(scope="module")
def be_full_redsec(full_redsec):
    return _backend(full_redsec[0], "redsec_small_v2")


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, np.int32)).cuda()


def _bits(ks, B, seed):
    rng = np.random.default_rng(seed)
    e8 = ol.to_torus(1, 8)
    bits = rng.integers(0, 2, B)
    return bits, ks.encrypt(np.where(bits == 1, e8, -e8), ALPHA, seed)


@pytest.mark.parametrize("which,half", [("be_toy_default", 64), ("be_toy_redsec", 4)])
def test_polymul_matches_schoolbook(which, half, request):
    be = request.getfixturevalue(which)
    rng = np.random.default_rng(1)
    a = rng.integers(-half, half, (9, 1024)).astype(np.int32)
    b = rng.integers(-2**31, 2**31, (9, 1024)).astype(np.int32)
    a[7] = -half; b[7] = -2**31                # largest magnitudes
    a[8] = half - 1; b[8] = 2**31 - 1
    out = be.polymul_host(a, b)
    for i in range(9):
        assert np.array_equal(out[i], ol.negacyclic_mul(a[i], b[i], "schoolbook")), i


@pytest.mark.parametrize("which,fix", [("be_toy_default", "toy_default"), ("be_toy_redsec", "toy_redsec")])
@pytest.mark.parametrize("B", [1, 7, 64, 130])
def test_toy_blind_rotate_and_keyswitch(which, fix, B, request):
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    mu = ol.to_torus(1, 8)
    _, ct = _bits(ks, B, 100 + B)
    u = be.bootstrap_wo_ks(_dev(ct), mu).cpu().numpy()
    ref_u = ctx.bootstrap_wo_ks(ct, mu)
    assert np.array_equal(u, ref_u)
    out = be.keyswitch(_dev(ref_u)).cpu().numpy()
    assert np.array_equal(out, ctx.keyswitch(ref_u))


@pytest.mark.parametrize("which,fix", [("be_toy_default", "toy_default"), ("be_toy_redsec", "toy_redsec")])
def test_toy_all_gates_and_mux(which, fix, request):
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    B = 33
    _, ca = _bits(ks, B, 1)
    _, cb = _bits(ks, B, 2)
    _, cc = _bits(ks, B, 3)
    for op in ALL_GATES:
        got = be.gate(op, _dev(ca), _dev(cb)).cpu().numpy()
        assert np.array_equal(got, ctx.gate_batch(op, ca, cb)), op
    got = be.mux(_dev(ca), _dev(cb), _dev(cc)).cpu().numpy()
    assert np.array_equal(got, ctx.mux_batch(ca, cb, cc))


def test_batched_ripple_carry_adder(be_full_default, full_default):
    """BinOps::add (lib/BinOps_enc.cpp:55-119: per bit 2 XOR + 2 AND + 1 OR, the full-adder popcount
    pattern) with every gate level run as ONE batch over all the additions: 32 independent 4-bit sums,
    each gate batch equal to the oracle's word for word, the decrypted sums equal to a + b mod 16."""
    be = be_full_default
    ks, ctx = full_default
    B, bits = 32, 4
    rng = np.random.default_rng(99)
    xa, xb = rng.integers(0, 16, B), rng.integers(0, 16, B)
    e8 = ol.to_torus(1, 8)
    enc = lambda v, seed: [ks.encrypt(np.where((v >> i) & 1, e8, -e8), ALPHA, seed + i) for i in range(bits)]
    ca, cb = enc(xa, 1000), enc(xb, 2000)

    def gate(op, u, v):
        got = be.gate(op, _dev(u), _dev(v)).cpu().numpy()
        assert np.array_equal(got, ctx.gate_batch(op, u, v)), op
        return got
    out = []
    carry = None
    for i in range(bits):
        t0 = gate("XOR", ca[i], cb[i])
        if carry is None:                      # carry-in is the constant 0 (bootsCONSTANT): sum = a ^ b, carry = a & b
            out.append(t0)
            carry = gate("AND", ca[i], cb[i])
            continue
        out.append(gate("XOR", carry, t0))
        if i + 1 < bits:
            carry = gate("OR", gate("AND", carry, t0), gate("AND", ca[i], cb[i]))
    total = sum(((ks.phase(o) > 0).astype(int) << i) for i, o in enumerate(out))
    assert np.array_equal(total, (xa + xb) % 16)


@pytest.mark.parametrize("which,fix", [("be_toy_default", "toy_default"), ("be_toy_redsec", "toy_redsec")])
def test_keyswitch_tile_boundaries(which, fix, request):
    """Tiled keyswitch: 256-ciphertext tiles, 32-word chunks (W = 25 / 21 here: one partial chunk);
    batches straddling a tile, and the summed-input form used by bootsMUX."""
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    rng = np.random.default_rng(3)
    for B in (255, 256, 257, 300):
        u = rng.integers(-2**31, 2**31, (B, 1025)).astype(np.int32)
        assert np.array_equal(be.keyswitch(_dev(u)).cpu().numpy(), ctx.keyswitch(u)), B


def test_generic_keyswitch_shape(arith_mode):
    """ks_t = 6 is not one of the tiled instantiations: the generic gather kernel must agree too."""
    p = ol.params("toy_ks6")
    ks = ol.KeySet(p, seed=9)
    ctx = ol.Ctx(ks)
    _, ca = _bits(ks, 19, 1)
    _, cb = _bits(ks, 19, 2)
    with _scratch(ks, "default128", ks_t=6) as be:
        be.set_mode(arith_mode)
        assert np.array_equal(be.gate("XNOR", _dev(ca), _dev(cb)).cpu().numpy(), ctx.gate_batch("XNOR", ca, cb))
    ctx.close()


def test_empty_batch_and_host_api(be_toy_default, toy_default):
    be = be_toy_default
    ks, ctx = toy_default
    import torch
    empty = torch.empty((0, ks.W), dtype=torch.int32, device="cuda")
    assert be.bootstrap(empty, 1 << 29).shape == (0, ks.W)
    assert be.gate("NAND", empty, empty).shape == (0, ks.W)
    _, ca = _bits(ks, 5, 9)
    _, cb = _bits(ks, 5, 10)
    assert np.array_equal(be.gate_host("NAND", ca, cb), ctx.gate_batch("NAND", ca, cb))
    assert np.array_equal(be.bootstrap_host(ca, 1 << 20), ctx.bootstrap_batch(ca, 1 << 20))
    assert np.array_equal(be.mux_host(ca, cb, ca), ctx.mux_batch(ca, cb, ca))


def test_large_host_pointer_calls_equal_the_device_calls(be_toy_default, toy_default):
    """rs_gate / rs_bootstrap / rs_mux with HOST pointers (the call shape of the reference's TFHE API, SURVEY.md section 8b;
    one upload, the device call, one download) on batches of several launch rounds: equal to the device-pointer call on the
    whole batch word for word, and to the oracle on a sample."""
    be = be_toy_default
    ks, ctx = toy_default
    rounds = 8 * be.info()["num_cus"]
    B = 2 * rounds + 7
    _, ca = _bits(ks, B, 33); _, cb = _bits(ks, B, 34); _, cc = _bits(ks, B, 35)
    assert np.array_equal(be.gate_host("XOR", ca, cb), be.gate("XOR", _dev(ca), _dev(cb)).cpu().numpy())
    assert np.array_equal(be.mux_host(ca, cb, cc), be.mux(_dev(ca), _dev(cb), _dev(cc)).cpu().numpy())
    want = be.bootstrap(_dev(ca), 1 << 20).cpu().numpy()
    assert np.array_equal(be.bootstrap_host(ca, 1 << 20), want)
    sample = np.r_[0:4, rounds - 2:rounds + 2, B - 4:B]
    assert np.array_equal(want[sample], ctx.bootstrap_batch(ca[sample], 1 << 20))


def test_full_default128_nand_bit_exact_and_decrypts(be_full_default, full_default):
    """BASELINE config 2 shape at oracle-checkable size: default-128 NANDs, every word equal."""
    be = be_full_default
    ks, ctx = full_default
    B = 48
    ba, ca = _bits(ks, B, 0xC0FFEE)
    bb, cb = _bits(ks, B, 0xC0FFEF)
    got = be.gate("NAND", _dev(ca), _dev(cb)).cpu().numpy()
    assert np.array_equal(got, ctx.gate_batch("NAND", ca, cb))
    assert np.array_equal((ks.phase(got) > 0).astype(int), 1 - (ba & bb))


def test_full_redsec_sign_bootstrap_bit_exact(be_full_redsec, full_redsec):
    """BinOps::binarize_int on the shipped REDsec parameter set (BinOps_enc.cpp:182-186)."""
    be = be_full_redsec
    ks, ctx = full_redsec
    mu = ol.to_torus(1, 4096)
    rng = np.random.default_rng(5)
    ms = rng.integers(-2000, 2000, 40)
    ct = ks.encrypt([ol.to_torus(int(m), 4096) for m in ms], ALPHA, 77)
    got = be.bootstrap(_dev(ct), mu).cpu().numpy()
    assert np.array_equal(got, ctx.bootstrap_batch(ct, mu))
    ph = ks.phase(got) / float(mu)
    strong = np.abs(ms) >= 32
    assert np.array_equal(np.sign(ph[strong]), np.sign(ms[strong]))


def test_large_batch_truth_table_property(be_full_default, full_default):
    """Size-independent property at a batch far beyond what the oracle can check word-by-word:
    every output decrypts to NAND(a, b) and re-running is idempotent (deterministic)."""
    be = be_full_default
    ks, _ = full_default
    B = 4096
    ba, ca = _bits(ks, B, 21)
    bb, cb = _bits(ks, B, 22)
    da, db = _dev(ca), _dev(cb)
    out1 = be.gate("NAND", da, db)
    out2 = be.gate("NAND", da, db)
    import torch
    assert torch.equal(out1, out2)
    got = out1.cpu().numpy()
    assert np.array_equal((ks.phase(got) > 0).astype(int), 1 - (ba & bb))
    # a different wave/block mapping (small batch) must give the same words
    sub = be.gate("NAND", da[:37].contiguous(), db[:37].contiguous()).cpu().numpy()
    assert np.array_equal(sub, got[:37])


@pytest.mark.parametrize("size", ["wg8", "duo"])
@pytest.mark.parametrize("which,fix", [("be_toy_default", "toy_default"), ("be_toy_redsec", "toy_redsec")])
def test_workgroup_kernel_ragged_groups_and_identity_steps(which, fix, size, request, monkeypatch):
    """The lock-step workgroup kernels of the FFT mode -- blind_rotate_wg_kernel (B > 4 x #CUs, 8
    ciphertexts per group; odd l also 2 x #CUs < B <= 4 x #CUs with 4 per group) and blind_rotate_duo_kernel
    (2 x #CUs < B <= 4 x #CUs, even l: 4 ciphertexts x 2 waves) -- on a batch whose last group is ragged AND spills past one group per workgroup, with
    mask words forced to 0 so that some CMUX steps are the identity (tfhe_blindRotate_FFT skips them;
    the lock-step waves must still keep their barriers). Checked word for word against the per-wave
    kernel (RS_NO_WG) and, on a sample, against the oracle."""
    import torch
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    cus = be.info()["num_cus"]
    B = 8 * cus + 3 if size == "wg8" else 3 * cus + 6
    bits, ct = _bits(ks, B, 4242)
    ct = ct.copy()
    ct[5, :3] = 0            # leading identity steps
    ct[6, 1::2] = 0          # every other step
    ct[B - 1, -3:-1] = 0     # in the ragged group, trailing steps
    ct[B - 2, : ks.p.n] = 0  # a ciphertext whose whole blind rotation is the identity
    mu = ol.to_torus(1, 8)
    d = _dev(ct)
    got = be.bootstrap(d, mu)
    if be.mode() == "fft":
        launch = be.last_launch()   # odd l has no duo form: half-size lock-step groups (4 waves per workgroup) instead
        assert (launch["form"], launch["waves_per_block"]) == (("workgroup", 8) if size == "wg8" else (("duo", 8) if ks.p.bk_l % 2 == 0 else ("workgroup", 4)))
    # the launch switches are read once, at context creation: a second context under RS_NO_WG runs the per-wave kernel
    monkeypatch.setenv("RS_NO_WG", "1")
    with _scratch(ks, "default128" if ks.p.bk_l == 3 else "redsec_small_v2") as be2:
        monkeypatch.delenv("RS_NO_WG")
        be2.set_mode(be.mode())
        ref = be2.bootstrap(d, mu)
        assert be2.last_launch()["form"] == "per_wave"
        assert torch.equal(got, ref)
    sample = np.r_[0:8, B - 11:B]
    assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ct[sample], mu))


@pytest.mark.parametrize("fix", ["toy_default", "full_default"])
def test_coop8_listed_step_equals_the_round5_kernel_and_the_oracle(fix, request, monkeypatch):
    """blind_rotate_coop8_listed_kernel (gadgets with l < 4, B <= #CUs; round 6: the CMUX steps listed once in LDS, the rotated
    difference built once per step by all 512 threads, key rows requested a phase early) against blind_rotate_coop8_kernel
    (RS_NO_COOP8_LISTED=1: the form every earlier round measured) on WHOLE batches word for word -- gates, a bootstrap whose test
    polynomial depends on the ciphertext's index, batches of 1, a few and #CUs ciphertexts, ciphertexts with identity steps at
    the front, in the middle, at the end, everywhere -- and against the oracle on sampled rows (tfhe_blindRotate_FFT skips
    bara = 0 as the step list does)."""
    import torch
    ks, ctx = request.getfixturevalue(fix)
    be = request.getfixturevalue("be_" + fix)
    if be.mode() != "fft":
        pytest.skip("the listed step is a form of the FFT mode")
    cus = be.info()["num_cus"]
    n = ks.p.n
    mu = ol.to_torus(1, 8)
    rng = np.random.default_rng(77)
    monkeypatch.setenv("RS_NO_COOP8_LISTED", "1")          # read once, in rs_create
    with _scratch(ks, "default128") as old:
        monkeypatch.delenv("RS_NO_COOP8_LISTED")
        old.set_mode("fft")
        luts = _dev(rng.integers(-2**31, 2**31, (5, ks.p.N)).astype(np.int32))
        for B in (1, 9, cus):
            _, ca = _bits(ks, B, 500 + B)
            _, cb = _bits(ks, B, 600 + B)
            ca = ca.copy()
            ca[0, : n // 3] = 0                  # leading identity steps
            if B > 4:
                ca[1, n // 3: 2 * n // 3] = 0    # in the middle
                ca[2, -1 - n // 4:-1] = 0        # trailing (the b word stays)
                ca[3, 1:n:2] = 0                 # every other one
                ca[4, :n] = 0                    # the whole blind rotation is the identity: an empty step list
            da, db = _dev(ca), _dev(cb)
            got = be.bootstrap(da, mu)
            assert be.last_launch() == {"form": "coop8_listed", "waves_per_block": 8, "resident": 1}
            ref = old.bootstrap(da, mu)
            assert old.last_launch()["form"] == "coop8"
            assert torch.equal(got, ref), B
            assert torch.equal(be.gate("XNOR", da, db), old.gate("XNOR", da, db)), B
            assert torch.equal(be.bootstrap_lut(da, luts, first=3), old.bootstrap_lut(da, luts, first=3)), B
            sample = np.unique(np.r_[0:min(B, 6), B - 1])
            if fix == "full_default":
                sample = sample[:3] if B > 1 else sample     # the oracle takes a third of a second per full-key gate
            assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ca[sample], mu)), B
        assert old.fft_fallbacks() == 0
    assert be.fft_fallbacks() == 0


def test_host_calls_are_certified_without_fallbacks(be_toy_default, toy_default, arith_mode):
    """Every FFT-mode call records its rounding certificate and folds it into the stream's running maximum
    (include/redsec_hip.h RS_CERTIFICATE_LIMIT); no call needs the exact recomputation."""
    be = be_toy_default
    ks, ctx = toy_default
    _, ca = _bits(ks, 9, 1)
    _, cb = _bits(ks, 9, 2)
    be.rounding_certificate(reset=True)
    got = be.gate_host("XOR", ca, cb)
    assert np.array_equal(got, ctx.gate_batch("XOR", ca, cb))
    first = be.rounding_certificate(reset=False)
    be.gate_host("AND", ca, cb)
    assert be.rounding_certificate(reset=False) >= first          # a running maximum
    assert (first > 0) == (arith_mode == "fft") and first < 0.25
    assert be.fft_fallbacks() == 0


@pytest.mark.parametrize("which", ["be_full_default", "be_full_redsec"])
def test_structured_inputs_fft_equals_exact_mode(which, request):
    """Adversarially regular ciphertext words (all 0, all ~0, 0x80000000, alternating, one-hot masks): the
    FFT mode's rounded products must still equal the exact-NTT mode's word for word -- the key, not the
    input, randomises the products -- and the rounding certificate stays far from 1/2."""
    import torch
    be = request.getfixturevalue(which)
    n, W = be.p.n, be.W
    pats = [np.zeros(W, np.int64), np.full(W, -1), np.full(W, -2**31), np.full(W, 2**31 - 1),
            np.where(np.arange(W) % 2 == 0, 0x55555555, -0x55555556), (np.arange(W) * 0x01010101) & 0xFFFFFFFF]
    for hot in (0, n // 2, n - 1):
        v = np.zeros(W, np.int64); v[hot] = 1 << 21; pats.append(v)          # exactly one CMUX step with bara = 1
    x = np.stack([(p & 0xFFFFFFFF).astype(np.uint32).view(np.int32) for p in pats])
    x = np.concatenate([x, x[::-1] ^ 0x0F0F0F0F])
    d = _dev(x)
    prev = be.mode()
    try:
        be.set_mode("fft"); be.rounding_certificate(reset=True)
        a = be.bootstrap(d, 1 << 29); g = be.gate("XNOR", d, torch.flip(d, [0]).contiguous())
        cert = be.rounding_certificate(reset=True)
        be.set_mode("exact")
        assert torch.equal(a, be.bootstrap(d, 1 << 29))
        assert torch.equal(g, be.gate("XNOR", d, torch.flip(d, [0]).contiguous()))
    finally:
        be.set_mode(prev)
    assert cert < 0.2


def test_two_contexts_interleaved_on_a_side_stream(be_toy_default, toy_default, be_toy_redsec, toy_redsec):
    """Two contexts (different parameter sets, different keys) alive at once, their calls interleaved on a
    non-default HIP stream: every result still equals its own oracle's (no shared mutable state besides the
    per-context workspaces; the launch stream is the caller's)."""
    import torch
    ks_a, ctx_a = toy_default
    ks_b, ctx_b = toy_redsec
    _, a0 = _bits(ks_a, 70, 5); _, a1 = _bits(ks_a, 70, 6)
    _, b0 = _bits(ks_b, 41, 7); _, b1 = _bits(ks_b, 41, 8)
    side = torch.cuda.Stream()
    da0, da1, db0, db1 = _dev(a0), _dev(a1), _dev(b0), _dev(b1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ra = be_toy_default.gate("XOR", da0, da1)
        rb = be_toy_redsec.gate("ORYN", db0, db1)
        ra2 = be_toy_default.gate("NOR", ra, da1)         # consumes the first result on the same stream
        rb2 = be_toy_redsec.bootstrap(rb, ol.to_torus(1, 8))
    side.synchronize()
    ea = ctx_a.gate_batch("XOR", a0, a1)
    eb = ctx_b.gate_batch("ORYN", b0, b1)
    assert np.array_equal(ra.cpu().numpy(), ea) and np.array_equal(rb.cpu().numpy(), eb)
    assert np.array_equal(ra2.cpu().numpy(), ctx_a.gate_batch("NOR", ea, a1))
    assert np.array_equal(rb2.cpu().numpy(), ctx_b.bootstrap_batch(eb, ol.to_torus(1, 8)))


def test_linear_stage_matches_oracle(be_toy_redsec, toy_redsec):
    be = be_toy_redsec
    ks, _ = toy_redsec
    import torch
    rng = np.random.default_rng(8)
    K, M, W = 53, 17, ks.W
    x = rng.integers(-2**31, 2**31, (K, W)).astype(np.int32)
    sign = rng.integers(0, 2, (K, M)).astype(np.uint8)
    zero = (rng.random((K, M)) < 0.25).astype(np.uint8)
    zb = -(1 << 20)
    ref = ol.linear_fc(x, sign, zero, zb)
    got = be.linear_fc(_dev(x), torch.from_numpy(sign).cuda(), torch.from_numpy(zero).cuda(), zero_tap_b=zb).cpu().numpy()
    assert np.array_equal(got, ref)
    got = be.linear_fc(_dev(x), torch.from_numpy(sign).cuda(), None).cpu().numpy()
    assert np.array_equal(got, ol.linear_fc(x, sign, None, 0))
    # K-sliced form (few outputs: partial sums meet by integer atomics), one-slice form, bias on the b word
    for K2, M2 in ((1024, 10), (196, 300), (3, 5), (70, 1)):
        x2 = rng.integers(-2**31, 2**31, (K2, W)).astype(np.int32)
        sign2 = rng.integers(0, 2, (K2, M2)).astype(np.uint8)
        zero2 = (rng.random((K2, M2)) < 0.3).astype(np.uint8)
        bias2 = rng.integers(-2**31, 2**31, M2).astype(np.int32)
        ref2 = ol.linear_fc(x2, sign2, zero2, zb).astype(np.int64)
        ref2[:, -1] += bias2
        got2 = be.linear_fc(_dev(x2), torch.from_numpy(sign2).cuda(), torch.from_numpy(zero2).cuda(), zero_tap_b=zb,
                            bias_b=torch.from_numpy(bias2).cuda()).cpu().numpy()
        assert np.array_equal(got2, (ref2 & 0xFFFFFFFF).astype(np.uint32).view(np.int32)), (K2, M2)
    # lincomb: lweSubTo / lweAddMulTo shapes
    y = rng.integers(-2**31, 2**31, (K, W)).astype(np.int32)
    got = be.lincomb(_dev(x), 3, _dev(y), -1, bconst=12345).cpu().numpy()
    ref = (3 * x.astype(np.int64) - y.astype(np.int64))
    ref[:, -1] += 12345
    assert np.array_equal(got, (ref & 0xFFFFFFFF).astype(np.uint32).view(np.int32))


def test_output_noise_matches_cggi_theory(be_full_default, full_default):
    """KAT (4) of SURVEY.md section 8c: the noise of a bootstrapped gate is what the CGGI analysis
    predicts -- blind-rotate term n (k+1) l N E[d^2] sigma_bk^2 + keyswitch term N t E[...] sigma_ks^2.
    A wrong gadget, a dropped digit, a mis-scaled key row or a broken keyswitch all move this by far
    more than the factor 2 allowed here."""
    be = be_full_default
    ks, _ = full_default
    p = ks.p
    B = 4096
    ba, ca = _bits(ks, B, 31)
    bb, cb = _bits(ks, B, 32)
    out = be.gate("AND", _dev(ca), _dev(cb)).cpu().numpy()
    ph = ks.phase(out).astype(np.float64) / 2.0**32
    ideal = np.where((ba & bb) == 1, 0.125, -0.125)
    err = ph - ideal
    Bg = 1 << p.bk_Bgbit
    var_br = p.n * 2 * p.bk_l * p.N * (Bg * Bg / 12.0) * p.bk_stdev ** 2          # digits ~ uniform in [-Bg/2, Bg/2)
    var_br += p.n * (1 + p.N / 2.0) * (2.0 ** -(p.bk_l * p.bk_Bgbit + 1)) ** 2 / 3     # gadget rounding
    base = 1 << p.ks_basebit
    var_ks = p.N * p.ks_t * (1 - 1.0 / base) * p.lwe_stdev ** 2                       # rows actually subtracted
    var_ks += p.N / 2.0 * (2.0 ** -(p.ks_t * p.ks_basebit + 1)) ** 2 / 3               # keyswitch rounding
    theory = np.sqrt(var_br + var_ks)
    # RMS, not std: with ONE key the keyswitch rows' noises are fixed numbers, so part of the predicted
    # variance shows up as a key-dependent offset common to all ciphertexts
    measured = np.sqrt(np.mean(err ** 2))
    assert 0.5 * theory < measured < 2.0 * theory, (measured, theory)
    assert abs(err.mean()) < 2.0 * theory


def test_bench_batch_65536_default128_nands(be_full_default, full_default, arith_mode):
    """BASELINE configs[1] at its full size inside the test suite: 65,536 independent default-128 NANDs. Every output
    decrypts to NAND(a, b); the whole batch is identical in the other arithmetic mode, compared on the device; and
    256 gates drawn from the first, middle and last workgroups of the launch equal the oracle word for word."""
    import torch
    be = be_full_default
    ks, ctx = full_default
    B = 65536
    rng = np.random.default_rng(0xC0FFEE)
    e8 = ol.to_torus(1, 8)
    ba, bb = rng.integers(0, 2, B), rng.integers(0, 2, B)
    # fresh encryptions, vectorised (oracle_lib.KeySet.encrypt is a per-sample loop): a uniform, b = a.s + mu + e
    def enc(bits, seed):
        r = np.random.default_rng(seed)
        a = r.integers(-2**31, 2**31, (B, ks.p.n)).astype(np.int32)
        e = np.round(r.normal(0.0, ALPHA, B) * 2.0**32).astype(np.int64)
        dot = (a.astype(np.int64) * ks.lwe_key.astype(np.int64)).sum(axis=1)
        b = (dot + np.where(bits == 1, e8, -e8) + e) & 0xFFFFFFFF
        return np.concatenate([a, b.astype(np.uint32).view(np.int32)[:, None]], axis=1)
    ca, cb = enc(ba, 1), enc(bb, 2)
    da, db = _dev(ca), _dev(cb)
    out = be.gate("NAND", da, db)
    if arith_mode == "fft":
        assert be.last_launch() == {"form": "workgroup", "waves_per_block": 8, "resident": 8 * be.info()["num_cus"]}
    other = "exact" if arith_mode == "fft" else "fft"
    be.set_mode(other)
    assert torch.equal(be.gate("NAND", da, db), out)
    be.set_mode(arith_mode)
    got = out.cpu().numpy()
    phase = (got[:, -1].astype(np.int64) - (got[:, :-1].astype(np.int64) * ks.lwe_key.astype(np.int64)).sum(axis=1)) & 0xFFFFFFFF
    assert np.array_equal((phase < (1 << 31)).astype(int), 1 - (ba & bb))
    sample = np.r_[0:86, B // 2 - 42:B // 2 + 43, B - 85:B]
    assert len(sample) == 256
    assert np.array_equal(got[sample], ctx.gate_batch("NAND", ca[sample], cb[sample]))
    assert be.fft_fallbacks() == 0


@pytest.mark.parametrize("which,fix,form", [("be_full_default", "full_default", "workgroup"), ("be_full_redsec", "full_redsec", "workgroup"),
                                            ("be_full_redsec", "full_redsec", "duo")])
def test_workgroup_and_duo_kernels_on_full_keys_against_oracle(which, fix, form, request, arith_mode):
    """The throughput forms against the ORACLE on the real parameter sets (n = 630 / n = 350), not only on toy keys:
    a batch large enough to select the form, 32 sampled ciphertexts (first, middle and last groups, incl. the ragged one)."""
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    cus = be.info()["num_cus"]
    B = 8 * cus + 5 if form == "workgroup" else 3 * cus + 2
    rng = np.random.default_rng(31)
    mu = ol.to_torus(1, 4096)
    sample = np.r_[0:11, B // 2:B // 2 + 10, B - 11:B]
    ct = rng.integers(-2**31, 2**31, (B, ks.p.n + 1)).astype(np.int32)       # arbitrary words: the kernel is a function of them
    ct[sample] = ks.encrypt(rng.integers(-2**31, 2**31, len(sample)), ALPHA, 3)
    got = be.bootstrap(_dev(ct), mu)
    if arith_mode == "fft":
        assert be.last_launch()["form"] == form
    assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ct[sample], mu))


@pytest.mark.parametrize("which,fix", [("be_toy_default", "toy_default"), ("be_toy_redsec", "toy_redsec")])
def test_last_partial_round_runs_in_its_own_form(which, fix, request, monkeypatch):
    """A batch of one full round of the lock-step form plus a remainder of at most 4 x #CUs: the remainder is cut off and
    launched in the form its size would take by itself (cooperative / duo / half-size groups). Outputs -- also of the
    programmable bootstrap, whose test polynomial is chosen by the ciphertext's index in the WHOLE batch -- equal the
    unsplit launch (RS_NO_TAIL context) word for word and the oracle on a sample around the cut."""
    import torch
    be = request.getfixturevalue(which)
    ks, ctx = request.getfixturevalue(fix)
    cus = be.info()["num_cus"]
    mu = ol.to_torus(1, 8)
    rng = np.random.default_rng(12)
    monkeypatch.setenv("RS_NO_TAIL", "1")         # read once, in rs_create
    with _scratch(ks, "default128" if ks.p.bk_l == 3 else "redsec_small_v2") as be2:
        monkeypatch.delenv("RS_NO_TAIL")
        be2.set_mode(be.mode())
        luts = _dev(rng.integers(-2**31, 2**31, (7, ks.p.N)).astype(np.int32))
        for tail in (5, cus + 3, 3 * cus + 1):
            B = 8 * cus + tail
            _, ct = _bits(ks, B, 900 + tail)
            d = _dev(ct)
            got, ref = be.bootstrap(d, mu), be2.bootstrap(d, mu)
            assert torch.equal(got, ref), tail
            assert torch.equal(be.bootstrap_lut(d, luts), be2.bootstrap_lut(d, luts)), tail
            sample = np.r_[0:4, 8 * cus - 4:8 * cus + 4, B - 4:B]
            assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ct[sample], mu)), tail


def test_rccl_gather_world_size_one():
    """backend="nccl" (= RCCL) has to have run once before a multi-GPU scaling run depends on it: one rank on this box, in a
    clean child process (tests/rccl_world1.py), drives OverlappedGather.launch/wait (async_op on RCCL's stream) and
    all_gather_rows on device tensors around real gate steps; the gathered blocks equal the oracle word for word."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("set_name", ["redsec_small_v2", "default128", "redsec_small"])
def test_sliced_keyswitch_two_step_equals_the_atomics_form_and_the_oracle(monkeypatch, set_name):
    """lweKeySwitch at the batch sizes of the latency forms, on the three tiled key shapes at their full size -- REDsec's shipped
    set (N = 1024, t = 9, basebit = 3, W = 351: 11 word blocks, 64 ... 4 input slices; one LDS lookup per digit), default-128
    (t = 8, basebit = 2, W = 631) and redsec_params_small (t = 18, basebit = 1, W = 501), the two shapes of the combined-digit
    kernel (keyswitch_tiled_comb_kernel: sums of 2 / 4 digits' rows built per workgroup in LDS): the two-step form (every slice leaves its partial sums in a
    per-stream scratch laid out [slice][word][ciphertext], keyswitch_reduce_kernel adds them up and transposes) against the
    round-1 form (RS_KS_ATOMICS=1: the slices meet by integer atomics in a zeroed output) on WHOLE batches, and against the
    oracle's lweKeySwitch on sampled rows. Batch sizes on both sides of every slice-count change, ragged against the 256-lane
    workgroups and the 16 x 16 reduce tiles. Synthetic keys (generated on the device; the oracle restates the generator)."""
    import torch
    import redsec_amd
    seed = 0xabc123
    p = ol.params(set_name)
    W, N = p.n + 1, p.N

    def backend():
        be = redsec_amd.Backend(redsec_amd.params(set_name), device=0)
        be.load_synthetic_keys(seed)
        return be
    with contextlib.ExitStack() as stack:            # both contexts are closed also when an assertion fails
        be = stack.enter_context(POOL.scratch(backend))
        monkeypatch.setenv("RS_KS_ATOMICS", "1")          # read once, in rs_create
        be_atomics = stack.enter_context(POOL.scratch(backend))
        monkeypatch.delenv("RS_KS_ATOMICS")
        _sliced_keyswitch_body(be, be_atomics, p, seed)


def _sliced_keyswitch_body(be, be_atomics, p, seed):
    import torch
    W, N = p.n + 1, p.N

    class K:
        pass
    ks = K(); ks.p = p
    ks.bk = np.zeros(ol.lib().ro_bk_words(ol.C.byref(p)), np.int32)        # never read: only the keyswitch runs on the oracle
    ks.ksk = ol.synthetic_key_words(seed ^ 0x6b73, N * p.ks_t * (1 << p.ks_basebit) * W)
    ctx = ol.Ctx(ks)
    rng = np.random.default_rng(3)
    for B in (1, 15, 17, 196, 255, 257, 600, 1024, 1500, 2048, 3000, 6000):
        u = rng.integers(-2**31, 2**31, (B, N + 1), dtype=np.int32)
        u[0, :7] = 0                                   # zero digits select the resident zero row
        d = _dev(u)
        got = be.keyswitch(d)
        assert torch.equal(got, be_atomics.keyswitch(d)), B
        rows = np.unique(np.r_[0, B // 2, B - 1, rng.integers(0, B, 5)])
        assert np.array_equal(got.cpu().numpy()[rows], ctx.keyswitch(u[rows])), B
    ctx.close()
