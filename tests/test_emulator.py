"""CPU parity of the kernel logic: the HIP phase functions (redsec_amd/csrc/rs_ntt.h) executed
lane-by-lane on the host must reproduce the oracle bit-for-bit. This checks the transform's index
mapping, the exactness of the FP64 field arithmetic and the CMUX bookkeeping without a GPU."""
import ctypes

import numpy as np
import pytest

import emu_lib
import oracle_lib as ol

ALPHA = 2.0 ** -15


@pytest.mark.parametrize("cfg", [0, 1])
def test_schedule_is_provably_exact(cfg):
    rc, msg = emu_lib.validate(cfg)
    assert rc == 0, msg
    p = emu_lib.lib().rs_emu_prime(cfg)
    l, bg = (3, 7) if cfg == 0 else (10, 3)
    bound = 2 * l * 1024 * (1 << (bg - 1)) * (1 << 31)
    assert p % 2048 == 1 and p > 2 * bound


@pytest.mark.parametrize("cfg,half", [(0, 64), (1, 4)])
def test_polymul_matches_schoolbook(cfg, half):
    rng = np.random.default_rng(10 + cfg)
    cases = []
    for _ in range(6):
        cases.append((rng.integers(-half, half, 1024), rng.integers(-2**31, 2**31, 1024)))
    cases.append((np.full(1024, -half), np.full(1024, -2**31)))                      # largest magnitudes
    cases.append((np.full(1024, half - 1), np.full(1024, 2**31 - 1)))
    alt = np.where(np.arange(1024) % 2 == 0, -half, half - 1)
    cases.append((alt, np.where(np.arange(1024) % 3 == 0, -2**31, 2**31 - 1)))
    e = np.zeros(1024); e[1023] = -half
    f = np.zeros(1024); f[1023] = -2**31
    cases.append((e, f))
    cases.append((np.zeros(1024), rng.integers(-2**31, 2**31, 1024)))
    for a, b in cases:
        a = a.astype(np.int32); b = b.astype(np.int32)
        assert np.array_equal(emu_lib.polymul(cfg, a, b), ol.negacyclic_mul(a, b, "schoolbook"))


@pytest.mark.parametrize("cfg", [0, 1])
def test_forward_transform_is_linear_mod_p(cfg):
    p = int(emu_lib.lib().rs_emu_prime(cfg))
    rng = np.random.default_rng(cfg)
    a = rng.integers(-2**20, 2**20, 1024).astype(np.int32)
    b = rng.integers(-2**20, 2**20, 1024).astype(np.int32)
    fa = emu_lib.forward(cfg, a); fb = emu_lib.forward(cfg, b); fs = emu_lib.forward(cfg, a + b)
    for x, y, z in zip(fa[:64], fb[:64], fs[:64]):
        assert (int(x) + int(y) - int(z)) % p == 0
    assert np.all(np.abs(fa) < 2.0 ** 53) and np.all(fa == np.rint(fa))


@pytest.mark.parametrize("cfg,l,bg", [(0, 3, 7), (1, 10, 3)])
def test_fused_digit_stages_equal_generic_transform(cfg, l, bg):
    """fwd_F1_digits (table / exact-FMA stages 0-1) == generic transform of the digit polynomial
    (mod p), for random coefficients and for coefficients whose digits are all extreme."""
    p = int(emu_lib.lib().rs_emu_prime(cfg))
    offset = sum((1 << (bg - 1)) << (32 - i * bg) for i in range(1, l + 1)) & 0xFFFFFFFF
    rng = np.random.default_rng(cfg)
    lo_digits = np.uint32((0 - offset) & 0xFFFFFFFF)                       # every digit = -Bg/2
    hi_digits = np.uint32((((1 << (bg * l)) - 1) << (32 - bg * l)) - offset & 0xFFFFFFFF)   # every digit = Bg/2-1
    cases = [rng.integers(-2**31, 2**31, 1024).astype(np.int32),
             np.full(1024, lo_digits, np.uint32).view(np.int32), np.full(1024, hi_digits, np.uint32).view(np.int32),
             np.where(np.arange(1024) % 2 == 0, lo_digits, hi_digits).astype(np.uint32).view(np.int32)]
    for coef in cases:
        for q in (0, l - 1):
            u = (coef.view(np.uint32).astype(np.uint64) + offset) & 0xFFFFFFFF
            digit = ((u >> (32 - (q + 1) * bg)) & ((1 << bg) - 1)).astype(np.int64) - (1 << (bg - 1))
            fused = emu_lib.forward_digits(cfg, coef, q)
            generic = emu_lib.forward(cfg, digit.astype(np.int32))
            assert np.all(np.abs(fused) < 2.0 ** 53)
            assert all((int(a) - int(b)) % p == 0 for a, b in zip(fused, generic))


@pytest.mark.parametrize("cfg,fixture", [(0, "toy_default"), (1, "toy_redsec")])
def test_blind_rotate_matches_oracle(cfg, fixture, request):
    ks, ctx = request.getfixturevalue(fixture)
    p = ks.p
    mu = ol.to_torus(1, 8)
    msgs = [mu, -mu, mu]
    ca = ks.encrypt(msgs, ALPHA, 11)
    cb = ks.encrypt(msgs[::-1], ALPHA, 12)
    ref_u = ctx.bootstrap_wo_ks(ca, mu)
    ref_acc = ctx.blind_rotate_acc(ca, mu, steps=5)
    for i in range(len(msgs)):
        u, _ = emu_lib.blind_rotate(cfg, p.n, ca[i], None, 1, 0, 0, mu, ks.bk)
        assert np.array_equal(u, ref_u[i])
        _, acc = emu_lib.blind_rotate(cfg, p.n, ca[i], None, 1, 0, 0, mu, ks.bk, steps=5)
        assert np.array_equal(acc, ref_acc[i])
    # fused gate pre-combination: XOR = (0,1/4) + 2(a+b)
    tmp = ol.gate_precombine("XOR", ca, cb)
    ref_u = ctx.bootstrap_wo_ks(tmp, mu)
    for i in range(len(msgs)):
        u, _ = emu_lib.blind_rotate(cfg, p.n, ca[i], cb[i], 2, 2, ol.to_torus(1, 4), mu, ks.bk)
        assert np.array_equal(u, ref_u[i])


def test_blind_rotate_sign_mu_4096(toy_redsec):
    ks, ctx = toy_redsec
    mu = ol.to_torus(1, 4096)
    ct = ks.encrypt([ol.to_torus(m, 4096) for m in (-700, 3, 900)], ALPHA, 31)
    ref = ctx.bootstrap_wo_ks(ct, mu)
    for i in range(3):
        u, _ = emu_lib.blind_rotate(1, ks.p.n, ct[i], None, 1, 0, 0, mu, ks.bk)
        assert np.array_equal(u, ref[i])


@pytest.mark.parametrize("half", [64, 4])
def test_fft_mode_polymul_is_exact_after_rounding(half):
    """rs_fft.h: the folded complex FFT, rounded, equals the exact negacyclic product; the distance
    to the nearest integer before rounding stays orders of magnitude below 1/2 -- also for inputs of
    the largest possible magnitude."""
    rng = np.random.default_rng(half)
    cases = [(rng.integers(-half, half, 1024), rng.integers(-2**31, 2**31, 1024)) for _ in range(8)]
    cases.append((np.full(1024, -half), np.full(1024, -2**31)))
    cases.append((np.where(np.arange(1024) % 2 == 0, -half, half - 1), np.where(np.arange(1024) % 3 == 0, -2**31, 2**31 - 1)))
    for a, b in cases:
        a = a.astype(np.int32); b = b.astype(np.int32)
        out, dev = emu_lib.polymul_fft(a, b)
        assert np.array_equal(out, ol.negacyclic_mul(a, b, "schoolbook"))
        assert dev < 0.1


def test_fft_twiddle_literals_match_generated_table():
    """rs_fft.h kFftTwU (stages 0-2, scalar-register literals on the device) equals the host-generated
    table, and every odd table entry is exactly i times its even sibling (the device fetches only the
    even ones)."""
    assert emu_lib.lib().rs_emu_fft_twiddle_check() == 0


@pytest.mark.parametrize("logn", [10, 11, 12, 13])
def test_general_ring_exchanges_keep_every_wavefront_in_its_own_region(logn):
    """rs_general.h: the exchanges whose groups fit one wavefront run without workgroup barriers. That is only safe if every
    exchange maps the 512 values of wavefront w to the same slots [576 w, 576 (w + 1)) -- the region it reads in the previous
    exchange and alone writes in the next. (With the exchanges of reader parameter >= 8 left unpadded, as they were for most of
    round 3, the barrier-free stores of exchange 1 overwrote slots another wavefront could still be reading as exchange 0's
    data at N = 4096 / 8192: a rare wrong product. This check fails on that layout.)"""
    assert emu_lib.lib().rs_emu_gen_wave_region_violations(logn) == 0


def test_general_ring_pass0_literals_match_generated_tables():
    """rs_general.h gen_pass_tw: pass 0 of every general ring takes its even twiddles (table entries 1, 2, 4, 6) from the
    literals of rs_fft.h instead of the table; they must be the table's values bit for bit, for N = 1024 ... 8192."""
    assert emu_lib.lib().rs_emu_gen_literal_twiddle_check() == 0


def test_fft_planar_exchange_is_bit_identical_to_interleaved():
    """The workgroup kernel moves the re and im planes through one half-size LDS buffer in turn
    (rs_fft.h fpl_exchange); same data movement, so products AND rounding distances are identical."""
    rng = np.random.default_rng(77)
    try:
        for _ in range(4):
            a = rng.integers(-64, 64, 1024).astype(np.int32); b = rng.integers(-2**31, 2**31, 1024).astype(np.int32)
            emu_lib.set_planar(False)
            o0, d0 = emu_lib.polymul_fft(a, b)
            emu_lib.set_planar(True)
            o1, d1 = emu_lib.polymul_fft(a, b)
            assert np.array_equal(o0, o1) and d0 == d1
            assert np.array_equal(o1, ol.negacyclic_mul(a, b, "schoolbook"))
    finally:
        emu_lib.set_planar(False)


@pytest.mark.parametrize("cfg,fixture", [(0, "toy_default"), (1, "toy_redsec")])
def test_fft_mode_blind_rotate_matches_exact_oracle(cfg, fixture, request):
    ks, ctx = request.getfixturevalue(fixture)
    mu = ol.to_torus(1, 8)
    ct = ks.encrypt([mu, -mu, mu, -mu], ALPHA, 21)
    ref = ctx.bootstrap_wo_ks(ct, mu)
    for i in range(4):
        u, _, dev = emu_lib.blind_rotate_fft(cfg, ks.p.n, ct[i], None, 1, 0, 0, mu, ks.bk)
        assert np.array_equal(u, ref[i])
        assert dev < 0.05


@pytest.mark.parametrize("cfg", [0, 1])
def test_signed_field_digits_equal_tfhe_decomposition(cfg):
    """FFT-mode kernels read each gadget digit as a signed bit field of (d + offset) ^ offset (one
    v_bfe_i32); it must equal tGswTorus32PolynomialDecompH's ((d + offset) >> decal & mask) - Bg/2
    for every digit level: an odd stride visits 2^24 values spread over the whole word, plus both
    ends of the range and the carry boundaries around the offset."""
    assert emu_lib.digit_mismatches(cfg, 0, 0x01000193, 1 << 24) == 0
    assert emu_lib.digit_mismatches(cfg, 0, 1, 1 << 16) == 0
    assert emu_lib.digit_mismatches(cfg, 0xFFFF0000, 1, 1 << 17) == 0
    assert emu_lib.digit_mismatches(cfg, 0x7FFF0000, 1, 1 << 17) == 0


# ---- general ring path (csrc/rs_general.h): N = 1024 ... 8192, split key ----
@pytest.mark.parametrize("logn", [10, 11, 12, 13])
def test_general_path_layouts(logn):
    """Every register of every exchange sits at register 0's position plus the compile-time offset the device code
    uses, inside the padded plane, and each 32-lane group's 8-byte accesses fall into 32 different bank pairs."""
    assert emu_lib.lib().rs_emu_gen_layout_violations(logn) == 0


def test_planar_exchange_positions_are_conflict_free():
    """rs_fft.h, 8-byte stores / 16-byte loads: the device's own position functions, all four exchange directions -- every value
    comes back in the reader's layout, the reader's register pairs are adjacent and aligned, and every store / load
    wave-instruction is bank-conflict free in the lane groups MI355X_MICROARCH.md gives for ds_write_b64 / ds_read_b128."""
    assert emu_lib.lib().rs_emu_plane_layout_violations() == 0


@pytest.mark.parametrize("logn,half", [(10, 512), (11, 64), (12, 512), (13, 512)])
def test_general_path_split_product_is_exact(logn, half):
    N = 1 << logn
    rng = np.random.default_rng(logn)
    a = rng.integers(-half, half, N).astype(np.int32)
    b = rng.integers(-2**31, 2**31, N).astype(np.int32)
    out, dev = emu_lib.gen_polymul(logn, a, b)
    assert np.array_equal(out, ol.negacyclic_mul(a, b, "ntt"))
    # operands of the largest 2-norm the bound allows for (random signs: the worst case for rounding)
    a = np.where(rng.integers(0, 2, N) == 1, half - 1, -half).astype(np.int32)
    b = np.where(rng.integers(0, 2, N) == 1, 0x7fff7fff, -0x80008000).astype(np.int32)
    out, dev2 = emu_lib.gen_polymul(logn, a, b)
    assert np.array_equal(out, ol.negacyclic_mul(a, b, "ntt"))
    bgbit = int(np.log2(half)) + 1
    bound = emu_lib.lib().rs_emu_gen_error_bound(logn, 1, bgbit)      # l = 1: two rows; one product is half of that
    assert max(dev, dev2) < bound < 0.25


def test_general_path_bound_covers_every_reference_parameter_set():
    """The split-key bound DERIVED in rs_general.h (no quoted theorem): below 1/2 -- rounding exact for every input -- for all
    five sets the reference defines (client/gen_secure_keyset.cpp:9-91 and TFHE's default), below 1/4 up to N = 4096."""
    want = {(10, 3, 7): 0.00135, (10, 10, 3): 0.0003, (10, 3, 10): 0.0108, (12, 3, 10): 0.098, (13, 3, 10): 0.296}
    for (logn, l, bgbit), v in want.items():
        b = emu_lib.lib().rs_emu_gen_error_bound(logn, l, bgbit)
        assert b < 0.5 and abs(b - v) < 0.03 * v + 1e-5, (logn, l, bgbit, b)
        if logn <= 12:
            assert b < 0.25


@pytest.mark.parametrize("logn", [10, 12, 13])
def test_general_path_transform_error_constants(logn):
    """Sanity check of the analysis' per-stage constants: the measured 2-norm error of one forward / one inverse transform
    (device butterflies, emulated; 80-bit reference) stays below g_f - 1 / g_i - 1, for random and for extreme-magnitude input."""
    import ctypes as C
    for seed, amp in ((1, 512), (2, 4), (3, 32768)):
        got, bound = (C.c_double * 2)(), (C.c_double * 2)()
        assert emu_lib.lib().rs_emu_gen_transform_errors(logn, seed, amp, got, bound) == 0
        assert 0 < got[0] < bound[0] and 0 < got[1] < bound[1], (logn, seed, list(got), list(bound))


@pytest.mark.parametrize("l,bgbit", [(3, 10), (3, 7), (10, 3), (2, 16), (4, 8)])
def test_general_path_digits_equal_tfhe_decomposition(l, bgbit):
    f = emu_lib.lib().rs_emu_gen_digit_mismatches
    assert f(l, bgbit, 0x7ffffff0, 1, 64) == 0 and f(l, bgbit, 0xfffffff0, 1, 64) == 0      # the wrap-around boundaries
    assert f(l, bgbit, 12345, 2654435761, 1 << 18) == 0


# form ids of rs_emu_lds_protocol_conflicts (2 and 3 were coop8's s_part exchange, removed with the switch that selected it)
LDS_FORMS = {0: "coop<2>", 1: "coop<4>", 4: "coops<2>", 5: "coops<4>", 6: "duo", 7: "duos", 8: "wgs<8>", 9: "wgs<4>", 10: "wg<8>",
             11: "coop8: sums by LDS atomics (l = 10)", 12: "coop8, listed step: shared rotated difference, sums by LDS atomics (l = 3)",
             13: "keyswitch, one lookup per digit (rows stored behind the lookups)", 14: "keyswitch, combined digits (base rows -> sums -> lookups)"}
# perturbations every form's model knows (rs_emulate.cpp): 1 = one placement / slot-count / hold decision changed the way a
# plausible edit would change it, 2 = one workgroup barrier dropped, 3 = (duo) the next quad requested before the swap / (coop8, listed step) the
# barrier between building the shared rotated difference and the rows that read it dropped
LDS_BROKEN = {0: (1, 2), 1: (1, 2), 4: (1, 2), 5: (1, 2), 6: (1, 2, 3), 7: (1, 2), 8: (1,), 9: (1,), 10: (2,), 11: (1, 2), 12: (1, 2, 3), 13: (1, 2), 14: (1, 2)}


@pytest.mark.parametrize("form", sorted(LDS_FORMS), ids=[LDS_FORMS[k] for k in sorted(LDS_FORMS)])
def test_lds_protocols_of_the_n1024_forms_have_no_cross_wave_conflict(form):
    """Every place where the waves of an N = 1024 blind-rotation workgroup hand data to each other through LDS -- the partial
    column sums of the cooperative forms (coop, coops, coop8), the partials the two waves of a ciphertext swap through the idle
    key buffer (duo, duos), the key (half-)rows that arrive by direct global->LDS loads into ring slots (wg, wgs, duo, duos) --
    replayed epoch by epoch (an epoch = between two workgroup barriers) with the placement functions the kernels themselves
    call (csrc/rs_lds_plan.h): no two different waves touch overlapping bytes in one epoch unless both read (or both are LDS
    atomics). Each model must ALSO see a deliberately broken protocol: the perturbed variants count > 0 (what the round-3
    suite lacked: the general kernels' exchange race passed it nine runs in ten; its host check is
    test_general_ring_exchanges_keep_every_wavefront_in_its_own_region). The planar exchange of a transform PAIR
    (fft_exchange_over_keep: two transforms through one per-wave buffer) involves a single wavefront, whose LDS operations
    execute in order; its addresses are covered by test_fft_planar_*."""
    L = emu_lib.lib()
    L.rs_emu_lds_protocol_conflicts.restype = ctypes.c_long
    assert L.rs_emu_lds_protocol_conflicts(form, 0) == 0
    for broken in LDS_BROKEN[form]:
        assert L.rs_emu_lds_protocol_conflicts(form, broken) > 0, (LDS_FORMS[form], broken)


def test_coop8_listed_step_shares_of_the_rotated_difference_cover_every_coefficient_once():
    L = emu_lib.lib()
    L.rs_emu_coop8_diff_cover_violations.restype = ctypes.c_long
    assert L.rs_emu_coop8_diff_cover_violations() == 0


def test_coop8_row_split_covers_every_row_once_and_balances_the_simds():
    L = emu_lib.lib()
    L.rs_emu_coop8_row_split_violations.restype = ctypes.c_long
    for l in range(1, 17):
        assert L.rs_emu_coop8_row_split_violations(l) == 0, l


def test_keyswitch_slicing_rule():
    """rs_host.h keyswitch_slices: how a small batch's keyswitch is cut into input-coefficient slices (each slice stores its partial
    sums to a scratch of slices x W x B words; keyswitch_reduce_kernel adds them). Properties the kernels rely on: a power of two
    that divides the N / IG staging groups of every tiled shape (IG = 2 or 4), at most 64, never more slices than needed for
    ~1,024 workgroups, one slice (the plain-store throughput form, no scratch) for the large batches, monotone in B."""
    L = emu_lib.lib()
    L.rs_emu_keyswitch_slices.restype = ctypes.c_long
    L.rs_emu_keyswitch_scratch_words.restype = ctypes.c_long
    for W, N in ((351, 1024), (631, 1024), (501, 1024), (3073, 4096), (6145, 8192), (25, 1024)):
        prev = 64
        for B in (1, 2, 15, 196, 256, 257, 600, 1024, 2048, 4096, 8192, 20000, 32768, 65536, 131072):
            s = L.rs_emu_keyswitch_slices(B, W, N)
            gx, gy = (B + 255) // 256, (W + 31) // 32
            assert s in (1, 2, 4, 8, 16, 32, 64) and (N // 4) % s == 0 and N // (2 * s) >= 2
            assert s == 1 or gx * gy * (s // 2) < 1024                       # the last doubling was needed
            assert s == 64 or gx * gy * s >= 1024 or N // (2 * s) < 4       # and it stopped for a reason
            assert s <= prev
            prev = s
            assert L.rs_emu_keyswitch_scratch_words(B, W, N) == (s * W * B if s > 1 else 0)
        if W >= 351:                                                          # (a toy width has a single word block: even 65,536 ciphertexts are sliced)
            assert L.rs_emu_keyswitch_slices(65536, W, N) == 1
    assert L.rs_emu_keyswitch_slices(196, 351, 1024) == 64 and L.rs_emu_keyswitch_slices(1024, 351, 1024) == 32


def test_keyswitch_combined_digit_index_recovers_every_digit():
    """keyswitch_tiled_comb_kernel looks up ONE row per D digits: the row index of a group must decompose into exactly the digits the
    per-digit kernel extracts from the same word, in the order the table builder adds the base rows up (rs_host.h: ks_comb_index /
    ks_comb_digit, the functions the kernel calls), and stay inside the group's table -- for the three shipped key shapes and the
    D each one could use."""
    L = emu_lib.lib()
    L.rs_emu_ks_comb_violations.restype = ctypes.c_long
    L.rs_emu_ks_comb_violations.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    rng = np.random.default_rng(5)
    words = [0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 1] + [int(x) for x in rng.integers(0, 2**32, 300)]
    for t, basebit, ds in ((8, 2, (1, 2, 3, 4)), (18, 1, (1, 2, 4, 5)), (9, 3, (1, 2, 3))):
        for D in ds:
            for w in words:
                assert L.rs_emu_ks_comb_violations(w, t, basebit, D) == 0, (t, basebit, D, hex(w))
