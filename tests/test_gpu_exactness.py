"""Exactness is a property of the product (VERDICT r1 item 1 / ADVICE): in FFT mode every bootstrapped call -- the
asynchronous *_dev ones included -- is followed on its stream by the exact-NTT kernels gated on the call's rounding
certificate. Forcing the limit to 0 makes every call take that recomputation, which must still equal the oracle word
for word; and one context serves several streams at once (per-stream workspace, counter, certificate slots)."""
import threading

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ALPHA = 2.0 ** -15


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x, np.int32)).cuda()


def _bits(ks, B, seed):
    rng = np.random.default_rng(seed)
    e8 = ol.to_torus(1, 8)
    bits = rng.integers(0, 2, B)
    return bits, ks.encrypt(np.where(bits == 1, e8, -e8), ALPHA, seed)


def _backend(ks, name):
    import redsec_amd
    be = redsec_amd.Backend(redsec_amd.params(name, n=ks.p.n), device=0)
    be.load_keys(ks.bk, ks.ksk)
    return be


@pytest.mark.parametrize("fix,name", [("toy_default", "default128"), ("toy_redsec", "redsec_small_v2")])
def test_forced_exact_recomputation_on_the_async_path(fix, name, request):
    import torch
    ks, ctx = request.getfixturevalue(fix)
    be = _backend(ks, name)
    assert be.mode() == "fft"
    cus = be.info()["num_cus"]
    mu = ol.to_torus(1, 8)
    for B in (5, 3 * cus + 1, 8 * cus + 3):          # cooperative, duo / per-wave, lock-step workgroup forms
        _, ca = _bits(ks, B, 10 + B)
        _, cb = _bits(ks, B, 11 + B)
        da, db = _dev(ca), _dev(cb)
        n0 = be.certify()[1]                         # the recomputed-call count is cumulative
        normal = be.gate("NAND", da, db)
        dist, n = be.certify()
        assert 0 < dist < 0.2 and n == n0            # certified, nothing recomputed
        be.set_certificate_limit(0.0)                # every call's certificate now "fails"
        forced = be.gate("NAND", da, db)             # *_dev call: no host round trip in between
        lut = torch.full((1, ks.p.N), int(mu), dtype=torch.int32, device="cuda")
        forced_lut = be.bootstrap_lut(da, lut)
        forced_mux = be.mux(da, db, da)
        dist, n = be.certify()
        assert n == n0 + 3 and be.fft_fallbacks() == n   # three calls were recomputed by the exact-NTT kernels
        be.set_certificate_limit(0.25)
        assert torch.equal(normal, forced)
        sample = np.r_[0:min(B, 6), max(0, B - 5):B]
        assert np.array_equal(forced.cpu().numpy()[sample], ctx.gate_batch("NAND", ca[sample], cb[sample]))
        assert np.array_equal(forced_lut.cpu().numpy()[sample], ctx.bootstrap_batch(ca[sample], mu))
        assert np.array_equal(forced_mux.cpu().numpy()[sample], ctx.mux_batch(ca[sample], cb[sample], ca[sample]))
        # host-pointer calls go through the same path
        be.set_certificate_limit(0.0)
        assert np.array_equal(be.gate_host("XOR", ca[:4], cb[:4]), ctx.gate_batch("XOR", ca[:4], cb[:4]))
        be.set_certificate_limit(0.25)
        be.certify()
    be.close()


def test_two_streams_and_two_threads_on_one_context(toy_default):
    """Calls on different streams of ONE context run concurrently, issued from two host threads: each stream has its
    own extracted-sample workspace, work counter and certificate slots (include/redsec_hip.h "Streams")."""
    import torch
    ks, ctx = toy_default
    be = _backend(ks, "default128")
    cus = be.info()["num_cus"]
    Bs = (8 * cus + 5, 2 * cus + 7)                   # persistent/lock-step form on one stream, per-wave form on the other
    inputs = [(_bits(ks, B, 50 + i)[1], _bits(ks, B, 60 + i)[1]) for i, B in enumerate(Bs)]
    dev = [(_dev(a), _dev(b)) for a, b in inputs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [None, None]
    errs = []

    def work(i):
        try:
            with torch.cuda.stream(streams[i]):
                for _ in range(3):                    # several calls back to back on each stream
                    outs[i] = be.gate("NAND" if i == 0 else "XOR", dev[i][0], dev[i][1])
                d, n = be.certify()
                assert 0 < d < 0.2 and n == 0
        except Exception as e:                        # surfaced below: an assert in a thread is silent otherwise
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    torch.cuda.synchronize()
    for i, op in enumerate(("NAND", "XOR")):
        a, b = inputs[i]
        sample = np.r_[0:6, Bs[i] - 6:Bs[i]]
        assert np.array_equal(outs[i].cpu().numpy()[sample], ctx.gate_batch(op, a[sample], b[sample])), op
        with torch.cuda.stream(streams[i]):           # and the same call alone gives the same words everywhere
            again = be.gate(op, dev[i][0], dev[i][1])
        streams[i].synchronize()
        assert torch.equal(again, outs[i])
    be.close()


def test_kernel_form_is_reported_per_launch(toy_redsec):
    ks, _ = toy_redsec
    be = _backend(ks, "redsec_small_v2")
    cus = be.info()["num_cus"]
    mu = ol.to_torus(1, 4096)
    seen = {}
    for B in (3, cus + 9, 3 * cus, 8 * cus):
        _, ct = _bits(ks, B, B)
        be.bootstrap(_dev(ct), mu)
        seen[B] = be.last_launch()
    assert seen[3]["form"] == "coop8" and seen[3]["waves_per_block"] == 8 and seen[cus + 9]["form"] == "coop2"
    assert seen[3 * cus]["form"] == "duo" and seen[3 * cus]["resident"] == 3 * cus
    assert seen[8 * cus]["form"] == "workgroup" and seen[8 * cus]["resident"] == 8 * cus and seen[8 * cus]["waves_per_block"] == 8
    assert be.info()["waves_per_block"] == 8
    be.close()


@pytest.mark.parametrize("mode", ["fft", "split"])
@pytest.mark.parametrize("fix,name", [("toy_default", "default128"), ("toy_redsec", "redsec_small_v2")])
def test_xcd_cohort_table_after_a_lock_step_launch(fix, name, mode, request, monkeypatch):
    """The XCD cohort step of the lock-step kernels is one asm block (rs_cohort.h) that computes the table addresses itself:
    after a launch every workgroup must have left kCohortGone + the CMUX steps it walked in ITS entry (xcd = blockIdx & 7,
    slot = blockIdx >> 3) and no other entry may have been touched; the words must equal those of a context that runs without
    cohorts (RS_NO_COHORT) and, on a sample, the oracle's."""
    import torch
    ks, ctx = request.getfixturevalue(fix)
    be = _backend(ks, name)
    be.set_mode(mode)
    cus, n = be.info()["num_cus"], ks.p.n
    B = 16 * cus + 40   # FFT mode: the 40 are cut off into a form of their own; split mode: five workgroups walk a third group
    _, ct = _bits(ks, B, 77)
    mu = ol.to_torus(1, 8)
    got = be.bootstrap(_dev(ct), mu)
    assert be.last_launch()["form"] == ("workgroup" if mode == "fft" else "split_workgroup")
    table = be.cohort_table()
    expected = np.full((8, 64), 0x7F7F7F7F, np.int32)
    groups = 2 * cus if mode == "fft" else 2 * cus + 5
    for b in range(cus):
        expected[b & 7, b >> 3] = 0x40000000 + n * len(range(b, groups, cus))
    assert np.array_equal(table, expected)
    monkeypatch.setenv("RS_NO_COHORT", "1")
    be2 = _backend(ks, name)
    monkeypatch.delenv("RS_NO_COHORT")
    be2.set_mode(mode)
    assert torch.equal(be2.bootstrap(_dev(ct), mu), got)
    sample = np.r_[0:4, 8 * cus - 2:8 * cus + 2, B - 4:B]
    assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ct[sample], mu))
    be2.close()
    be.close()


def test_cohort_table_is_left_alone_by_launches_without_cohorts(toy_default):
    """Round-4 advisor finding: the half-size lock-step launch (default-128, 2 x #CUs < B <= 4 x #CUs: blind_rotate_wg_kernel<.., 4>)
    used to reach the kernel with a non-null progress table and cohort_every = cohort_lag = 0 -- the workgroups then compared against
    stale entries (48 polls each) and wrote their step counts into a table no launch had cleared. cohort_setup is now the only
    place that hands a kernel the table: after a launch WITH cohorts the table holds that launch's entries, and a launch without
    cohorts behind it (B = 3 x #CUs) must leave every word of it as it was -- and still equal the oracle."""
    ks, ctx = toy_default
    be = _backend(ks, "default128")
    cus, n = be.info()["num_cus"], ks.p.n
    mu = ol.to_torus(1, 8)
    _, ct = _bits(ks, 16 * cus, 78)
    be.bootstrap(_dev(ct), mu)
    assert be.last_launch()["form"] == "workgroup" and be.last_launch()["waves_per_block"] == 8
    before = be.cohort_table()
    if cus == 256:                      # cohorts run on the whole chip as one partition only (8 XCDs, round-robin dispatch)
        assert (before[:, :cus // 8] == 0x40000000 + 2 * n).all()
    B = 3 * cus
    _, ct3 = _bits(ks, B, 79)
    got = be.bootstrap(_dev(ct3), mu)
    assert be.last_launch()["form"] == "workgroup" and be.last_launch()["waves_per_block"] == 4
    assert np.array_equal(be.cohort_table(), before)
    sample = np.r_[0:6, B // 2 - 3:B // 2 + 3, B - 6:B]
    assert np.array_equal(got.cpu().numpy()[sample], ctx.bootstrap_batch(ct3[sample], mu))
    be.close()
