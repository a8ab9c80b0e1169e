"""BASELINE configs[3] shape through the Python layer chain (redsec_amd.nets.EncryptedCifar, the chain
the gate-parallel multi-GPU path shards): CIFAR binarynet_small, 348,160 bootstraps, one clear-margin
image -- decrypted logits against the plaintext model that is pinned to the reference's own
plaintext build (tests/golden/cifar_binarynet_small.json)."""
import numpy as np
import pytest

import plain_model as pm

pytestmark = pytest.mark.gpu


def test_binarynet_small_python_chain_matches_plaintext_model():
    import torch
    import redsec_amd
    from redsec_amd import client, nets
    sk = client.SecretKeySet("redsec_small_v2", seed=11)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    net = pm.CifarNet("binarynet_small")
    enc = nets.EncryptedCifar(be, net)
    labels, pix = pm.load_cifar_images()
    i = 1                                                        # plaintext margin 342 vs 36
    ct = torch.from_numpy(sk.encrypt_image(pix[i], seed=3)).cuda()
    out = enc.run(ct)
    assert out.shape == (10, be.W)
    logits = sk.decrypt_ints(out.cpu().numpy())
    plain = pm.cifar_forward(net, pix[i])
    assert int(np.argmax(logits)) == int(np.argmax(plain)) == int(labels[i])
    assert np.corrcoef(logits, plain)[0, 1] > 0.5                # weak-margin units flip (SURVEY.md hard part 7): 0.73 with this key
    assert be.rounding_certificate() < 0.2
    # unsharded call of the sharded entry point (no process group): identical ciphertexts
    assert torch.equal(enc.run(ct, shard=False), out)
    # (the OR-chain form of the max-pool is checked stage by stage, at full size, by the test below)
    be.close()


# ---- BASELINE configs[3] itself, stage by stage against the oracle ---------------------------------------------------
def _sample_rows(B, num_cus, rng, exact=32, wide=96):
    """Rows of a stage's slab to check: the first and the last workgroup's ciphertexts, eight across the middle
    (a workgroup boundary of the 8-ciphertext lock-step form), the cut-off last round of the throughput launcher (a
    tail of at most 4 x #CUs ciphertexts behind whole rounds of 8 x #CUs runs in another kernel form: four rows on
    either side of the cut) or eight seeded rows elsewhere -- `exact` rows for the oracle's exact product path -- and
    `wide` more rows spread over the whole slab for its rounding-based FFT path."""
    rows = set(range(min(8, B))) | set(range(max(0, B - 8), B)) | set(range(max(0, B // 2 - 4), min(B, B // 2 + 4)))
    cap, tail = 8 * num_cus, B % (8 * num_cus)
    if B > cap and 0 < tail <= 4 * num_cus:
        rows |= set(range(B - tail - 4, B - tail + 4))
    while len(rows) < min(exact, B):
        rows.add(int(rng.integers(0, B)))
    first = np.array(sorted(rows), np.int64)
    more = np.setdiff1d(np.unique(rng.integers(0, B, min(wide, B))), first)
    return first, more


def _oracle_stage(octx, rec, rows):
    import oracle_lib as ol
    ins = [x[rows].cpu().numpy() for x in rec["inputs"]]
    if rec["kind"] == "or":         # bootsOR with the output value mu: (0, 1/8) + a + b, bootstrapped (lib/BinOps_enc.cpp:164-167)
        return octx.bootstrap_batch(ol.gate_precombine("OR", ins[0], ins[1]), rec["mu"])
    return octx.bootstrap_batch(ins[0], rec["mu"])


@pytest.mark.parametrize("maxpool", ["fused", "chain"])
def test_binarynet_full_every_bootstrapped_stage_against_the_oracle(maxpool):
    """nets/cifar/binarynet/net.cpp:114-209 (the reference's widths 128-128-256-256-512-512, FC 1024-1024-10), one
    encrypted image, REDsec's shipped parameter set, at its REAL batch sizes (3,072 ... 131,072 ciphertexts per
    launch: the throughput form, its cut-off last round, the duo form of the 1,024-neuron layers; Quantize::execute
    lib/BinFunc.cpp:1056-1071, MaxPooling::execute :896-921 in both of this backend's forms). At every bootstrapped
    stage rows of the stage's ACTUAL input slab go through the CPU oracle -- exact product path on the boundary rows,
    its rounding-based FFT path on rows spread over the slab -- and must equal the stage's output word for word; the
    whole run is then repeated in the split-key mode (exact by an a-priori bound) and every stage's WHOLE output slab
    must equal the default mode's, so the sampled oracle rows vouch for both."""
    import torch
    import redsec_amd
    from redsec_amd import client, nets
    import oracle_lib as ol
    sk = client.SecretKeySet("redsec_small_v2", seed=13)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
    be.load_keys(sk.bk, sk.ksk)
    num_cus = be.info()["num_cus"]
    net = pm.CifarNet("binarynet")
    labels, pix = pm.load_cifar_images()
    i = 13                                                       # plaintext margin 292
    ct = torch.from_numpy(sk.encrypt_image(pix[i], seed=4)).cuda()
    enc = nets.EncryptedCifar(be, net, maxpool=maxpool)
    taps = []
    out = enc.run(ct, taps=taps)
    torch.cuda.synchronize()
    sizes = [int(r["out"].shape[0]) for r in taps]
    signs = [131072, 131072, 65536, 65536, 32768, 32768]
    if maxpool == "fused":
        assert sizes == [3072, signs[0], signs[1], 32768, signs[2], signs[3], 16384, signs[4], signs[5], 8192, 1024, 1024]
    else:
        assert sizes == [3072, signs[0], signs[1]] + [32768] * 3 + [signs[2], signs[3]] + [16384] * 3 + [signs[4], signs[5]] + [8192] * 3 + [1024, 1024]

    class _K:
        pass
    k = _K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
    octx = ol.Ctx(k)
    rng = np.random.default_rng(5)
    checked = 0
    for rec in taps:
        B = int(rec["out"].shape[0])
        first, more = _sample_rows(B, num_cus, rng)
        octx.set_fft(False)
        assert np.array_equal(rec["out"][first].cpu().numpy(), _oracle_stage(octx, rec, first)), (rec["name"], "exact path")
        octx.set_fft(True)
        assert np.array_equal(rec["out"][more].cpu().numpy(), _oracle_stage(octx, rec, more)), (rec["name"], "fft path")
        checked += len(first) + len(more)
    assert checked >= 100 * len(taps)
    assert be.rounding_certificate() < 0.2 and be.fft_fallbacks() == 0
    # decrypt level, EVERY row of every stage: wherever the stage's (pre-combined) input phase is at least 32 message steps
    # away from both decision boundaries, the output decrypts to sign(input) * mu within half of mu (SURVEY.md hard part 7:
    # closer inputs flip under the mod-switch rounding in any TFHE implementation of this parameter set)
    key = torch.from_numpy(np.ascontiguousarray(sk.lwe_key, dtype=np.int64)).cuda()

    def phase(ct):
        ph = ct[:, -1].long() - (ct[:, :-1].long() * key).sum(dim=1)
        return ((ph + (1 << 31)) % (1 << 32)) - (1 << 31)

    def wrap(v):
        return ((v + (1 << 31)) % (1 << 32)) - (1 << 31)
    n_strong = 0
    for rec in taps:
        ph_in = phase(rec["inputs"][0]) if rec["kind"] == "sign" else wrap(phase(rec["inputs"][0]) + phase(rec["inputs"][1]) + (1 << 29))
        strong = (ph_in.abs() >= (32 << 20)) & (ph_in.abs() <= (1 << 31) - (32 << 20))
        want = torch.where(ph_in >= 0, rec["mu"], -rec["mu"])
        err = (phase(rec["out"]) - want).abs()
        assert bool((err[strong] < rec["mu"] // 2).all()), rec["name"]      # (conv1 has 27 taps: no pre-activation reaches 32)
        n_strong += int(strong.sum())
        if rec["kind"] == "or":
            assert bool(strong.all()), rec["name"]            # gate inputs at +-1/8 are never near a boundary
    assert n_strong > 100000
    # the logits are the final layer of the bits the last bootstrapped stage produced (no bootstrap behind it)
    bits = torch.where(phase(taps[-1]["out"]) >= 0, 1, -1).cpu().numpy()
    sgn, zero, bias = net.fcs[-1]
    w = np.where(zero == 1, 0, np.where(sgn == 1, 1, -1)).astype(np.int64)
    logits = sk.decrypt_ints(out.cpu().numpy())
    assert np.abs(logits - (bits @ w + bias.astype(np.int64))).max() <= 8
    if maxpool == "fused":
        # ... and, for this key and image, the class of the plaintext model (pinned to the reference's plaintext build,
        # tests/golden/cifar_binarynet.json). Weak-margin units flip differently under another noise realisation -- the
        # OR-chain form re-randomises every pooled bit -- so the class is asserted for the default form only.
        plain = pm.cifar_forward(net, pix[i])
        assert int(np.argmax(logits)) == int(np.argmax(plain)) == int(labels[i])
    if maxpool == "chain":          # (the split-mode repeat runs on the default form only: the OR stages are the same kernels, and the suite has a time budget)
        be.close()
        return
    # the split-key mode: every stage's whole slab, word for word
    be.set_mode("split")
    taps_s = []
    out_s = enc.run(ct, taps=taps_s)
    assert [r["name"] for r in taps_s] == [r["name"] for r in taps]
    for a, b in zip(taps, taps_s):
        assert torch.equal(a["out"], b["out"]), a["name"]
    assert torch.equal(out, out_s)
    be.sync()                                                    # an enforced split certificate would surface here
    be.close()
