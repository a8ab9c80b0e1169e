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
    # the OR-chain form of the max-pool computes the same bits: logits decrypt to the same class and stay
    # close (the two forms differ only in noise, which moves weak-margin units of the following layers)
    chain = sk.decrypt_ints(nets.EncryptedCifar(be, net, maxpool="chain").run(ct).cpu().numpy())
    assert int(np.argmax(chain)) == int(labels[i])
    assert np.corrcoef(chain, logits)[0, 1] > 0.5
