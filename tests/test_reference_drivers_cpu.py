"""The reference's own client tools, compiled UNMODIFIED against the TFHE-compatible shim
(redsec_amd/host/tfhe), work without a GPU: keygen with the shipped parameter set, image encryption,
file formats. (The encrypted network driver itself needs the GPU: tests/test_gpu_reference_drivers.py.)"""
import os

import numpy as np
import pytest

import plain_model as pm
import refdrivers as rd

REF_PRESENT = os.path.isdir("/root/reference/nets")


@pytest.fixture(scope="module")
def drivers():
    from redsec_amd import build
    if REF_PRESENT:
        build.build_reference_drivers()
    if not rd.available():
        pytest.skip("build/refnets not built (needs /root/reference at build time)")
    return True


def test_unmodified_client_tools_roundtrip(drivers, tmp_path):
    client, _ = rd.make_tree(str(tmp_path))
    r = rd.run("client_gen_secure_keyset.out", client)
    assert r.returncode == 0 and "Keyset generated!" in r.stdout, r.stderr
    params, lwe_key = rd.read_secret_key(os.path.join(client, "secret.key"))
    # redsec_params_small_v2, client/gen_secure_keyset.cpp:70-97
    assert params == dict(n=350, N=1024, k=1, l=10, Bgbit=3, t=9, basebit=3)
    assert set(np.unique(lwe_key)) <= {0, 1}
    labels, pixels = pm.load_images()
    rd.write_image_csv(os.path.join(client, "img.csv"), labels[0], pixels[0])
    r = rd.run("client_encrypt_image.out", client, "img.csv")
    assert r.returncode == 0, r.stderr
    ct = rd.read_ciphertexts(os.path.join(client, "image.ctxt"), 350, 784)
    phase = (ct[:, 350].astype(np.int64) - (ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
    dec = ((phase + (1 << 19)) >> 20) & 0xFFF
    dec = np.where(dec >= 2048, dec - 4096, dec)
    assert np.array_equal(dec, 2 * pixels[0] - 255)            # encrypt_image.cpp:76-77


def test_layers_library_exports_reference_api():
    """Same mangled names as the reference's lib/{Bin,Int}Layer.cpp, lib/Layer.cpp, lib/*Ops_enc.cpp."""
    import subprocess
    from redsec_amd import build
    lib = build.build_layers()
    syms = subprocess.run(["nm", "-DC", "--defined-only", lib], capture_output=True, text=True).stdout
    for want in ("BinLayer::BinLayer(_CONVTYPE, unsigned short, _POOLTYPE, _QUANT_TYPE, _NET_PARAMS*, TFheGateBootstrappingCloudKeySet*)",
                 "BinLayer::prep(_IO_FILE*, _DIMS*)", "BinLayer::execute(LweSample*)",
                 "IntLayer::IntLayer(_CONVTYPE, unsigned short, _POOLTYPE, _QUANT_TYPE, _NET_PARAMS*, TFheGateBootstrappingCloudKeySet*)",
                 "IntLayer::execute(tMultiBits*)", "BinOps::binarize_int(", "BinOps::max(", "BinOps::add(", "IntOps::relu(",
                 "BinOps::get_ternfilters(", "mbit_calloc(", "bit_free(", "tfhe_bootstrap_FFT(", "bootsMUX(", "lweAddMulTo(",
                 "BinFunc::Convolution::execute(LweSample*)", "BinFunc::MaxPooling::execute(LweSample*)", "BinFunc::SumPooling::execute(tMultiBits*)",
                 "BinFunc::Quantize::relu_shift(tMultiBits*, tMultiBits*, unsigned int*)", "IntFunc::Convolution::execute(tMultiBits*)",
                 "IntFunc::Quantize::add_bias(tMultiBits*, tMultiBits*)", "IntFunc::SumPooling::prep(_DIMS*, TFheGateBootstrappingCloudKeySet*)"):
        assert want in syms, want
