"""TFHE v1.1 key-file format (client/gen_secure_keyset.cpp:108-114, nets/mnist/sign1024x1/net.cpp:53-55):
the shim's C++ reader/writer and redsec_amd/client.py's Python reader/writer are two restatements of the same layout
and must read each other's files; files in the backend's first (private) format still load. The layout itself is
[TFHE-recalled] -- libtfhe is not in this image; tools/tfhe_crosscheck.md is the recipe to pin it with the real library."""
import io
import os

import numpy as np
import pytest

import cppbuild
from redsec_amd import client


@pytest.fixture(scope="module")
def exe():
    e = cppbuild.build("keyio_roundtrip")
    if e is None:
        pytest.skip("no host compiler")
    return e


def _write(sk, path, secret):
    (_, _, _, _, _, _, _, ks_stdev, bk_stdev) = client.PARAM_SETS[sk.name]
    with open(path, "wb") as f:
        client.write_tfhe_keyset(f, sk, secret, ks_stdev, bk_stdev)


def test_python_writer_cpp_reader_cpp_writer_python_reader(exe, tmp_path):
    sk = client.SecretKeySet("redsec_small_v2", seed=5, n=12)       # reduced n: the files stay small
    s_in, c_in, s_out, c_out = (str(tmp_path / n) for n in ("s.in", "c.in", "s.out", "c.out"))
    _write(sk, s_in, True)
    _write(sk, c_in, False)
    r = cppbuild.run(exe, s_in, c_in, s_out, c_out, env={"REDSEC_TFHE_STRICT": "1"})
    assert r.returncode == 0, r.stderr
    assert "n=12 N=1024 l=10 Bgbit=3 t=9 basebit=3" in r.stdout
    # the C++ writer reproduces the Python writer's bytes except for the formatting of the doubles in the text sections
    back = client.read_tfhe_keyset(open(s_out, "rb"), secret=True)
    assert np.array_equal(back["bk"], sk.bk) and np.array_equal(back["ksk"], sk.ksk)
    assert np.array_equal(back["lwe_key"], sk.lwe_key) and np.array_equal(back["tlwe_key"], sk.tlwe_key)
    assert (back["n"], back["l"], back["Bgbit"], back["t"], back["basebit"]) == (12, 10, 3, 9, 3)
    assert back["ks_stdev"] == 2.0 ** -25 and back["bk_stdev"] == 2.0 ** -30
    cloud = client.read_tfhe_keyset(open(c_out, "rb"), secret=False)
    assert np.array_equal(cloud["bk"], sk.bk) and np.array_equal(cloud["ksk"], sk.ksk) and "lwe_key" not in cloud
    assert back["uids"][:2] == [client.TFHE_UID["bk_key"], client.TFHE_UID["ks_key"]]
    assert back["uids"][-2:] == [client.TFHE_UID["lwe_key"], client.TFHE_UID["tlwe_key"]]


def test_private_format_still_loads_and_converts(exe, tmp_path):
    sk = client.SecretKeySet("default128", seed=6, n=10)
    s_in, c_in = str(tmp_path / "s.in"), str(tmp_path / "c.in")
    _write(sk, s_in, True)
    _write(sk, c_in, False)
    s_rs, c_rs, s_tf, c_tf = (str(tmp_path / n) for n in ("s.rs", "c.rs", "s.tf", "c.tf"))
    assert cppbuild.run(exe, s_in, c_in, s_rs, c_rs, env={"REDSEC_KEY_FORMAT": "rs"}).returncode == 0      # TFHE -> private
    assert open(s_rs, "rb").read(4) == b"RSS1" and open(c_rs, "rb").read(4) == b"RSK1"
    assert cppbuild.run(exe, s_rs, c_rs, s_tf, c_tf).returncode == 0                                        # private -> TFHE
    back = client.read_tfhe_keyset(open(s_tf, "rb"), secret=True)
    assert np.array_equal(back["bk"], sk.bk) and np.array_equal(back["ksk"], sk.ksk) and np.array_equal(back["lwe_key"], sk.lwe_key)


def test_foreign_uids_are_tolerated_unless_strict(exe, tmp_path, monkeypatch):
    """The uid constants are the least certain part of the restatement: sizes come from the text sections, so a file
    whose uids differ still loads (with a note), and REDSEC_TFHE_STRICT turns that into a failure."""
    sk = client.SecretKeySet("redsec_small_v2", seed=7, n=8)
    monkeypatch.setitem(client.TFHE_UID, "tgsw_sample", 1234)
    s_in, c_in = str(tmp_path / "s.in"), str(tmp_path / "c.in")
    _write(sk, s_in, True)
    _write(sk, c_in, False)
    outs = [str(tmp_path / n) for n in ("s.out", "c.out")]
    r = cppbuild.run(exe, s_in, c_in, *outs)
    assert r.returncode == 0 and "type uid 1234" in r.stderr
    assert cppbuild.run(exe, s_in, c_in, *outs, env={"REDSEC_TFHE_STRICT": "1"}).returncode != 0


def test_ciphertext_records_roundtrip():
    sk = client.SecretKeySet("redsec_small_v2", seed=8, n=16)
    ct = sk.encrypt_image(np.arange(10), seed=2)
    buf = io.BytesIO()
    client.write_ciphertexts(buf, ct)
    assert len(buf.getvalue()) == 10 * (4 * 16 + 16)              # 4n + 16 bytes per sample
    buf.seek(0)
    assert np.array_equal(client.read_ciphertexts(buf, 16, 10), ct)
