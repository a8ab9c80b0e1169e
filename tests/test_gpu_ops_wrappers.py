"""SURVEY.md section 8 rows a2-a9 through the C++ wrappers themselves: tests/cpp/ops_driver.cpp calls BinOps::add /
add_bit / inc / multiply / max / relu / shift / binarize_int / unbinarize_int / add_int* / multiply_pc_ints /
add_pc_ints and IntOps::invert / add / add_inplace / subtract / relu plus bootsMUX as a REDsec translation unit would
(lib/BinOps_enc.h:8-49, lib/IntOps_enc.h:9-32), on the shipped parameter set, and checks every result by decryption."""
import pytest

import cppbuild

pytestmark = pytest.mark.gpu


def test_per_ciphertext_wrappers_on_the_gpu(tmp_path):
    """... and, for the bootstrapped and the copying wrappers (rows a2 binarize_int, a3 unbinarize_int, a4 max, a5 relu in both
    namespaces, a6 bootsMUX, a8 multiply / invert), WORD FOR WORD: the driver leaves its key and every such call's input and
    output ciphertexts in files; the outputs are recomputed here with the CPU oracle (exact product path) from the same
    inputs and key -- tfhe_bootstrap_FFT with the wrapper's mu, bootsAND / bootsOR / bootsMUX, bootsNOT / bootsCOPY."""
    import numpy as np
    import oracle_lib as ol
    from redsec_amd import client
    exe = cppbuild.build("ops_driver")
    if exe is None:
        pytest.skip("no host compiler and no prebuilt test program")
    r = cppbuild.run(exe, str(tmp_path))
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout + r.stderr
    for what in ("BinOps::add 14+9", "BinOps::inc", "IntOps::invert b=0", "BinOps::relu top bit 1", "BinOps::unbinarize_int", "bootsMUX sel=0",
                 "BinOps::multiply_pc_ints", "IntOps::subtract"):
        assert "PASS " + what in r.stdout, what
    keys = client.read_tfhe_keyset(open(tmp_path / "secret.key", "rb"), secret=True)
    assert (keys["n"], keys["N"], keys["l"], keys["Bgbit"], keys["t"], keys["basebit"]) == (350, 1024, 10, 3, 9, 3)

    class _K:
        pass
    k = _K(); k.p = ol.params("redsec_small_v2")
    k.bk = np.ascontiguousarray(keys["bk"]).ravel(); k.ksk = np.ascontiguousarray(keys["ksk"]).ravel()
    octx = ol.Ctx(k)
    records = [ln.split() for ln in open(tmp_path / "vectors.txt").read().splitlines()]
    total = sum(int(rec[2]) + 1 for rec in records)
    ct = client.read_ciphertexts(open(tmp_path / "vectors.ctxt", "rb"), 350, total)
    pos, seen = 0, set()
    for name, kind, n_in, mu in records:
        n_in, mu = int(n_in), int(mu)
        ins, out = [ct[pos + i:pos + i + 1] for i in range(n_in)], ct[pos + n_in]
        pos += n_in + 1
        if kind == "bootstrap":
            want = octx.bootstrap_batch(ins[0], mu)[0]
        elif kind in ("AND", "OR"):
            want = octx.gate_batch(kind, ins[0], ins[1])[0]
        elif kind == "MUX":
            want = octx.mux_batch(ins[0], ins[1], ins[2])[0]
        elif kind == "not":
            want = (-ins[0][0].astype(np.int64)).astype(np.uint64).astype(np.uint32).view(np.int32)     # bootsNOT negates every word
        else:
            assert kind == "copy"
            want = ins[0][0]
        assert np.array_equal(out, want), name
        seen.add(name)
    assert {"BinOps::binarize_int", "BinOps::unbinarize_int", "BinOps::max", "BinOps::relu", "IntOps::relu", "bootsMUX_sel0",
            "BinOps::multiply_by_0", "IntOps::invert_b0", "IntOps::invert_b1"} <= seen and len(records) >= 25


def test_per_stage_classes_equal_the_layers():
    """BinFunc::* / IntFunc::* (SURVEY.md 8b "C++ surface to keep", lib/BinFunc.h:37-173, lib/IntFunc.h:27-140): tests/cpp/func_driver.cpp."""
    exe = cppbuild.build("func_driver")
    if exe is None:
        pytest.skip("no host compiler and no prebuilt test program")
    r = cppbuild.run(exe)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("PASS") == 16
