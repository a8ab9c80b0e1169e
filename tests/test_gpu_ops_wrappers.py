"""SURVEY.md section 8 rows a2-a9 through the C++ wrappers themselves: tests/cpp/ops_driver.cpp calls BinOps::add /
add_bit / inc / multiply / max / relu / shift / binarize_int / unbinarize_int / add_int* / multiply_pc_ints /
add_pc_ints and IntOps::invert / add / add_inplace / subtract / relu plus bootsMUX as a REDsec translation unit would
(lib/BinOps_enc.h:8-49, lib/IntOps_enc.h:9-32), on the shipped parameter set, and checks every result by decryption."""
import pytest

import cppbuild

pytestmark = pytest.mark.gpu


def test_per_ciphertext_wrappers_on_the_gpu():
    exe = cppbuild.build("ops_driver")
    if exe is None:
        pytest.skip("no host compiler and no prebuilt test program")
    r = cppbuild.run(exe)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout + r.stderr
    for what in ("BinOps::add 14+9", "BinOps::inc", "IntOps::invert b=0", "BinOps::relu top bit 1", "BinOps::unbinarize_int", "bootsMUX sel=0",
                 "BinOps::multiply_pc_ints", "IntOps::subtract"):
        assert "PASS " + what in r.stdout, what


def test_per_stage_classes_equal_the_layers():
    """BinFunc::* / IntFunc::* (SURVEY.md 8b "C++ surface to keep", lib/BinFunc.h:37-173, lib/IntFunc.h:27-140): tests/cpp/func_driver.cpp."""
    exe = cppbuild.build("func_driver")
    if exe is None:
        pytest.skip("no host compiler and no prebuilt test program")
    r = cppbuild.run(exe)
    assert r.returncode == 0 and "failures: 0" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("PASS") == 12
