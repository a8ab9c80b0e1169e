"""The fixture logic of the GPU parity modules (tests/backend_pool.py), driven on the CPU with a stub context that behaves like
the C ABI on a destroyed handle: every call after close() fails with "null context". Round 5's GPU run went red on exactly this
(a test body closed contexts the module list still held); that class of bug now fails here, in `-m "not gpu"`."""
import ast
import glob
import os

import pytest

from backend_pool import BackendPool

HERE = os.path.dirname(os.path.abspath(__file__))


class StubBackend:
    def __init__(self, cert=0.01):
        self.h = object()
        self.mode_ = "fft"
        self.cert = cert
        self.calls = []

    @property
    def closed(self):
        return self.h is None

    def _live(self):
        if self.h is None:
            raise RuntimeError("redsec_hip error 1: null context")

    def set_mode(self, mode):
        self._live(); self.mode_ = mode; self.calls.append(("set_mode", mode))

    def rounding_certificate(self, reset=True):
        self._live(); self.calls.append(("cert", reset))
        return self.cert

    def close(self):
        self.h = None


def test_a_context_closed_inside_a_test_body_is_never_called_again():
    pool = BackendPool()
    keep = pool.add(StubBackend())
    pool.enter_mode("fft")
    a, b = pool.add(StubBackend()), pool.add(StubBackend())      # what round 5's test did: create two, close both, tell nobody
    a.close(); b.close()
    pool.leave_mode("fft")                                       # teardown of that test
    pool.enter_mode("exact")                                     # setup of the next one
    pool.leave_mode("exact")
    assert pool.live() == [keep]
    assert keep.mode_ == "exact"
    assert a.calls == [] and b.calls == []


def test_scratch_contexts_are_closed_and_forgotten_also_on_failure():
    pool = BackendPool()
    with pool.scratch(StubBackend) as be:
        assert pool.live() == [be]
    assert be.closed and pool.live() == []
    with pytest.raises(ZeroDivisionError):
        with pool.scratch(StubBackend) as be2:
            1 / 0
    assert be2.closed and pool.live() == []
    pool.enter_mode("fft"); pool.leave_mode("fft")


def test_leave_mode_checks_the_certificate_only_in_fft_mode_and_only_on_live_contexts():
    pool = BackendPool()
    bad = pool.add(StubBackend(cert=0.4))
    pool.leave_mode("exact")
    with pytest.raises(AssertionError):
        pool.leave_mode("fft")
    bad.close()
    pool.leave_mode("fft")
    pool.close_all()
    assert pool.live() == []


def test_real_backend_reports_closed_without_a_library_call():
    """Backend.closed must not need the shared library (it is what protects a destroyed handle from being used)."""
    import redsec_amd.backend as rb
    be = rb.Backend.__new__(rb.Backend)
    assert be.closed
    be.h = object()
    assert not be.closed
    be.h = None
    assert be.closed
    be.close()            # a no-op on a closed context


def test_gpu_test_modules_keep_no_bare_module_level_context_lists():
    """No GPU test module may hold contexts in a plain module-level list again: module-lifetime contexts go through a BackendPool."""
    for path in glob.glob(os.path.join(HERE, "test_gpu_*.py")):
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.Assign) and isinstance(node.value, ast.List) and not node.value.elts:
                names = [t.id for t in node.targets if isinstance(t, ast.Name)]
                assert not any("BACKEND" in n.upper() for n in names), (path, names)
