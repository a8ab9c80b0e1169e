"""Registry of the GPU contexts one test module keeps alive, and the per-test arithmetic-mode logic around them.

Test infrastructure. Round 5's GPU run went red because a test closed two contexts that the module-global list still held and
the next fixture step called into the destroyed handles. Here a closed context can never be touched again: `live()` drops every
entry whose handle is gone before anything is called on it, `scratch()` contexts are closed and dropped by the pool itself, and
tests/test_backend_pool_cpu.py drives exactly this logic with a stub backend on the CPU.
"""
import contextlib


class BackendPool:
    def __init__(self):
        self._entries = []

    def add(self, be):
        self._entries.append(be)
        return be

    def live(self):
        """The registered contexts whose handle still exists (closed ones are forgotten here)."""
        self._entries = [be for be in self._entries if not be.closed]
        return list(self._entries)

    def discard(self, be):
        self._entries = [e for e in self._entries if e is not be]

    @contextlib.contextmanager
    def scratch(self, make):
        """A context that lives for one `with` block: registered (so it follows the test's arithmetic mode when created before
        the mode is set), closed and forgotten on the way out -- also when the body raises."""
        be = self.add(make())
        try:
            yield be
        finally:
            self.discard(be)
            be.close()

    def close_all(self):
        for be in self.live():
            be.close()
        self._entries = []

    # ---- what the autouse fixture of a parity module does around every test ----
    def enter_mode(self, mode):
        for be in self.live():
            be.set_mode(mode)
            be.rounding_certificate(reset=True)

    def leave_mode(self, mode, limit=0.2):
        """After an FFT-mode test the rounding certificate of every context still alive must be far below 1/2."""
        if mode != "fft":
            return
        for be in self.live():
            cert = be.rounding_certificate(reset=True)
            assert cert < limit, cert
