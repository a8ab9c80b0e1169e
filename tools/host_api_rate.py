"""PCIe-inclusive rate of the synchronous host-pointer API (rs_gate with host buffers), 65,536 default-128 NANDs."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, redsec_amd
from redsec_amd import client
sk = client.SecretKeySet("default128", seed=1)
be = redsec_amd.Backend(redsec_amd.params("default128"), 0); be.load_keys(sk.bk, sk.ksk)
G = 65536
rng = np.random.default_rng(0)
a = sk.encrypt_bits(rng.integers(0, 2, G), seed=1); b = sk.encrypt_bits(rng.integers(0, 2, G), seed=2)
be.gate_host("NAND", a[:64], b[:64])
for _ in range(2):
    t = time.perf_counter(); out = be.gate_host("NAND", a, b); dt = time.perf_counter() - t
    print("host-pointer rs_gate, 65536 NANDs: %.1f ms -> %.1f k/s" % (dt * 1e3, G / dt / 1e3))
