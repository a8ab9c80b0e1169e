#!/usr/bin/env python3
"""Differential stress of the three arithmetic modes: random batch sizes across every kernel-form boundary, random gates
and test polynomials, two streams; every output word of the FFT and split-key modes compared with the exact-NTT mode on the
device. A race in a hand-rolled barrier / vmcnt protocol (key rings, mask windows) would show up as a differing word.

  python tools/stress_modes.py [rounds] [seed]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import redsec_amd
from redsec_amd import client

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
GATES = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ORYN"]
bad = 0
t0 = time.time()
for name, n in (("default128", int(os.environ.get("STRESS_N", 48))), ("redsec_small_v2", int(os.environ.get("STRESS_N", 40)))):
    sk = client.SecretKeySet(name, seed=seed, n=n)
    be = redsec_amd.Backend(redsec_amd.params(name, n=n), 0)
    be.load_keys(sk.bk, sk.ksk)
    cus = be.info()["num_cus"]
    edges = [1, 2, cus - 1, cus, cus + 1, 2 * cus, 2 * cus + 1, 4 * cus, 4 * cus + 1, 8 * cus - 1, 8 * cus, 8 * cus + 1, 12 * cus + 5, 16 * cus + 3, 24 * cus]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for r in range(rounds):
        B = int(edges[r % len(edges)] if r < 2 * len(edges) else rng.integers(1, 30 * cus))
        a = torch.from_numpy(rng.integers(-2**31, 2**31, (B, n + 1), dtype=np.int32)).cuda()
        b = torch.from_numpy(rng.integers(-2**31, 2**31, (B, n + 1), dtype=np.int32)).cuda()
        a[rng.integers(0, B), : rng.integers(0, n)] = 0                 # identity steps
        op = GATES[r % len(GATES)]
        lut = torch.from_numpy(rng.integers(-2**31, 2**31, (3, 1024), dtype=np.int32)).cuda()
        torch.cuda.synchronize()
        outs = {}
        for mode in ("exact", "fft", "split"):
            be.set_mode(mode)
            with torch.cuda.stream(streams[r & 1]):
                g = be.gate(op, a, b)
                l = be.bootstrap_lut(a, lut)
                m = be.mux(a, b, g) if B <= 4 * cus else None
            torch.cuda.synchronize()
            outs[mode] = (g, l, m)
        for mode in ("fft", "split"):
            for k in range(3):
                x, y = outs[mode][k], outs["exact"][k]
                if x is not None and not torch.equal(x, y):
                    bad += 1
                    print("MISMATCH", name, "round", r, "B", B, op, mode, ("gate", "lut", "mux")[k], int((x != y).any(dim=1).sum()), "rows", flush=True)
    print(name, "done:", rounds, "rounds, mismatches", bad, "(%.0f s)" % (time.time() - t0), "fft fallbacks", be.fft_fallbacks(), flush=True)
    be.close()
sys.exit(1 if bad else 0)
