#!/usr/bin/env python3
"""Child process of tests/test_gpu_first_launch.py: ONE fresh process = the first launch of every kernel form.

A race in a hand-rolled barrier / waitcnt protocol (key rings, partner exchanges through the key buffer, barrier-free
wave-local exchanges) needs wavefronts that run far apart, which is what the first launch of a kernel in a process gives
(round 3: the write-after-read race of the general kernels showed in about one fresh process in six and never in a warm
one). So: (1) a differential stress of the three arithmetic modes of both shipped N = 1024 gadgets at every batch-size
boundary between kernel forms -- every output word of the FFT and split-key modes against the exact-NTT mode, on the
device; (2) one 1,024-ciphertext redsec_params_medium batch at its full n = 3072 on a synthetic key, run twice: equal
runs, and no RS_ERR_INEXACT from the enforced rounding certificate. Prints one JSON line; exit code 1 on any finding.

  python tests/first_launch_stress.py <seed>
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import redsec_amd
from redsec_amd import client

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
GATES = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ORYN"]
findings = []
launched = set()
t0 = time.time()
for name, n in (("default128", 24), ("redsec_small_v2", 20)):
    sk = client.SecretKeySet(name, seed=seed, n=n)
    be = redsec_amd.Backend(redsec_amd.params(name, n=n), 0)
    be.load_keys(sk.bk, sk.ksk)
    cus = be.info()["num_cus"]
    # one batch size inside every size class of the launchers, and the boundaries between them
    edges = [1, cus, cus + 1, 2 * cus, 2 * cus + 1, 4 * cus, 4 * cus + 1, 8 * cus, 8 * cus + 1, 12 * cus + 5, 16 * cus + 3]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for r, B in enumerate(edges):
        a = torch.from_numpy(rng.integers(-2**31, 2**31, (B, n + 1), dtype=np.int32)).cuda()
        b = torch.from_numpy(rng.integers(-2**31, 2**31, (B, n + 1), dtype=np.int32)).cuda()
        a[int(rng.integers(0, B)), : int(rng.integers(0, n))] = 0                 # identity steps
        op = GATES[(r + seed) % len(GATES)]
        lut = torch.from_numpy(rng.integers(-2**31, 2**31, (3, 1024), dtype=np.int32)).cuda()
        torch.cuda.synchronize()
        outs = {}
        for mode in ("exact", "fft", "split"):
            be.set_mode(mode)
            with torch.cuda.stream(streams[r & 1]):
                g = be.gate(op, a, b)
                launched.add("%s/%s" % (mode, be.last_launch()["form"]))
                l = be.bootstrap_lut(a, lut)
                m = be.mux(a, b, g) if B <= 4 * cus else None
            torch.cuda.synchronize()
            outs[mode] = (g, l, m)
        for mode in ("fft", "split"):
            for k in range(3):
                x, y = outs[mode][k], outs["exact"][k]
                if x is not None and not torch.equal(x, y):
                    findings.append("%s B=%d %s %s %s: %d rows differ from the exact mode" %
                                    (name, B, op, mode, ("gate", "lut", "mux")[k], int((x != y).any(dim=1).sum())))
    if be.fft_fallbacks():
        findings.append("%s: %d FFT-mode calls were recomputed exactly" % (name, be.fft_fallbacks()))
    be.close()
# (2) the general kernels at full size: barrier-free wave-local exchanges between passes
p = redsec_amd.params("redsec_medium")
be = redsec_amd.Backend(p, device=0)
be.load_synthetic_keys(11)
x = torch.from_numpy(rng.integers(-2**31, 2**31, (1024, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
first = None
for rep in range(2):
    try:
        out = be.bootstrap_wo_ks(x, 1 << 29).clone()
        be.sync()
    except Exception as e:                                   # RS_ERR_INEXACT: the enforced certificate refused the run
        findings.append("redsec_medium rep %d: %s" % (rep, str(e)[:120]))
        break
    if first is None:
        first = out
    elif not torch.equal(out, first):
        findings.append("redsec_medium: run %d differs from run 0 in %d ciphertexts" % (rep, int((out != first).any(dim=1).sum())))
try:
    be.close()
except Exception as e:
    findings.append("redsec_medium close: %s" % str(e)[:120])
print(json.dumps({"seed": seed, "findings": findings, "forms_launched": sorted(launched), "seconds": round(time.time() - t0, 1)}), flush=True)
sys.exit(1 if findings else 0)
