#!/usr/bin/env python3
"""repeat the full-size medium random-key parity check in one process and report mismatches (diagnosing a one-off failure)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib as ol, redsec_amd
name, n = "redsec_medium", 3072
p = ol.params(name); p.n = n
rng = np.random.default_rng(2024)
class K: pass
ks = K(); ks.p = p
ks.bk = rng.integers(-2**31, 2**31, p.n * 2 * p.bk_l * 2 * p.N, dtype=np.int32)
ks.ksk = rng.integers(-2**31, 2**31, p.N * p.ks_t * (1 << p.ks_basebit) * (p.n + 1), dtype=np.int32)
ctx = ol.Ctx(ks)
be = redsec_amd.Backend(redsec_amd.params(name, n=n), device=0)
be.load_keys(ks.bk, ks.ksk)
ct = rng.integers(-2**31, 2**31, (3, p.n + 1), dtype=np.int32); ct[2, 5:9] = 0
mu = ol.to_torus(1, 4096)
want = ctx.bootstrap_batch(ct, mu)
d = torch.from_numpy(ct).cuda()
bad_gpu = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    got = be.bootstrap(d, mu).cpu().numpy()
    if not np.array_equal(got, want):
        bad_gpu += 1
        w = np.argwhere(got != want)
        print("rep", rep, "GPU differs:", len(w), "words, first", w[:3].tolist(), "cert", be.rounding_certificate(), flush=True)
want2 = ctx.bootstrap_batch(ct, mu)
print("gpu mismatching runs:", bad_gpu, "oracle repeat equal:", np.array_equal(want, want2), flush=True)
