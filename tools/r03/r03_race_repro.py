#!/usr/bin/env python3
"""Determinism of the general blind rotation at full size: the same batch bootstrapped `reps` times on a synthetic key, every
output compared with the first run's. A write-after-read race between wavefronts shows up as runs that differ.
  python tools/r03/r03_race_repro.py <set> <batch> <reps>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, redsec_amd
name = sys.argv[1] if len(sys.argv) > 1 else "redsec_medium"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
p = redsec_amd.params(name)
be = redsec_amd.Backend(p, device=0)
be.load_synthetic_keys(11)
be.set_mode("split")
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.integers(-2**31, 2**31, (B, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
first, bad, errs = None, 0, 0
for r in range(reps):
    try:
        out = be.bootstrap_wo_ks(x, 1 << 29).clone()
        torch.cuda.synchronize()
    except Exception as e:
        errs += 1
        print("rep", r, "error:", str(e)[:90], flush=True)
        try: be.certify(reset=True)
        except Exception: pass
        continue
    if first is None: first = out
    elif not torch.equal(out, first):
        bad += 1
        rows = (out != first).any(dim=1).nonzero().flatten().tolist()
        print("rep", r, "differs from rep 0 in", len(rows), "ciphertexts, first rows", rows[:5], flush=True)
print(name, "batch", B, "reps", reps, "-> runs differing from the first:", bad, "runs refused by the enforced certificate:", errs, flush=True)
