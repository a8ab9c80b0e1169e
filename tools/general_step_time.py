#!/usr/bin/env python3
"""Per-CMUX-step time of the general ring kernels as a function of the key's size: the same parameter set at small LWE dimensions n
(key = n x 2l rows: a few MB stay in the L2s, tens of MB in the Infinity Cache, GB come from HBM at every step). Separates the
first-touch HBM latency of the key stream from everything else a step costs. usage: python tools/general_step_time.py [set ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import redsec_amd

for name in (sys.argv[1:] or ["redsec_medium", "redsec_large"]):
    full = redsec_amd.params(name).n
    for n in (4, 32, 256, full):
        p = redsec_amd.params(name, n=n)
        be = redsec_amd.Backend(p, device=0)
        be.load_synthetic_keys(2026)
        be.set_mode("split"); be.set_timing(True)
        cus = be.info()["num_cus"]
        B = 2 * {1024: 8, 2048: 4, 4096: 2, 8192: 1}[p.N] * cus
        x = torch.from_numpy(np.random.default_rng(0).integers(-2**31, 2**31, (B, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
        out = be.empty(B, p.n + 1)
        reps = 20 if n <= 32 else (5 if n <= 256 else 2)
        br = []
        for _ in range(reps + 1):
            be.bootstrap(x, 1 << 29, out=out); torch.cuda.synchronize(); br.append(be.last_kernel_ms()[0])
        t = float(np.median(br[1:]))
        print(json.dumps({"set": name, "n": n, "key_MB": round(be.info()["bk_device_bytes"] / 1e6, 1), "batch": B, "blind_rotate_ms": round(t, 3),
                          "us_per_cmux_step_and_round": round(1e3 * t / (n * 2), 3)}), flush=True)
        be.close(); del x, out; torch.cuda.empty_cache()
