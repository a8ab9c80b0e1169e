#!/usr/bin/env python3
"""Throughput of the general ring path (RS_MODE_FFT_SPLIT, csrc/rs_general.hip) per parameter set, on random keys
(timing does not depend on key values). The sets with large rings are timed at a reduced LWE dimension n' and the rate
at the full n is derived from the time per CMUX step (a blind rotation is n sequential steps of identical cost).

  python tools/general_rate.py [set ...]      sets: default128 redsec_small_v2 redsec_small redsec_medium redsec_large
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import redsec_amd

N_TIMED = {"default128": 630, "redsec_small_v2": 350, "redsec_small": 500, "redsec_medium": 256, "redsec_large": 128}


def main():
    sets = sys.argv[1:] or list(N_TIMED)
    rng = np.random.default_rng(0)
    for name in sets:
        full = redsec_amd.params(name)
        p = redsec_amd.params(name, n=N_TIMED[name])
        be = redsec_amd.Backend(p, device=0)
        bk = rng.integers(-2**31, 2**31, p.n * 2 * p.bk_l * 2 * p.N, dtype=np.int64).astype(np.int32)
        ksk = rng.integers(-2**31, 2**31, p.N * p.ks_t * (1 << p.ks_basebit) * (p.n + 1), dtype=np.int64).astype(np.int32)
        be.load_keys(bk, ksk)
        be.set_mode("split")
        be.set_timing(True)
        cus = be.info()["num_cus"]
        B = {1024: 16, 2048: 8, 4096: 4, 8192: 2}[p.N] * cus
        x = torch.from_numpy(rng.integers(-2**31, 2**31, (B, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
        out = be.empty(B, p.n + 1)
        be.bootstrap(x, 1 << 29, out=out)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            be.bootstrap(x, 1 << 29, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        br, ks = be.last_kernel_ms()
        steps_per_s = B * p.n / (br * 1e-3)
        full_rate = 1.0 / (full.n / steps_per_s + (ks * 1e-3 / B) * (full.n + 1) / (p.n + 1))
        print(json.dumps({"set": name, "N": p.N, "l": p.bk_l, "Bgbit": p.bk_Bgbit, "n_timed": p.n, "n_full": full.n, "batch": B,
                          "launch": be.last_launch(), "blind_rotate_ms": round(br, 3), "keyswitch_ms": round(ks, 3), "wall_ms": round(dt * 1e3, 3),
                          "cmux_steps_per_s": round(steps_per_s), "bootstraps_per_s_at_full_n": round(full_rate, 1),
                          "split_bound": be.split_bound(), "max_rounding_distance": be.rounding_certificate()}), flush=True)
        be.close()
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
