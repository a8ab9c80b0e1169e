#!/usr/bin/env python3
"""Throughput of the general ring path (RS_MODE_FFT_SPLIT, csrc/rs_general.hip) per parameter set, MEASURED at each set's full
LWE dimension n on a synthetic key generated on the device (rs_load_synthetic_keys: timing does not depend on key values, and
redsec_params_large's 2.4 GB + 7.2 GB of key never exist on the host).

  python tools/general_rate.py [set ...]      sets: default128 redsec_small_v2 redsec_small redsec_medium redsec_large
  REDSEC_GENERAL_BATCH_X=<k>: batch = k x the resident workgroups (default 2)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import redsec_amd

SETS = ["default128", "redsec_small_v2", "redsec_small", "redsec_medium", "redsec_large"]


def main():
    sets = sys.argv[1:] or SETS
    rng = np.random.default_rng(0)
    kx = int(os.environ.get("REDSEC_GENERAL_BATCH_X", "2"))
    for name in sets:
        p = redsec_amd.params(name)
        be = redsec_amd.Backend(p, device=0)
        t0 = time.perf_counter()
        be.load_synthetic_keys(2026)
        load_s = time.perf_counter() - t0
        be.set_mode("split")
        be.set_timing(True)
        cus = be.info()["num_cus"]
        B = kx * {1024: 8, 2048: 4, 4096: 2, 8192: 1}[p.N] * cus
        x = torch.from_numpy(rng.integers(-2**31, 2**31, (B, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
        out = be.empty(B, p.n + 1)
        be.bootstrap(x, 1 << 29, out=out)
        torch.cuda.synchronize()
        reps = 2 if p.N >= 4096 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            be.bootstrap(x, 1 << 29, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        br, ks = be.last_kernel_ms()
        try:
            cert = be.rounding_certificate()
        except Exception as e:          # timing-only variant builds compute wrong values on purpose
            cert = "refused: %s" % (str(e)[:60],)
        print(json.dumps({"set": name, "N": p.N, "l": p.bk_l, "Bgbit": p.bk_Bgbit, "n": p.n, "batch": B, "key": "synthetic, generated on the device",
                          "key_load_s": round(load_s, 2), "key_device_GB": round((be.info()["bk_device_bytes"] + be.info()["ksk_device_bytes"]) / 1e9, 2),
                          "launch": be.last_launch(), "blind_rotate_ms": round(br, 3), "keyswitch_ms": round(ks, 3), "wall_ms": round(dt * 1e3, 3),
                          "cmux_steps_per_s": round(B * p.n / (br * 1e-3)), "bootstraps_per_s": round(B / dt, 1),
                          "split_bound": be.split_bound(), "max_rounding_distance": cert}), flush=True)
        be.close()
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
