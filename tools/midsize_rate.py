#!/usr/bin/env python3
"""Blind-rotation + keyswitch time against the batch size, across the kernel-form boundaries (cooperative / duo or half-size
lock-step groups / per-wave / full lock-step workgroups): the curve should have no step where a larger-batch form would
already be faster at a smaller batch.

  python tools/midsize_rate.py [default128|redsec_small_v2] [B ...]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, redsec_amd
from redsec_amd import client

name = sys.argv[1] if len(sys.argv) > 1 else "default128"
sizes = [int(v) for v in sys.argv[2:]] or [64, 128, 196, 256, 257, 384, 512, 513, 600, 800, 1024, 1025, 1536, 2047, 2048, 3072, 4096]
sk = client.SecretKeySet(name, seed=3)
be = redsec_amd.Backend(redsec_amd.params(name), 0); be.load_keys(sk.bk, sk.ksk); be.set_timing(True)
rng = np.random.default_rng(0)
be.reserve(max(sizes))
for B in sizes:
    a = torch.from_numpy(rng.integers(-2**31, 2**31, (B, be.W), dtype=np.int32)).cuda()
    be.bootstrap(a, 1 << 20); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); be.bootstrap(a, 1 << 20); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    l = be.last_launch()
    print("%s B %5d  %-15s wpb %d  blind rotate %.3f ms  keyswitch %.3f ms  wall %.3f ms  (%.1f us per ciphertext)"
          % (name, B, l["form"], l["waves_per_block"], *be.last_kernel_ms(), min(ts) * 1e3, min(ts) * 1e6 / B), flush=True)
