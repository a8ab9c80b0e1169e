#!/usr/bin/env python3
"""Where the eight waves of a latency-form workgroup spend a CMUX step (REDsec set, MNIST layer sizes).

  tools/build_variant.sh stamps8 . -DRS_DIAG=8     blind_rotate_coop8_kernel, B = 196 (one ciphertext per workgroup)
  tools/build_variant.sh stamps8l . -DRS_DIAG=256  blind_rotate_coop8_listed_kernel (STAMP_PARAMS=default128: the l < 4 deal)
  tools/build_variant.sh stampsduo . -DRS_DIAG=4     blind_rotate_duo_kernel, B = 1024 (four ciphertexts x two waves)
  REDSEC_HIP_LIB=$PWD/variants/lib_stamps8.so python tools/stamp_coop8.py [B]

Diagnostic build only (every stamp drains the wave's LDS reads): read the SHARES and the per-wave differences.
Cycles are s_memtime ticks (100 MHz constant clock on this chip: 10 ns each), per wave, summed over the steps of the launch."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import redsec_amd  # noqa: E402
from redsec_amd import client  # noqa: E402

PHASES_COOP8 = ["mask word", "rotated difference", "rows: forward + mac", "atomics issued", "barrier 1", "inverse + update", "barrier 2", "prologue / extract"]
PHASES_COOP8_LISTED = ["step entry + shared rotated difference", "barrier 0 (rotated difference complete)"] + PHASES_COOP8[2:]
PHASES_DUO = ["prologue + rotated difference", "digits + forward pair", "key wait + barrier 1", "multiply-accumulate", "barrier 2 + next quad",
              "partial exchange (2 barriers)", "inverse + update", "group prologue / extract"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 196
    lib = redsec_amd.load_library()
    if not hasattr(lib, "rs_debug_read_stamps"):
        raise SystemExit("not a -DRS_DIAG stamps build: set REDSEC_HIP_LIB to variants/lib_stamps8.so")
    params = os.environ.get("STAMP_PARAMS", "redsec_small_v2")
    sk = client.SecretKeySet(params, seed=7)
    be = redsec_amd.Backend(redsec_amd.params(params), device=0)
    be.load_keys(sk.bk, sk.ksk)
    x = torch.randint(-2**31, 2**31 - 1, (B, be.W), dtype=torch.int64).to(torch.int32).cuda()
    n_words = 256 * 8 * 8
    host = (C.c_ulonglong * n_words)()
    mu = 1 << 20
    be.set_timing(True)
    for _ in range(2):
        be.bootstrap(x, mu)
        torch.cuda.synchronize()
        assert lib.rs_debug_read_stamps(host, C.c_size_t(n_words)) == 0
    ms = be.last_kernel_ms()
    form = be.last_launch()
    PHASES = PHASES_DUO if form["form"] == "duo" else (PHASES_COOP8_LISTED if be.p.bk_l < 4 else PHASES_COOP8)
    blocks = min(256, -(-B // 4)) if form["form"] == "duo" else B
    a = np.frombuffer(host, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)[:blocks]
    n = be.p.n
    res = {"B": B, "form": form, "blind_rotate_ms_stamped_build": ms[0], "ticks_per_step_per_wave": {}}
    for w in range(8):
        res["ticks_per_step_per_wave"]["wave %d" % w] = {ph: round(float(a[:, w, k].mean() / n), 1) for k, ph in enumerate(PHASES)}
        res["ticks_per_step_per_wave"]["wave %d" % w]["total"] = round(float(a[:, w, :].sum(axis=1).mean() / n), 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
