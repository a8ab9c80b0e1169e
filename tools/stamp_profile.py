#!/usr/bin/env python3
"""Where a wave of blind_rotate_wg_kernel spends its cycles: reads the per-wave phase sums of a -DRS_DIAG=1 build (csrc/rs_diag.h).

  tools/build_variant.sh stamps . -DRS_DIAG=1
  REDSEC_HIP_LIB=$PWD/variants/lib_stamps.so python tools/stamp_profile.py [default128|redsec_small_v2] [gates]
  tools/build_variant.sh stampswgs . -DRS_DIAG=2       (the split lock-step kernel instead)
  REDSEC_HIP_LIB=$PWD/variants/lib_stampswgs.so python tools/stamp_profile.py default128 16384 --split

Diagnostic build only: every stamp drains the wave's LDS reads, so the run is slower than the product; read the SHARES.
Cycles are shader cycles (s_memtime), per wave, summed over the CMUX steps of every ciphertext group the wave walked.
"""
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

PHASES = ["step prologue", "digits + forward pair", "key wait + barrier 1", "multiply-accumulate", "barrier 2 + next rows",
          "acc pre-read + inverse pair", "rounding + acc update", "group prologue / extract"]


PHASES_SPLIT = ["step prologue + rotated differences", "digits + forward transform", "key wait + barrier (low half)", "multiply-accumulate low",
                "key wait + barrier (high half)", "multiply-accumulate high", "two inverse pairs + update", "group prologue / extract"]


def main():
    global PHASES
    split = "--split" in sys.argv
    if split:
        sys.argv.remove("--split")
        PHASES = PHASES_SPLIT
    name = sys.argv[1] if len(sys.argv) > 1 else "default128"
    gates = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    lib = redsec_amd.load_library()
    if not hasattr(lib, "rs_debug_read_stamps"):
        raise SystemExit("not a -DRS_DIAG stamps build: set REDSEC_HIP_LIB to variants/lib_stamps.so")
    sk = client.SecretKeySet(name, seed=7)
    be = redsec_amd.Backend(redsec_amd.params(name), device=0)
    be.load_keys(sk.bk, sk.ksk)
    if split:
        be.set_mode("split")
    rng = np.random.default_rng(3)
    ba, bb = rng.integers(0, 2, gates), rng.integers(0, 2, gates)
    ca = torch.from_numpy(sk.encrypt_bits(ba, seed=1)).cuda()
    cb = torch.from_numpy(sk.encrypt_bits(bb, seed=2)).cuda()
    out = be.empty(gates, be.W)
    n_words = 256 * 8 * len(PHASES)
    host = (C.c_ulonglong * n_words)()
    be.gate("NAND", ca, cb, out=out)
    torch.cuda.synchronize()
    lib.rs_debug_read_stamps(host, C.c_size_t(n_words))     # discard the warm-up
    be.gate("NAND", ca, cb, out=out)
    torch.cuda.synchronize()
    assert lib.rs_debug_read_stamps(host, C.c_size_t(n_words)) == 0
    ok = bool(np.array_equal(sk.decrypt_bits(out.cpu().numpy()), 1 - (ba & bb)))
    a = np.frombuffer(host, dtype=np.uint64).reshape(256, 8, len(PHASES)).astype(np.float64)
    used = a.sum(axis=2) > 0
    per_wave = a[used]                                    # [waves, phases]
    tot = per_wave.sum(axis=1)
    n = be.p.n
    groups = -(-gates // 8)
    rounds = max(1, -(-groups // min(groups, be.info()["num_cus"])))
    steps = n * rounds
    res = {"params": name, "gates": gates, "waves": int(used.sum()), "outputs_ok": ok, "form": be.last_launch(),
           "cycles_per_wave_total_mean": float(tot.mean()), "cycles_per_cmux_step": float(tot.mean() / steps),
           "phases": {}}
    for k, ph in enumerate(PHASES):
        res["phases"][ph] = {"share": round(float(per_wave[:, k].sum() / tot.sum()), 4),
                             "cycles_per_cmux_step": round(float(per_wave[:, k].mean() / steps), 1)}
    # the two halves of a workgroup separately: waves 4-7 lose the issue arbitration to their SIMD partners 0-3 and set the pace
    for name_, sl in (("waves_0_3", slice(0, 4)), ("waves_4_7", slice(4, 8))):
        sel = a[:, sl, :][used[:, sl]]
        if sel.size:
            res["cycles_per_cmux_step_" + name_] = {ph: round(float(sel[:, k].mean() / steps), 1) for k, ph in enumerate(PHASES)}
    # spread between the two halves of a workgroup (waves 0-3 dispatched first, 4-7 second)
    first = a[:, :4, :][used[:, :4]].sum(axis=0) if used[:, :4].any() else None
    second = a[:, 4:, :][used[:, 4:]].sum(axis=0) if used[:, 4:].any() else None
    if first is not None and second is not None:
        res["waves_0_3_vs_4_7_share_of_wait_phases"] = {
            "waves_0_3": round(float((first[2] + first[4]) / first.sum()), 4), "waves_4_7": round(float((second[2] + second[4]) / second.sum()), 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
