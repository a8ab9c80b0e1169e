"""bench.py's host-side arithmetic on the CPU (the driver runs bench.py unattended at the end of a round: its helper functions must not
be first exercised there): the instruction-count models against their documented values, and `sign_agreement` -- the statistic
SURVEY.md section 8(d) asks for -- on a SIMULATED sign bootstrap: ciphertexts of known phase under a random binary key, the output
sign decided exactly as TFHE's mod-switch decides it (every word through modSwitchFromTorus32(., 2N), lib/GPU/gates.cu:39-42
corroborates the rounding). The fraction of sign-preserving bootstraps the statistic MEASURES on that simulation must match the
fraction it PREDICTS from the rounding-noise model to a percent -- which is the claim the bench line makes about the GPU run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_instruction_count_models():
    assert bench.fp64_ops_per_bootstrap_fft(630, 3) == 630 * 2480 * 64 == 99_993_600            # the 2,480 FP64 operations per lane and CMUX of DESIGN.md 4.2
    assert bench.fp64_ops_per_bootstrap_fft(350, 10) == 148_377_600
    counts, cycles = bench.lds_model(3)
    assert counts == {"plane_stores_ds_write_b64": 256, "plane_loads_ds_read_b128": 128, "key_reads_ds_read_b128": 96, "accumulator_32bit": 83}
    assert sum(counts.values()) == 563 and cycles == 256 * 6 + 224 * 4 + 83 * 3                 # 563 LDS instructions per wave and step (SQ_INSTS_LDS: 562.9)
    counts10, _ = bench.lds_model(10)
    assert sum(counts10.values()) == 704 + 352 + 320 + 83                                        # 1,459 (counters: 1,458)
    assert bench.host_cpu_share() >= 1


def test_record_helpers(tmp_path):
    """What VERDICT round 5 asked the line to carry: the CPU model beside the thread count, the median step beside the mean,
    the reference's own bootstrap count of CIFAR binarynet beside the fused form's."""
    f = tmp_path / "cpuinfo"
    f.write_text("processor\t: 0\nvendor_id\t: X\nmodel name\t: Some CPU 9654 96-Core @ 2.4GHz\nmodel name\t: other\n")
    assert bench.host_cpu_model(str(f)) == "Some CPU 9654 96-Core @ 2.4GHz"
    assert bench.host_cpu_model(str(tmp_path / "missing")) == "unknown"
    assert isinstance(bench.host_cpu_model(), str) and bench.host_cpu_model()
    med, lo, hi = bench.step_time_stats([10.0, 10.3, 10.61, 10.93, 11.5, 11.81])
    assert abs(med - 310.0) < 1e-6 and abs(lo - 300.0) < 1e-6 and abs(hi - 570.0) < 1e-6
    assert bench.step_time_stats([1.0]) == (None, None, None)
    assert bench.CIFAR_BINARYNET_REFERENCE_BOOTSTRAPS == 693248
    import oracle_lib as ol
    assert ol.lib().ro_max_threads() >= 1                                 # what `omp_threads` in cpu_baseline reports


def test_sign_agreement_on_a_simulated_bootstrap():
    import torch
    rng = np.random.default_rng(42)
    n, N, B = 350, 1024, 20000
    key = rng.integers(0, 2, n).astype(np.int32)
    mu = 1 << 20

    def encrypt(phase):                                          # phase: int64 [B] in torus32 units
        a = rng.integers(-2**31, 2**31, (len(phase), n), dtype=np.int64)
        b = phase + (a * key).sum(axis=1)
        return np.concatenate([a, b[:, None]], axis=1).astype(np.uint64).astype(np.uint32).view(np.int32)

    def bootstrap_sign(ct):                                      # the decision of tfhe_bootstrap_FFT, without its noise: sign of the mod-switched phase
        ms = ((ct.astype(np.int64) + (1 << 20)) >> 21) & (2 * N - 1)
        slot = (ms[:, -1] - (ms[:, :-1] * key).sum(axis=1)) % (2 * N)
        out = np.zeros_like(ct)
        out[:, -1] = np.where(slot < N, mu, -mu)
        return out

    stages = []
    for name, spread in (("weak inputs", 12), ("wider inputs", 60)):
        pre = rng.integers(-spread, spread + 1, B)               # pre-activations in message steps of 1/4096
        ct = encrypt(pre.astype(np.int64) << 20)
        stages.append((name, (torch.from_numpy(ct),), torch.from_numpy(bootstrap_sign(ct)), pre, np.where(pre >= 0, 1, -1)))
    # a stage of TRIVIAL inputs (a = 0): decided by the b word alone; a bias of -1 step rounds to slot 0 and flips, deterministically
    triv = np.zeros((4, n + 1), np.int32)
    triv[:, -1] = np.array([0, 5 << 20, -(1 << 20), -(7 << 20)], np.int64).astype(np.int32)
    stages.append(("trivial", (torch.from_numpy(triv),), torch.from_numpy(bootstrap_sign(triv)), np.array([0, 5, -1, -7]), np.array([1, 1, -1, -1])))
    res = bench.sign_agreement(stages, key, "cpu", strong=32, N=N)
    per = {r["stage"]: r for r in res["per_stage"]}
    for name in ("weak inputs", "wider inputs"):
        r = per[name]
        assert abs(r["bootstrap_agree"] - r["bootstrap_agree_predicted"]) < 0.012, r
        assert r["bootstrap_agree"] < 0.97                        # these inputs DO flip: the test is not vacuous
        assert r["bootstrap_agree_strong_input"] in (None, 1.0)
    assert per["weak inputs"]["bootstrap_agree"] < per["wider inputs"]["bootstrap_agree"]
    t = per["trivial"]
    assert t["trivial_inputs"] == 4 and t["bootstrap_agree"] == 0.75 and t["bootstrap_agree_predicted"] == 0.75
    assert "h = %d" % int(key.sum()) in res["predicted_from"]
    assert 0.0 < res["agree"] <= 1.0 and res["hidden_units"] == 2 * B + 4
