#!/usr/bin/env python3
"""Gate-bootstrap microbenchmark (BASELINE.json configs[1]): 65,536 independent bootstrapped NANDs,
N=1024, TFHE default 128-bit parameters (n=630, l=3, Bgbit=7, t=8, basebit=2), one MI355X per rank.

A "step" is one pass of the hot path (gate pre-combination + blind rotation + sample extract +
keyswitch) over one batch of 65,536 ciphertext pairs that are already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gates G] [--params default128|redsec_small_v2]

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL); `python bench.py --gpus N` alone starts
that launcher itself as a child process. The gates of a batch are independent, so ranks shard them with full key
replicas: "weak" (default) = every rank runs its own 65,536 gates, "--scaling strong" = ONE batch of 65,536 split
evenly. The outputs are all-gathered over RCCL inside the timed region (the north star's "final RCCL gather"),
overlapped with the next step's kernels on RCCL's own stream; its un-overlapped cost is reported separately.

Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (blind rotation) against the HBM
roofline as the contract asks; `roofline_valu` prices it against the FP64 vector-ALU issue rate,
which is what actually bounds it. `cpu_baseline` is the exact-integer CPU
oracle (kind "port": TFHE itself is not available) timed on the host cores on a bounded sample of
the same inputs and keys, and doubles as the in-run parity check. At N = 1 the line also carries the second half of
BASELINE.json's metric, `mnist_sign1024x1`: the latency of one encrypted MNIST image (configs[2]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_GOPS = 256 * 4 * 16 * 2.4   # CUs x SIMDs x fp64 lanes/clk x GHz = 39,322 G lane-ops/s


def fp64_ops_per_bootstrap(n, l, fwd_red, inv_red, fused_ops):
    """FP64 VALU lane-operations per bootstrap. Per lane (16 coefficients)
    and CMUX: 2l forward transforms = fused stages 0-1 (`fused_ops`) + 8 stages x 8 butterflies x 8 ops
    + full reductions of 16 x 3 ops; pointwise 2 columns x 16 x 7 ops per row; one mid-reduction of
    32 x 3; 2 inverse transforms = 80 butterflies x 8 + (inv_red + 2) reductions + 16 conversions."""
    fwd = fused_ops + 8 * 8 * 8 + fwd_red * 48
    inv = 80 * 8 + (inv_red + 2) * 48 + 16
    per_lane = 2 * l * (fwd + 2 * 16 * 7) + 2 * inv + 96
    return n * per_lane * 64


def fp64_ops_per_bootstrap_fft(n, l):
    """Same count for the FFT mode (folded 512-point complex FFT, 8 complex points per lane, 9 stages
    of 4 butterflies): forward = 36 butterflies x 6 FMAs + 16 int->f64 conversions; pointwise = 2 columns
    x 8 points x 4 FMAs per row; inverse = 36 butterflies x 8 ops + 16 x 4 for rounding to the torus and
    the rounding certificate."""
    fwd = 36 * 6 + 16
    inv = 36 * 8 + 16 * 4
    per_lane = 2 * l * (fwd + 2 * 8 * 4) + 2 * inv
    return n * per_lane * 64


def fp64_ops_per_bootstrap_split(n, l):
    """Split-key FFT mode: the forward transforms of the FFT mode, twice its pointwise products (two key halves) and
    four inverse transforms instead of two."""
    fwd = 36 * 6 + 16
    inv = 36 * 8 + 16 * 4
    per_lane = 2 * l * (fwd + 2 * 2 * 8 * 4) + 4 * inv
    return n * per_lane * 64


def host_cpu_share():
    """CPUs granted to this process: the affinity mask, capped by the cgroup v2 / v1 CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


def host_cpu_model(path="/proc/cpuinfo"):
    """First `model name` of /proc/cpuinfo (print the CPU model beside omp_get_max_threads())."""
    try:
        for line in open(path):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def step_time_stats(stamps):
    """Per-step wall times in ms from perf_counter stamps [t0, end of step 0, end of step 1 ...]: (median, min, max).
    SURVEY.md section 8d asks for the median of >= 5 steps; `ms_per_step` stays the mean the bench contract defines."""
    import statistics
    d = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
    if not d:
        return None, None, None
    return statistics.median(d), min(d), max(d)


# nets/cifar/binarynet in the reference's own form: 463,872 sign bootstraps + 229,376 chained ORs of its
# max-pool loops (lib/BinFunc.cpp:880-925); the fused max-pool of this build needs 521,216 (DESIGN.md "Max-pool semantics")
CIFAR_BINARYNET_REFERENCE_BOOTSTRAPS = 463872 + 229376


def pmc_child(args, kernel_name, counters):
    """ONE rocprofv3 --pmc pass (counters only, as MI355X_MICROARCH.md prescribes: never together with a trace) over a child
    run of one step of the same workload on the same GPU; returns {counter: value of the largest dispatch of exactly the timed
    kernel form} or None (and says why on stderr) when rocprofv3 is missing or the pass fails."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        print("bench.py: rocprofv3 not on PATH", file=sys.stderr)
        return None
    out = tempfile.mkdtemp(prefix="redsec_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
           "--steps", "1", "--warmup", "0", "--cpu-sample", "0", "--no-exact-check", "--no-mnist", "--no-cifar", "--no-live-traffic",
           "--params", args.params, "--mode", args.mode, "--gates", str(args.gates), "--seed", str(args.seed)]
    env = dict(os.environ, TMPDIR="/tmp", REDSEC_BENCH_PMC_CHILD="1")
    try:
        # own session: on a timeout the WHOLE process group goes (rocprofv3 and the bench child that holds the GPU)
        proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            proc.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            raise
        rows = [row for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)) for row in csv.DictReader(open(f))]
        # the dispatches of exactly the timed kernel form (the gated exact-NTT recomputation is another blind_rotate_* kernel);
        # one step = one such dispatch, or two when the launcher cuts a last round off: the largest one is the launch priced
        vals = {}
        for counter in counters:
            hit = [float(row["Counter_Value"]) for row in rows
                   if row["Counter_Name"] == counter and row["Kernel_Name"].split("<")[0].split("(")[0].strip().endswith(kernel_name)]
            if proc.returncode != 0 or not hit:
                print("bench.py: rocprofv3 --pmc %s pass failed (rc %d, %d rows)" % (counter, proc.returncode, len(rows)), file=sys.stderr)
                return None
            vals[counter] = max(hit)
        keep = os.environ.get("REDSEC_BENCH_KEEP_PMC")      # a directory: the raw counter rows of the timed kernel are copied there (profiles/)
        if keep:
            os.makedirs(keep, exist_ok=True)
            with open(os.path.join(keep, "bench_pmc_%s.csv" % "_".join(counters)[:80]), "w") as f:
                f.write("Kernel_Name,Counter_Name,Counter_Value\n")
                for row in rows:
                    f.write("\"%s\",%s,%s\n" % (row["Kernel_Name"], row["Counter_Name"], row["Counter_Value"]))
        return vals
    except Exception as e:              # noqa: BLE001 -- a profiler problem must not fail the benchmark
        print("bench.py: rocprofv3 --pmc %s pass: %s" % (" ".join(counters), e), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_traffic(args, kernel_name):
    """Fabric-side bytes of ONE launch of the dominant kernel, measured now: rocprofv3 --pmc cannot be collected from inside a
    process, so two child runs (FETCH_SIZE, WRITE_SIZE: separate passes, counters only, as MI355X_MICROARCH.md prescribes) execute
    one step of the same workload on the same GPU; FETCH_SIZE (KiB) is doubled (the guide's gfx950 correction for wide coalesced
    reads), WRITE_SIZE is in KiB. Returns None when rocprofv3 is missing or a pass fails (the committed profile's figure is
    reported instead, labelled as such)."""
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        v = pmc_child(args, kernel_name, [counter])
        if v is None:
            return None
        vals.update(v)
    return int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024)


# LDS instructions of the lock-step kernel per wave and CMUX step (csrc/rs_bootstrap.hip blind_rotate_wg_kernel; counted in the
# ISA by tools/isa_scan.py and equal to SQ_INSTS_LDS / (waves x steps), MEASUREMENTS.md section 4.2): every transform passes its 8
# complex values per lane through two planar exchanges (re plane, im plane: 2 x 2 x 8 eight-byte stores, read back as 2 x 2 x 4
# sixteen-byte loads); a key row is 2 columns x 8 KB = 16 wave-wide 16-byte reads per digit row; ~83 accumulator accesses (the
# rotated difference's 32-bit reads, the update's read-modify-writes, the mask word). Cycles per wave-instruction on the CU's ONE
# LDS pipe from MI355X_MICROARCH.md "LDS": ds_write_b64 ~6 (the VGPR -> LDS store path, 85 B/clk), ds_read_b128 4, 32-bit
# accesses 2 (reads) to 4 (writes).
LDS_CYCLES = {"ds_write_b64": 6.0, "ds_read_b128": 4.0, "acc_32bit": 3.0}


def lds_model(l, split=False):
    transforms = 2 * l + (4 if split else 2)
    rows = 2 * l * (2 if split else 1)
    counts = {"plane_stores_ds_write_b64": 32 * transforms, "plane_loads_ds_read_b128": 16 * transforms,
              "key_reads_ds_read_b128": 16 * rows, "accumulator_32bit": 83}
    cycles = (counts["plane_stores_ds_write_b64"] * LDS_CYCLES["ds_write_b64"]
              + (counts["plane_loads_ds_read_b128"] + counts["key_reads_ds_read_b128"]) * LDS_CYCLES["ds_read_b128"]
              + counts["accumulator_32bit"] * LDS_CYCLES["acc_32bit"])
    return counts, cycles


def _phase(ct, key):
    """Torus phase b - <a, s> of every ciphertext of a device slab [B][W], as signed 32-bit values in int64."""
    ph = ct[:, -1].long() - (ct[:, :-1].long() * key).sum(dim=1)
    return ((ph + (1 << 31)) % (1 << 32)) - (1 << 31)


def sign_agreement(stages, lwe_key, device, strong=32, N=1024):
    """how many hidden units of an encrypted run carry the sign the plaintext network
    computes (its logits are pinned to the reference's plaintext build, tests/golden). `stages`: [(name, input slab(s) of the
    bootstrapped stage, its output slab, plaintext pre-activations or None, plaintext +-1 bits)]. Three fractions per stage:
    `agree` = encrypted output sign == plaintext bit; `agree_strong` = the same over units whose PLAINTEXT |pre-activation| >= 32
    message steps (weak-margin units flip under the 4096-level / 2N = 2048 mod-switch in any TFHE implementation, SURVEY hard part
    7); `bootstrap_agree` = output sign == sign of the phase of the stage's OWN encrypted input (what the bootstrap itself does,
    independent of flips inherited from earlier layers), and the same over inputs at least 32 steps from a decision boundary.
    `bootstrap_agree_predicted` is what ANY exact TFHE implementation does to these inputs, from first principles: the bootstrap
    decides the sign of the phase AFTER modSwitchFromTorus32(., 2N) of the n + 1 words (lib/GPU/gates.cu:39-42 corroborates the
    rounding), i.e. of phase + e with e the sum of the rounding errors of b and of the a_i under a key bit 1 (binary key: h of the n),
    each uniform within half a step of 2^32 / 2N: standard deviation sqrt((h + 1) / 12) steps of 2^21 -- about 7.7 message steps for
    the shipped set (n = 350, h ~ 175). The expected fraction of inputs that keep
    their sign is the mean of Phi(d / sigma), d = the input's distance to the nearest decision boundary. Measured beside predicted
    shows whose flips these are: the parameter set's, not this backend's."""
    import math
    import torch
    key = torch.from_numpy(lwe_key.astype("int64")).to(device)
    weight = int((lwe_key != 0).sum())          # binary key: only the words under a 1 carry their rounding error into the phase (+ the b word)
    sigma = math.sqrt((weight + 1) / 12.0) * (1 << 32) / (2 * N)       # torus32 units
    per, tot = [], {"units": 0, "agree": 0, "strong": 0, "agree_strong": 0, "bs_agree": 0, "bs_strong": 0, "bs_agree_strong": 0, "pred": 0.0}
    for name, ins, out, pre, bits in stages:
        enc = torch.where(_phase(out, key) >= 0, 1, -1)
        pb = torch.from_numpy(bits.astype("int64")).to(device)
        ph_in = _phase(ins[0], key)
        if len(ins) == 2:                                      # bootsOR: (0, 1/8) + a + b
            ph_in = ((ph_in + _phase(ins[1], key) + (1 << 29) + (1 << 31)) % (1 << 32)) - (1 << 31)
        own = torch.where(ph_in >= 0, 1, -1)
        far = (ph_in.abs() >= (strong << 20)) & (ph_in.abs() <= (1 << 31) - (strong << 20))
        agree = enc == pb
        dist = torch.minimum(ph_in.abs(), (1 << 31) - ph_in.abs()).double()      # to the boundary at 0 or at 1/2
        p_keep = 0.5 * (1.0 + torch.erf(dist / (sigma * math.sqrt(2.0))))
        # a TRIVIAL input (a = 0: a channel whose ternary weights are all zero leaves the bias alone) carries no rounding noise at all
        trivial = (ins[0][:, :-1] == 0).all(dim=1) if len(ins) == 1 else torch.zeros_like(ph_in, dtype=torch.bool)
        # ... so its fate is decided by the mod-switch of its b word alone: modSwitchFromTorus32(b, 2N) = (b + 2^(31 - log2 2N)) >> (32 - log2 2N),
        # +mu for a result in [0, N) (a bias of -1 message step rounds to slot 0 and comes out positive: a deterministic flip)
        sh = 32 - (2 * N).bit_length() + 1
        slot = ((ph_in + (1 << (sh - 1))) >> sh) & (2 * N - 1)
        kept = ((slot < N) == (ph_in >= 0)).double()
        p_keep = torch.where(trivial, kept, p_keep)
        pred = float(p_keep.sum())
        tot["pred"] += pred
        rec = {"stage": name, "units": int(enc.numel()), "trivial_inputs": int(trivial.sum()), "agree": round(float(agree.float().mean()), 5),
               "bootstrap_agree": round(float((enc == own).float().mean()), 5), "bootstrap_agree_predicted": round(pred / max(1, enc.numel()), 5),
               "bootstrap_agree_strong_input": round(float((enc == own)[far].float().mean()), 5) if bool(far.any()) else None}
        tot["units"] += int(enc.numel()); tot["agree"] += int(agree.sum())
        tot["bs_agree"] += int((enc == own).sum()); tot["bs_strong"] += int(far.sum()); tot["bs_agree_strong"] += int((enc == own)[far].sum())
        if pre is not None:
            st = torch.from_numpy((abs(pre) >= strong)).to(device)
            rec["strong_units"] = int(st.sum())
            rec["agree_strong"] = round(float(agree[st].float().mean()), 5) if bool(st.any()) else None
            tot["strong"] += int(st.sum()); tot["agree_strong"] += int(agree[st].sum())
        per.append(rec)
    return {"strong_means": "|pre-activation| >= %d message steps of 1/4096" % strong, "hidden_units": tot["units"],
            "agree": round(tot["agree"] / max(1, tot["units"]), 5),
            "agree_strong": round(tot["agree_strong"] / max(1, tot["strong"]), 5), "strong_units": tot["strong"],
            "bootstrap_agree": round(tot["bs_agree"] / max(1, tot["units"]), 5),
            "bootstrap_agree_predicted": round(tot["pred"] / max(1, tot["units"]), 5),
            "predicted_from": "mod-switch rounding noise alone: sigma = sqrt((h + 1) / 12) steps of 2^32 / 2N (h = %d key bits set) = %.1f message steps; mean of Phi(distance to the decision boundary / sigma)" % (weight, sigma / (1 << 20)),
            "bootstrap_agree_strong_input": round(tot["bs_agree_strong"] / max(1, tot["bs_strong"]), 5), "strong_inputs": tot["bs_strong"],
            "per_stage": per}


def redsec_set_legs(device_index, gates, with_cifar=True, with_cpu=True):
    """The second half of BASELINE.json's metric (configs[2]): ONE encrypted MNIST sign1024x1 image, device-resident, through
    the layer chain of redsec_amd/nets.py on the parameter set REDsec ships (1,220 bootstraps in batches of 196 and 1,024;
    trained weights and a bundled test image from tests/golden). The same image is pushed through the split-key mode as
    well: the 10 logit ciphertexts must be equal word for word."""
    import numpy as np
    import torch
    import redsec_amd
    from redsec_amd import client, nets
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import plain_model as pm
    sk = client.SecretKeySet("redsec_small_v2", seed=7)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), device=device_index)
    be.load_keys(sk.bk, sk.ksk)
    enc = nets.EncryptedMnist(be, pm.load_net("sign1024x1"))
    labels, pixels = pm.load_images()
    ct = torch.from_numpy(sk.encrypt_image(pixels[1], seed=5)).cuda(device_index)

    def timed(reps):
        for _ in range(2):
            out = enc.run(ct)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = enc.run(ct)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return out, 1e3 * float(np.median(ts))
    out_f, ms_f = timed(5)
    be.set_mode("split")
    out_s, ms_s = timed(3)
    be.set_mode("fft")
    logits = sk.decrypt_ints(out_f.cpu().numpy())
    res = {"ms_per_image": round(ms_f, 3), "unit": "ms", "images_per_call": 1, "bootstraps_per_image": 196 + 1024,
           "params": "redsec_small_v2", "mode": "fft", "split_mode_ms_per_image": round(ms_s, 3),
           "logit_ciphertexts_equal_in_split_mode": bool(torch.equal(out_f, out_s)),
           "data": "bundled MNIST test image, trained sign1024x1 weights (tests/golden)"}
    res["encrypted_argmax"] = int(np.argmax(logits))
    res["label"] = int(labels[1])
    ptaps, etaps = {}, {}
    res["plaintext_argmax"] = int(np.argmax(pm.forward(enc.net, pixels[1], ptaps)))
    enc.run(ct, taps=etaps)
    res["sign_agreement"] = sign_agreement([("layer%d" % k, (etaps["pre%d" % k],), etaps["bits%d" % k], ptaps["pre%d" % k], ptaps["bits%d" % k]) for k in (0, 1)],
                                           sk.lwe_key, ct.device)
    if with_cpu:
        # BASELINE configs[0], the CPU plumbing baseline, beside configs[2] in the same run: the SAME encrypted image under the SAME key
        # through the oracle's layer chain (tests/oracle_net.py; its FP64-FFT product path, OpenMP over the gates of a layer) on this box's
        # host cores -- and the ten logit ciphertexts it returns must equal the GPU's word for word (the checker checking the product:
        # nothing measured above went through it)
        import oracle_lib as ol
        import oracle_net

        class _K:
            pass
        k = _K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
        cores = host_cpu_share()
        ol.lib().ro_set_threads(cores)
        octx = ol.Ctx(k)
        octx.set_fft(True)
        t0 = time.perf_counter()
        out_cpu = oracle_net.run(octx, enc.net, sk.encrypt_image(pixels[1], seed=5))
        cpu_s = time.perf_counter() - t0
        res["cpu_baseline"] = {"s_per_image": round(cpu_s, 3), "cores": int(cores), "omp_threads": int(ol.lib().ro_max_threads()), "cpu_model": host_cpu_model(),
                               "kind": "port", "bootstraps_per_s": round(1220 / cpu_s, 1),
                               "logit_ciphertexts_equal_gpu": bool(np.array_equal(out_cpu, out_f.cpu().numpy())),
                               "what": "BASELINE configs[0] (nets/mnist/sign1024x1, 1 encrypted image, CPU): the oracle's layer chain on the host cores, same key and image"}
    # "also run the REDsec set" -- the same 65,536-NAND step on the shipped parameters
    rng = np.random.default_rng(11)
    ba, bb = rng.integers(0, 2, gates), rng.integers(0, 2, gates)
    ca = torch.from_numpy(sk.encrypt_bits(ba, seed=21)).cuda(device_index)
    cb = torch.from_numpy(sk.encrypt_bits(bb, seed=22)).cuda(device_index)
    out = be.empty(gates, be.W)
    be.gate("NAND", ca, cb, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        be.gate("NAND", ca, cb, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    ok = bool(np.array_equal(sk.decrypt_bits(out.cpu().numpy()), 1 - (ba & bb)))
    nands = {"value": round(gates / dt, 1), "unit": "bootstraps/s", "ms_per_step": round(1e3 * dt, 3), "steps": 2, "gates": int(gates),
             "params": "redsec_small_v2 (n=350 N=1024 l=10 Bgbit=3 t=9 basebit=3)", "mode": "fft", "kernel_form": be.last_launch()["form"],
             "all_outputs_decrypt_to_nand": ok, "fft_rounding_certificate": round(be.rounding_certificate(), 6)}
    cifar = cifar_leg(be, sk, device_index) if with_cifar else None
    be.close()
    return res, nands, cifar


class _StageTimer:
    """Backend proxy for the per-stage breakdown of an encrypted image: bootstrapped calls report the HIP-event kernel times of
    rs_last_kernel_ms (blind rotation, keyswitch), every other stage call is bracketed by events on the launch stream."""
    BOOT = ("bootstrap", "gate_mu", "gate", "bootstrap_lut")
    LINEAR = ("sumpool", "conv_ternary", "linear_fc", "gather_rows", "lincomb")

    def __init__(self, be):
        self._be = be
        self.blind_rotate_ms = self.keyswitch_ms = self.linear_ms = 0.0
        self.bootstraps = 0

    def __getattr__(self, name):
        import torch
        f = getattr(self._be, name)
        if name in self.BOOT:
            def boot(x, *a, **kw):
                out = f(x, *a, **kw)
                br, ks = self._be.last_kernel_ms()
                self.blind_rotate_ms += br; self.keyswitch_ms += ks; self.bootstraps += int(x.shape[0])
                return out
            return boot
        if name in self.LINEAR:
            def lin(*a, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = f(*a, **kw)
                e1.record(); e1.synchronize()
                self.linear_ms += e0.elapsed_time(e1)
                return out
            return lin
        return f


def cifar_leg(be, sk, device_index):
    """BASELINE configs[3]: ONE encrypted CIFAR-10 image through nets/cifar/binarynet (net.cpp:114-209: six 3x3 convolutions
    128-128-256-256-512-512 with a 2x2 max-pool after every second one, FC 1024-1024-10), device-resident, on the parameter set
    REDsec ships; the fused max-pool form (DESIGN.md "Max-pool semantics": 521,216 bootstraps, largest launch 131,072). Run 1 takes
    the per-stage kernel times (and warms the allocator), run 2 is the wall-clock figure."""
    import numpy as np
    import torch
    from redsec_amd import nets
    import plain_model as pm
    net = pm.CifarNet("binarynet")
    labels, pix = pm.load_cifar_images()
    i = 13                                                  # the clearest correctly classified bundled image (plaintext margin 292)
    ct = torch.from_numpy(sk.encrypt_image(pix[i], seed=4)).cuda(device_index)
    timer = _StageTimer(be)
    be.set_timing(True)
    nets.EncryptedCifar(timer, net).run(ct)
    be.set_timing(False)
    enc = nets.EncryptedCifar(be, net)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = enc.run(ct)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    be.set_mode("split")
    t0 = time.perf_counter()
    out_s = enc.run(ct)
    torch.cuda.synchronize()
    ms_s = 1e3 * (time.perf_counter() - t0)
    be.set_mode("fft")
    logits = sk.decrypt_ints(out.cpu().numpy())
    taps, ptaps = [], {}
    plain = pm.cifar_forward(net, pix[i], ptaps)
    out_t = enc.run(ct, taps=taps)                            # third run: the stage slabs for the agreement statistics
    agreement = sign_agreement([(r["name"], r["inputs"], r["out"]) + ptaps[r["name"]] for r in taps], sk.lwe_key, ct.device)
    largest = max(int(r["out"].shape[0]) for r in taps)
    del taps
    return {"ms_per_image": round(ms, 1), "unit": "ms", "bootstraps": timer.bootstraps, "bootstraps_reference_count": CIFAR_BINARYNET_REFERENCE_BOOTSTRAPS,
            "largest_launch": largest, "maxpool": enc.maxpool,
            "blind_rotate_ms": round(timer.blind_rotate_ms, 1), "keyswitch_ms": round(timer.keyswitch_ms, 1), "linear_ms": round(timer.linear_ms, 1),
            "bootstraps_per_s": round(timer.bootstraps / (ms * 1e-3), 1), "argmax": int(np.argmax(logits)), "label": int(labels[i]),
            "plaintext_argmax": int(np.argmax(plain)), "logit_correlation_with_plaintext": round(float(np.corrcoef(logits, plain)[0, 1]), 4),
            "sign_agreement": agreement, "rerun_logit_ciphertexts_equal": bool(torch.equal(out, out_t)),
            "class_note": "kernel-level parity is exact (tests/test_gpu_cifar.py: every bootstrapped stage = the oracle word for word, every linear stage = numpy); "
                          "the CLASS of one encrypted image rests on weak-margin units: bootstrap_agree_strong_input shows the bootstraps deciding every clear input "
                          "as its sign, agree < 1 is what 4096 message levels through the 2N = 2048 mod-switch do to inputs near a boundary (SURVEY hard part 7; "
                          "tools/cifar_agreement.py, profiles/r04: class equal to the plaintext class in 5 of 12 runs of the clearest images)",
            "params": "redsec_small_v2", "mode": "fft", "split_mode_ms_per_image": round(ms_s, 1),
            "logit_ciphertexts_equal_in_split_mode": bool(torch.equal(out, out_s)),
            "fft_rounding_certificate": round(be.rounding_certificate(), 6),
            "data": "bundled CIFAR-10 test image, trained binarynet weights (tests/golden)"}


def cifar_batch_leg(local_rank, rank, world, dist_on, rehearsal, net_name, n_images):
    """BASELINE configs[4]: a batch of encrypted CIFAR images, ONE IMAGE PER GPU (image-parallel replicas: full key replica on
    every rank, sharding.image_parallel), the 10 x W logit words of every image gathered over RCCL INSIDE the timed region.
    Inputs are resident in HBM before timing; one untimed warm-up image per rank; the batch is timed once, barrier +
    synchronize on both sides, maximum over ranks. Afterwards every rank re-runs the image of its NEIGHBOUR rank by itself and
    compares with what the gather delivered, word for word (a bootstrap's output depends on its input and the key only, so
    where an image ran cannot show): `logits_equal_single_gpu`. The reference's shape: enc_segs[NUM_GPUS], one host thread per
    GPU, no merge step (lib/GPU/Layer.cuh:15,22-37, nets/mnist/sign1024x1/main.cu:81-83)."""
    import numpy as np
    import torch
    import redsec_amd
    from redsec_amd import client, nets, sharding
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import plain_model as pm
    dev = torch.device("cuda", local_rank)
    sk = client.SecretKeySet("redsec_small_v2", seed=7)             # the same key on every rank (seeded)
    be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), device=local_rank)
    be.load_keys(sk.bk, sk.ksk)
    net = pm.CifarNet(net_name)
    enc = nets.EncryptedCifar(be, net)
    labels, pix = pm.load_cifar_images()
    n = n_images if n_images > 0 else world
    images = [(k + 1) % len(labels) for k in range(n)]              # image 0 of the bundle is misclassified by the plaintext net too
    mine = sharding.image_assignment(n, rank, world)
    check = [(k + 1) % n for k in mine]                             # the neighbour's images, re-run after the timed region
    cts = {k: torch.from_numpy(sk.encrypt_image(pix[images[k]], seed=100 + images[k])).to(dev) for k in set(mine) | set(check)}
    if cts:                                                         # (a rank of a batch smaller than the job has neither an image nor a check)
        enc.run(cts[mine[0] if mine else check[0]])                 # warm-up: allocator, first launches
    torch.cuda.synchronize()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    logits, t_compute, t_gather = sharding.image_parallel(lambda k: enc.run(cts[k]), list(range(n)), (10, be.W), force=dist_on, device=dev)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [{"rank": rank, "images": [images[k] for k in mine], "compute_ms": round(1e3 * t_compute, 1), "gather_ms": round(1e3 * t_gather, 3)}]
    ok = all(bool(torch.equal(enc.run(cts[k]), logits[k])) for k in check)
    fallbacks = int(be.fft_fallbacks())        # calls recomputed exactly on the device: legitimate, reported beside (not folded into) the comparison
    if dist_on:
        tm = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        fb = torch.tensor([fallbacks], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
        fallbacks = int(fb.item())
        every = [None] * world
        dist.all_gather_object(every, per_rank[0])
        per_rank = every
    res = None
    if rank == 0:
        dec = [sk.decrypt_ints(logits[k].cpu().numpy()) for k in range(n)]
        plain = [pm.cifar_forward(net, pix[images[k]]) for k in range(n)]
        res = {"workload": "nets/cifar/%s, %d encrypted image(s), image-parallel over %d GPU(s) (one each when equal), logits gathered" % (net_name, n, world),
               "images": n, "s_per_batch": round(elapsed, 3), "images_per_s": round(n / elapsed, 4),
               "gather_ms": round(max(r["gather_ms"] for r in per_rank), 3), "gather_words_per_image": 10 * be.W,
               "collective": "none (one rank, no process group)" if not dist_on else ("all_gather_into_tensor over " + ("gloo (one-GPU rehearsal)" if rehearsal else "nccl (RCCL)")),
               "inside_timed_region": True, "per_rank_ms": per_rank, "logits_equal_single_gpu": ok, "fft_fallbacks": fallbacks,
               "encrypted_argmax": [int(np.argmax(d)) for d in dec], "plaintext_argmax": [int(np.argmax(q)) for q in plain],
               "labels": [int(labels[images[k]]) for k in range(n)],
               "logit_correlation_with_plaintext": [round(float(np.corrcoef(d, q)[0, 1]), 3) for d, q in zip(dec, plain)],
               "params": "redsec_small_v2", "mode": "fft", "maxpool": enc.maxpool, "fft_rounding_certificate": round(be.rounding_certificate(), 6),
               "data": "bundled CIFAR-10 test images, trained weights (tests/golden)"}
    be.close()
    return res


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process has
    not touched the GPU yet: no torch.cuda call, no HIP call) and leave with its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates", type=int, default=65536, help="gates per step: per rank (weak scaling) or in total (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank runs --gates gates; strong = --gates gates are split across the ranks "
                         "(sharding.shard_range). Either way the outputs are all-gathered over RCCL inside the timed region, "
                         "overlapped with the next step's kernels (--no-gather leaves the slices where they are)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--params", default="default128", choices=["default128", "redsec_small_v2"])
    ap.add_argument("--mode", default="fft", choices=["fft", "exact", "split"],
                    help="ring arithmetic of the blind rotation: fft = FP64 complex FFT rounded to the integer result, every call "
                         "followed by its certificate-gated exact recomputation on the device (library default); exact = NTT over "
                         "a 51-bit prime (exact by construction)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="gates timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-fft-sample", type=int, default=-1, help="gates of the batch pushed through the oracle's FP64-FFT path for the CPU baseline and "
                    "compared word for word (default: 1,024 per host core, about 16 s; the whole batch -- 65,536 -- takes about a minute)")
    ap.add_argument("--no-exact-check", action="store_true", help="skip the exact-NTT mode leg (its throughput and the full-batch cross-check)")
    ap.add_argument("--no-mnist", action="store_true", help="skip the legs on the parameter set REDsec ships: encrypted-MNIST-image latency and the same NAND step (N = 1 only)")
    ap.add_argument("--no-cifar", action="store_true", help="skip the encrypted CIFAR binarynet image (BASELINE configs[3]; about 20 s at N = 1)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child runs "
                    "of one step (N = 1 only, about 40 s); the committed profile's figure is reported instead, labelled as such")
    ap.add_argument("--cifar-batch", default="auto", choices=["auto", "on", "off"],
                    help="BASELINE configs[4]: a batch of encrypted CIFAR images, one per GPU, logits gathered over RCCL inside the timed "
                         "region (JSON key cifar_batch). auto = on when N > 1 (at N = 1 the cifar_binarynet leg is the same image path)")
    ap.add_argument("--cifar-batch-net", default="binarynet", choices=["binarynet", "binarynet_small"])
    ap.add_argument("--cifar-batch-images", type=int, default=0, help="images in the batch (0 = one per GPU)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0xC0FFEE)
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and run the output all-gather also at ONE rank: "
                    "walks the N > 1 code path on a one-GPU box (tools/scale_sweep.sh checks it against the plain run)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): a scaling run must not silently measure another size" % (args.gpus, world))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # REDSEC_BENCH_REHEARSAL=1: walk the N>1 code path on a ONE-GPU box (ranks share device 0, gloo
    # instead of RCCL, which refuses two ranks on one device); never set by the driver.
    rehearsal = world > 1 and os.environ.get("REDSEC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import redsec_amd
    from redsec_amd import client, sharding

    # ---- keys (same on every rank: seeded) and synthetic inputs, resident in HBM before timing ----
    t_setup = time.time()
    sk = client.SecretKeySet(args.params, seed=args.seed)
    be = redsec_amd.Backend(redsec_amd.params(args.params), device=local_rank)
    be.load_keys(sk.bk, sk.ksk)
    be.set_mode(args.mode)
    strong = args.scaling == "strong" and world > 1
    if strong:
        # ONE batch of --gates gates, every rank takes a contiguous slice; slices padded to equal length for the gather
        total_gates = args.gates
        lo, hi = sharding.shard_range(total_gates, rank, world)
        width_rows = max(sharding.shard_range(total_gates, r, world)[1] - sharding.shard_range(total_gates, r, world)[0] for r in range(world))
        rng = np.random.default_rng(args.seed)
        bits_a_all, bits_b_all = rng.integers(0, 2, total_gates), rng.integers(0, 2, total_gates)
        bits_a, bits_b = bits_a_all[lo:hi], bits_b_all[lo:hi]
        G = hi - lo
    else:
        G = args.gates
        total_gates = G * world
        width_rows = G
        rng = np.random.default_rng(args.seed + 17 * rank)
        bits_a, bits_b = rng.integers(0, 2, G), rng.integers(0, 2, G)
    ca_h = sk.encrypt_bits(bits_a, seed=args.seed + 1000 + rank)
    cb_h = sk.encrypt_bits(bits_b, seed=args.seed + 2000 + rank)
    ca = torch.from_numpy(ca_h).to(dev)
    cb = torch.from_numpy(cb_h).to(dev)
    gather = dist_on and not args.no_gather
    outs = [torch.zeros((width_rows, be.W), dtype=torch.int32, device=dev) for _ in range(2 if gather else 1)]
    pipe = sharding.OverlappedGather(width_rows, be.W, torch.int32, dev) if gather else None
    be.reserve(G)
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()

    def step(k):
        o = outs[k % len(outs)]
        be.gate("NAND", ca, cb, out=o[:G])
        return o

    for k in range(args.warmup):
        step(k)
    if gather and args.warmup:                     # the first collective also builds the RCCL rings: keep it out of the timed region
        h, _ = pipe.launch(outs[0])
        pipe.wait(h)
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    be.set_timing(True)
    br_ms, ks_ms = [], []
    pending = []
    gathered = None
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = [t0]
    for k in range(args.steps):
        if gather and len(pending) >= 2:
            pipe.wait(pending[-2])                 # the buffer this step overwrites has left the GPU
        o = step(k)
        # HIP events recorded on the launch stream around each kernel; reading them waits for the
        # step (one step is one ~third-of-a-second batch, so this costs nothing measurable)
        b_ms, k_ms = be.last_kernel_ms()
        br_ms.append(b_ms); ks_ms.append(k_ms)
        stamps.append(time.perf_counter())        # the step's kernels have finished (the event read above waited for them)
        if gather:
            h, gathered = pipe.launch(o)           # runs on RCCL's stream while the next step computes
            pending.append(h)
    for h in pending[-2:]:
        pipe.wait(h)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    be.set_timing(False)
    last_br = sum(br_ms) / len(br_ms)   # average launch duration over the timed region
    last_ks = sum(ks_ms) / len(ks_ms)
    launch = be.last_launch()           # kernel form of the timed launches (the exact-mode leg below launches another)
    out = outs[(args.steps - 1) % len(outs)][:G]

    gather_ms = None
    if dist_on:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if gather:                                  # the same collective alone, un-overlapped, for the record
            barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            h, _ = pipe.launch(outs[0]); pipe.wait(h)
            torch.cuda.synchronize()
            gather_ms = 1e3 * (time.perf_counter() - t1)

    # per-rank kernel times of the timed region (HIP events on each rank's launch stream), for the scaling record
    kernels_per_rank = [{"rank": rank, "blind_rotate": round(last_br, 3), "keyswitch": round(last_ks, 3)}]
    if dist_on:
        import torch.distributed as dist
        every = [None] * world
        dist.all_gather_object(every, kernels_per_rank[0])
        kernels_per_rank = every

    # ---- correctness of the timed output: every gate decrypts to NAND(a, b) ----
    got = out.cpu().numpy()
    decrypt_ok = bool(np.array_equal(sk.decrypt_bits(got), 1 - (bits_a & bits_b)))
    gather_ok = None
    if gather:
        # what arrived from the other ranks: this rank's own block is where it belongs, and (strong scaling, where every
        # rank knows the whole batch's plaintext bits) the WHOLE gathered batch decrypts to the truth table
        full = gathered.cpu().numpy().reshape(world, width_rows, be.W)
        gather_ok = bool(np.array_equal(full[rank, :G], got))
        if strong:
            whole = np.concatenate([full[r, :sharding.shard_range(total_gates, r, world)[1] - sharding.shard_range(total_gates, r, world)[0]]
                                    for r in range(world)])
            gather_ok = gather_ok and bool(np.array_equal(sk.decrypt_bits(whole), 1 - (bits_a_all & bits_b_all)))

    # ---- the other arithmetic mode in the same run: exact-NTT throughput next to the FFT headline, and the WHOLE
    # batch compared word for word on the device (outside the timed region) ----
    all_equal_exact = None
    exact_mode = None
    split_mode = None
    if args.mode == "fft" and not args.no_exact_check:
        be.set_mode("exact")
        ref_exact = be.gate("NAND", ca, cb)                      # warm-up + the reference result
        torch.cuda.synchronize()
        all_equal_exact = bool(torch.equal(ref_exact, out))
        esteps = max(1, min(2, args.steps))
        barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(esteps):
            be.gate("NAND", ca, cb, out=ref_exact)
        torch.cuda.synchronize()
        e_elapsed = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            tm = torch.tensor([e_elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            e_elapsed = float(tm.item())
        exact_mode = {"value": round(total_gates * esteps / e_elapsed, 1), "unit": "bootstraps/s", "ms_per_step": round(1e3 * e_elapsed / esteps, 3),
                      "steps": esteps, "note": "RS_MODE_EXACT_NTT (exact by construction), same batch, no gather"}
        # ... and the split-key FFT mode (exact by an a-priori bound, include/redsec_hip.h): one step, whole batch compared
        be.set_mode("split")
        be.gate("NAND", ca, cb, out=ref_exact)
        torch.cuda.synchronize()
        split_equal = bool(torch.equal(ref_exact, out))
        barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        be.gate("NAND", ca, cb, out=ref_exact)
        torch.cuda.synchronize()
        s_elapsed = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            tm = torch.tensor([s_elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            s_elapsed = float(tm.item())
        split_mode = {"value": round(total_gates / s_elapsed, 1), "unit": "bootstraps/s", "ms_per_step": round(1e3 * s_elapsed, 3), "steps": 1,
                      "all_words_equal_full_batch": split_equal, "a_priori_bound": be.split_bound(),
                      "note": "RS_MODE_FFT_SPLIT (key in two 16-bit halves: exact by a worst-case bound), kernel form '%s'" % be.last_launch()["form"]}
        del ref_exact
        be.set_mode("fft")

    # ---- BASELINE configs[4]: image-parallel CIFAR batch (every rank takes part; own backend on the shipped parameter set) ----
    cifar_batch = None
    if args.cifar_batch == "on" or (args.cifar_batch == "auto" and world > 1):
        cifar_batch = cifar_batch_leg(local_rank, rank, world, dist_on, rehearsal, args.cifar_batch_net, args.cifar_batch_images)

    # FFT mode: largest distance of any inverse-transform output from an integer over the whole run
    # (exactness needs < 0.5), and how many calls the device recomputed exactly
    certificate = round(be.rounding_certificate(), 6) if args.mode == "fft" else None
    recomputed = be.fft_fallbacks() if args.mode == "fft" else None

    value = total_gates * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    step_median, step_min, step_max = step_time_stats(stamps)       # this rank's steps (kernel end to kernel end)

    if rank == 0:
        p = be.p
        info = be.info()
        # ---- HBM roofline of the dominant kernel (blind rotation), per launch ----
        # algorithmic bytes: key swept once per R resident ciphertexts (R as the launcher
        # reports it for the kernel form that actually ran), two input ciphertexts read, one extracted sample written.
        R = launch["resident"]
        bk_bytes = info["bk_device_bytes"]
        per_boot = bk_bytes / R + 2 * be.W * 4 + (p.N + 1) * 4
        alg_bytes = per_boot * G
        achieved = alg_bytes / (last_br * 1e-3) / 1e9
        # fabric-side traffic of the same launch: rocprofv3 --pmc passes cannot be collected from inside the process, so
        # this is READ FROM THE COMMITTED PROFILE of the same command (tools/pmc_traffic.py), and labelled as such
        traffic, traffic_src = None, None
        kernel_name = {"workgroup": "blind_rotate_wg_kernel", "duo": "blind_rotate_duo_kernel", "per_wave": "blind_rotate_kernel",
                       "coop2": "blind_rotate_coop_kernel", "coop4": "blind_rotate_coop_kernel", "coop8": "blind_rotate_coop8_kernel", "coop8_listed": "blind_rotate_coop8_listed_kernel",
                       "general": "gen_blind_rotate_kernel", "split_workgroup": "blind_rotate_wgs_kernel",
                       "split_coop": "blind_rotate_coops_kernel", "split_duo": "blind_rotate_duos_kernel"}[launch["form"]]
        under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if world == 1 and not args.no_live_traffic and not under_profiler and not os.environ.get("REDSEC_BENCH_PMC_CHILD"):
            traffic = live_traffic(args, kernel_name)
            if traffic is not None:
                traffic_src = "measured_in_this_run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one child run each of one step of the same workload on this GPU"
        for rnd in (() if traffic is not None else ("r04", "r03", "r02", "r01")):
            try:
                path = os.path.join("profiles", rnd, "pmc_traffic.json")
                pmc = json.load(open(os.path.join(ROOT, path)))
                traffic = pmc.get("%s_%d_%s" % (args.params, G, args.mode), {}).get("traffic_bytes")
                if traffic is not None:
                    traffic_src = "from_committed_profile:" + path
                    break
            except Exception:
                pass
        roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel_ms": round(last_br, 3), "algorithmic_bytes_per_launch": int(alg_bytes),
                    "resident_ciphertexts_per_key_sweep": R, "waves_per_workgroup": launch["waves_per_block"]}
        # SURVEY.md section 8(d)'s whole-step figure: + the keyswitch term KS_rows * W * 4 / T, with KS_rows = N t (1 - 1/base) rows
        # touched per ciphertext and T = the 256 ciphertexts of a keyswitch workgroup that share every key tile through LDS
        # (csrc/rs_kernels.hip keyswitch_tiled*_kernel), + the extracted sample read back and the output ciphertext written
        T_ks = 256
        ks_rows = p.N * p.ks_t * (1.0 - 1.0 / (1 << p.ks_basebit))
        ks_per_boot = ks_rows * be.W * 4 / T_ks + (p.N + 1) * 4 + be.W * 4
        step_bytes = (per_boot + ks_per_boot) * G
        step_ms = last_br + last_ks
        roofline_step = {"bound": "hbm", "kernels": [kernel_name, "keyswitch"], "algorithmic_bytes_per_step": int(step_bytes),
                         "keyswitch_bytes_per_bootstrap": int(ks_per_boot), "blind_rotate_bytes_per_bootstrap": int(per_boot),
                         "ciphertexts_sharing_a_keyswitch_tile": T_ks, "kernels_ms": round(step_ms, 3),
                         "achieved": round(step_bytes / (step_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)}
        fwd_red, inv_red, fused = (2, 3, 36) if p.bk_l == 3 else (0, 1, 48)
        ops_per = fp64_ops_per_bootstrap_fft(p.n, p.bk_l) if args.mode == "fft" else \
            (fp64_ops_per_bootstrap_split(p.n, p.bk_l) if args.mode == "split" else fp64_ops_per_bootstrap(p.n, p.bk_l, fwd_red, inv_red, fused))
        ops = ops_per * G
        valu = ops / (last_br * 1e-3) / 1e9
        roofline_valu = {"bound": "fp64-valu-issue", "achieved": round(valu, 1), "peak": round(FP64_VALU_PEAK_GOPS, 1),
                         "unit": "G fp64 lane-ops/s", "frac": round(valu / FP64_VALU_PEAK_GOPS, 4),
                         "fp64_ops_per_bootstrap": ops_per,
                         # what this formulation can reach at all: ops-limited bootstraps/s of the WHOLE step at issue fraction 1.0 and at
                         # the 0.82 of nominal the SIMDs were measured to sustain on a pure FP64 stream
                         "ceiling_value": {"at_frac_1.0": round(FP64_VALU_PEAK_GOPS * 1e9 / ops_per, 1),
                                           "at_sustained_0.82": round(0.82 * FP64_VALU_PEAK_GOPS * 1e9 / ops_per, 1), "unit": "bootstraps/s",
                                           "note": "blind rotation alone at that issue fraction, keyswitch not counted; the 1.0e6/s north-star figure is "
                                                   "beyond both for an FP64-carried transform product on this chip"}}

        # box calibration: the FP64 FMA rate this device sustains right now at the kernel's occupancy (boxes of one pool were seen
        # 5-6 % apart in it, and with it in every kernel of this path); the kernel's share of THAT is the box-independent figure
        try:
            fp64_now = be.fp64_rate() / 1e9
            roofline_valu["sustained_on_this_box"] = {"fp64_fma_G_lane_ops_per_s": round(fp64_now, 1), "frac_of_nominal_peak": round(fp64_now / FP64_VALU_PEAK_GOPS, 4),
                                                      "kernel_frac_of_it": round(valu / fp64_now, 4),
                                                      "note": "pure FMA stream, 8 waves per CU, measured after the timed steps (rs_debug_fp64_rate)"}
        except Exception as e:   # an older library without the tap: the line stays valid
            roofline_valu["sustained_on_this_box"] = {"error": str(e)[:120]}

        # ---- the third bound: the CU's LDS pipe (what explains an FP64 issue fraction of 0.56) ----
        roofline_lds = None
        if launch["form"] in ("workgroup", "split_workgroup") and args.mode in ("fft", "split"):
            counts, cyc_wave = lds_model(p.bk_l, split=args.mode == "split")
            waves = launch["waves_per_block"]
            cus = info["num_cus"]
            rounds = -(-G // (waves * cus))
            steps_per_cu = rounds * p.n                      # CMUX steps a CU walks per launch (identity steps, bara = 0, are 1 in 2N)
            pmc = None
            if world == 1 and not args.no_live_traffic and not under_profiler and not os.environ.get("REDSEC_BENCH_PMC_CHILD"):
                sq = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
                pmc = pmc_child(args, kernel_name, sq)
                gui = pmc_child(args, kernel_name, ["GRBM_GUI_ACTIVE"]) if pmc else None
                if pmc and gui:
                    pmc.update(gui)
            xcds = 8
            clk_ghz = (pmc["GRBM_GUI_ACTIVE"] / xcds / (last_br * 1e-3) / 1e9) if pmc and "GRBM_GUI_ACTIVE" in pmc else 2.4
            step_cycles = last_br * 1e-3 * clk_ghz * 1e9 / steps_per_cu
            lds_cycles = cyc_wave * waves
            fp64_cycles = ops_per / p.n / 64 * waves / 4 * 4.0       # per SIMD: (waves / 4) waves x FP64 wave-instructions x 4 cycles
            roofline_lds = {"bound": "lds-pipe (one per CU)", "kernel": kernel_name,
                            "lds_instructions_per_wave_and_cmux_step": counts, "cycles_per_wave_instruction": LDS_CYCLES,
                            "lds_pipe_cycles_per_cu_and_cmux_step": round(lds_cycles), "fp64_issue_cycles_per_simd_and_cmux_step": round(fp64_cycles),
                            "measured_cycles_per_cmux_step": round(step_cycles), "clock_ghz": round(clk_ghz, 3),
                            "clock_source": "GRBM_GUI_ACTIVE / 8 XCDs / kernel time (pmc child)" if pmc and "GRBM_GUI_ACTIVE" in pmc else "nominal",
                            "frac": round(lds_cycles / step_cycles, 4), "fp64_frac_same_clock": round(fp64_cycles / step_cycles, 4),
                            "sum_of_both_over_step": round((lds_cycles + fp64_cycles) / step_cycles, 4),
                            "note": "the step takes about the SUM of its FP64 issue time and its LDS-pipe time, not their maximum: beside a busy FP64 stream every "
                                    "LDS instruction costs the issuing SIMD 1.1-2.0 FMA slots (tools/lds_issue_bench.hip), and the 8 lock-step waves of a CU reach their "
                                    "exchange bursts together; fewer exchange bytes, not faster instructions, is what would move it (MEASUREMENTS.md section 4.2)"}
            if pmc:
                cu_cycles = (pmc.get("GRBM_GUI_ACTIVE", 0) / xcds) or (last_br * 1e-3 * 2.4e9)
                roofline_lds["pmc"] = {k: int(v) for k, v in pmc.items()}
                roofline_lds["pmc_derived"] = {
                    "SQ_ACTIVE_INST_LDS_over_SQ_BUSY_CYCLES": round(pmc["SQ_ACTIVE_INST_LDS"] / pmc["SQ_BUSY_CYCLES"], 4),
                    "SQ_ACTIVE_INST_VALU_over_SQ_BUSY_CYCLES": round(pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_BUSY_CYCLES"], 4),
                    # SQ_LDS_IDX_ACTIVE = all LDS-array cycles (MI355X_MICROARCH.md "LDS"), summed over the CUs
                    "lds_array_busy_frac_per_cu": round(pmc["SQ_LDS_IDX_ACTIVE"] / (cus * cu_cycles), 4),
                    "lds_instructions_per_wave_and_cmux_step": round(pmc["SQ_INSTS_LDS"] / (rounds * cus * waves * p.n), 1),
                    "valu_instructions_per_wave_and_cmux_step": round(pmc["SQ_INSTS_VALU"] / (rounds * cus * waves * p.n), 1),
                    "bank_conflict_cycles": int(pmc["SQ_LDS_BANK_CONFLICT"]),
                    "units": "SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_BUSY_CYCLES in cycles summed over the 32 shader engines",
                    "source": "measured_in_this_run: one rocprofv3 --pmc child run of one step of the same workload (counters only)"}

        # ---- CPU baseline + parity on a bounded sample of the same workload ----
        cpu = None
        parity = None
        if args.cpu_sample != 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as ol
            # threads = the CPUs this process may actually use: affinity mask capped by the cgroup CPU quota
            # (a GPU box grants 16 CPUs of its 256 hardware threads; 128 OpenMP threads on that share ran
            # 25 % slower than 16) -- and `cores` reports exactly that number
            cores = host_cpu_share()
            ol.lib().ro_set_threads(cores)
            # the CPU baseline is an N=1 figure (torch.distributed.run also pins OMP_NUM_THREADS=1);
            # multi-rank runs only keep a small in-run parity sample
            timed_baseline = world == 1
            sample = args.cpu_sample if args.cpu_sample > 0 else (max(256, 16 * cores) if timed_baseline else 4)   # exact product path: ~7 gates per second and core
            sample = min(sample, G)
            op = ol.params(args.params)

            class _K:
                pass
            k = _K(); k.p = op; k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
            octx = ol.Ctx(k)
            # (1) parity: the EXACT oracle path on `sample` gates
            ref = octx.gate_batch("NAND", ca_h[:sample], cb_h[:sample])
            parity = bool(np.array_equal(ref, got[:sample]))
            if timed_baseline:
                # (2) CPU baseline: the oracle's double-precision FFT path (the arithmetic class of TFHE's CPU
                # library; ~8x faster than the exact path and bit-equal to it) on a larger sample of the batch
                octx.set_fft(True)
                bsample = min(G, args.cpu_fft_sample if args.cpu_fft_sample > 0 else 1024 * cores)     # ~16 s of CPU work at ~16 ms per gate and core
                octx.gate_batch("NAND", ca_h[:min(cores, bsample)], cb_h[:min(cores, bsample)])  # warm caches / threads
                t1 = time.perf_counter()
                ref_fft = octx.gate_batch("NAND", ca_h[:bsample], cb_h[:bsample])
                cpu_s = time.perf_counter() - t1
                parity = parity and bool(np.array_equal(ref_fft, got[:bsample]))
                cpu = {"value": round(bsample / cpu_s, 3), "unit": "bootstraps/s", "cores": int(cores), "omp_threads": int(ol.lib().ro_max_threads()),
                       "cpu_model": host_cpu_model(), "per_core": round(bsample / cpu_s / cores, 2), "kind": "port",
                       "sample": "%d NAND gates of the same batch (same keys, same inputs), %.1f s wall on %d OpenMP threads; "
                                 "the oracle's FP64-FFT product path (exact after rounding, equal to the GPU output word for word); "
                                 "TFHE itself unavailable" % (bsample, cpu_s, cores)}

        # what "exact" means per arithmetic mode (include/redsec_hip.h)
        exactness = {
            "fft": {"status": "certificate-gated", "gate_distance": 0.25, "largest_rounding_distance_this_run": certificate,
                    "calls_recomputed_exactly_on_device": recomputed,
                    "statement": "every call is followed on the device by the exact-NTT kernels, which recompute it unless its largest rounding distance stayed below 1/4; "
                                 "an undetected wrong word needs an FFT error beyond 3/4 in a call whose every distance stayed below 1/4 (largest ever observed: 0.0156 = one unit in the last place at the magnitudes reached, "
                                 "over 2.9e14 rounded values, profiles/r02/r_certificate_survey_large.jsonl + profiles/r03/q_certificate_survey_*.jsonl + profiles/r04/{o,s,au,av,bl,bq}_certificate_survey_*.jsonl + profiles/r05/{k,m,s}_certificate_survey_*.jsonl: 48 times the largest deviation seen); no a-priori proof"},
            "split": {"status": "proved", "a_priori_bound": split_mode["a_priori_bound"] if split_mode else None, "needs": "< 1/2",
                      "statement": "worst-case FFT error bound derived in csrc/rs_general.h for the butterflies used; rounding is exact for every input"},
            "exact": {"status": "proved", "statement": "exact NTT over a 51-bit prime carried in FP64; every step exact by construction, schedule validated at rs_create"},
            "headline_mode": args.mode,
            "guaranteed_exact_throughput_form": "split (lock-step workgroup kernel on the split key); the exact-NTT mode is the per-wave form and serves as the gated recomputation path",
        }
        mnist, redsec_nands, cifar = redsec_set_legs(local_rank, G, not args.no_cifar, args.cpu_sample != 0) if (world == 1 and not args.no_mnist and args.params == "default128") else (None, None, None)

        line = {
            "metric": "gate bootstraps/sec (N=1024)", "value": round(value, 1), "unit": "bootstraps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_median": None if step_median is None else round(step_median, 3), "ms_per_step_min_max": None if step_median is None else [round(step_min, 3), round(step_max, 3)],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "int32 torus; ring products in f64 (%s)" % ({"fft": "complex FFT, exact after rounding", "exact": "exact NTT mod a 51-bit prime", "split": "complex FFT on a split key, exact by an a-priori bound"}[args.mode]),
            "data": "synthetic",
            "config": {"workload": "%d independent bootstrapped NAND gates per GPU per step, %s (n=%d N=%d l=%d Bgbit=%d t=%d basebit=%d)"
                                   % (G, args.params, p.n, p.N, p.bk_l, p.bk_Bgbit, p.ks_t, p.ks_basebit),
                       "gates_per_gpu": G, "total_gates": total_gates, "params": args.params, "mode": args.mode,
                       "parallelism": "gate-sharded x%d%s" % (world, "" if world == 1 else (", outputs all-gathered over RCCL, overlapped with the next step" if gather else ", no gather"))},
            "roofline": roofline, "roofline_step": roofline_step, "roofline_valu": roofline_valu, "roofline_lds": roofline_lds, "cpu_baseline": cpu, "exact_mode": exact_mode, "split_mode": split_mode,
            "mnist_sign1024x1": mnist, "cifar_binarynet": cifar, "cifar_batch": cifar_batch, "redsec_params_nands": redsec_nands,
            "collective": None if not gather else {"op": "all_gather_into_tensor", "bytes_per_rank": int(width_rows * be.W * 4),
                                                   "bytes_received_per_rank": int(world * width_rows * be.W * 4),
                                                   "ms_alone_unoverlapped": round(gather_ms, 3), "inside_timed_region": True,
                                                   "backend": "gloo (one-GPU rehearsal)" if rehearsal else "nccl (RCCL)"},
            "kernels_ms": {"blind_rotate": round(last_br, 3), "keyswitch": round(last_ks, 3)},
            "kernels_ms_per_rank": kernels_per_rank,
            "exactness": exactness,
            "checks": {"all_outputs_decrypt_to_nand": decrypt_ok, "bit_exact_vs_oracle_on_sample": parity,
                       "oracle_sample_gates": {"exact_path": int(sample) if args.cpu_sample != 0 else 0},
                       "all_words_equal_exact_ntt_mode_full_batch": all_equal_exact,
                       "fft_rounding_certificate": certificate, "calls_recomputed_exactly_on_device": recomputed,
                       "gathered_batch_ok": gather_ok},
            "setup_s": round(setup_s, 1),
        }
        print(json.dumps(line), flush=True)
    barrier()
    be.close()
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
