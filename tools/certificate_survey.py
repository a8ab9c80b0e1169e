"""How far is the FFT mode from ever rounding wrongly? Many 65,536-gate batches (fresh key per round,
fresh encryptions per batch, random gate type), each run in the FFT mode and again in the exact-NTT
mode and in the split-key mode, every output word compared on the device; per batch the rounding certificate
(largest |x - rint(x)| over all inverse-transform outputs of the FFT mode). An error of +-1 needs a distance > 0.5.

  python tools/certificate_survey.py [params] [keys] [batches_per_key] [gates] [seed_base]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import redsec_amd
from redsec_amd import client

params = sys.argv[1] if len(sys.argv) > 1 else "default128"
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 3
per_key = int(sys.argv[3]) if len(sys.argv) > 3 else 8
G = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
S = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # shifts every key / encryption seed: disjoint from earlier surveys
gates = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR"]
p = redsec_amd.params(params)
certs, mismatched_words, split_mismatched, total = [], 0, 0, 0
t0 = time.time()
for k in range(n_keys):
    sk = client.SecretKeySet(params, seed=1000 + S + k)
    be = redsec_amd.Backend(p, 0)
    be.load_keys(sk.bk, sk.ksk)
    be.reserve(G)
    for b in range(per_key):
        rng = np.random.default_rng(77 * k + b + 1000003 * S)
        a = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G), seed=5000 + 100 * k + 2 * b + 10007 * S)).cuda()
        c = torch.from_numpy(sk.encrypt_bits(rng.integers(0, 2, G), seed=5001 + 100 * k + 2 * b + 10007 * S)).cuda()
        op = gates[(k * per_key + b) % len(gates)]
        be.set_mode("fft")
        be.rounding_certificate(reset=True)
        out_f = be.gate(op, a, c)
        torch.cuda.synchronize()
        certs.append(be.rounding_certificate(reset=True))
        be.set_mode("exact")
        out_e = be.gate(op, a, c)
        mismatched_words += int((out_f != out_e).sum().item())
        be.set_mode("split")                                   # the third arithmetic: split-key FFT, exact by an a-priori bound
        split_mismatched += int((be.gate(op, a, c) != out_e).sum().item())
        total += G
        print("key %d batch %d %-4s certificate %.6f mismatched words so far %d (%.0f s)" % (k, b, op, certs[-1], mismatched_words, time.time() - t0), flush=True)
    del be
cmux = total * p.n
print(json.dumps({"params": params, "seed_base": S, "gates": total, "cmux_steps": cmux, "rounded_values": cmux * 2048,
                  "max_certificate": max(certs), "median_certificate": float(np.median(certs)),
                  "words_differing_from_exact_ntt_mode": mismatched_words,
                  "split_mode_words_differing_from_exact_ntt_mode": split_mismatched}))
