"""Blind-rotation / keyswitch kernel times of small REDsec-set batches (the latency forms), HIP events on the launch stream.
usage: python tools/small_batch_probe.py [B ...] [--mode fft|split] [--reps 5] [--params redsec_small_v2]
Random input words (timing does not depend on values); prints one line per batch size: form, median and best ms."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import redsec_amd
from redsec_amd import client

ap = argparse.ArgumentParser()
ap.add_argument("B", nargs="*", type=int, default=[196, 1024])
ap.add_argument("--mode", default="fft")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--params", default="redsec_small_v2")
args = ap.parse_args()
p = redsec_amd.params(args.params)
be = redsec_amd.Backend(p, 0)
be.load_synthetic_keys(1)
be.set_mode(args.mode)
be.set_timing(True)
for B in args.B:
    x = torch.randint(-2**31, 2**31 - 1, (B, be.W), dtype=torch.int64).to(torch.int32).cuda()
    out = be.empty(B, be.W)
    be.bootstrap(x, 1 << 20, out=out); torch.cuda.synchronize()
    br, ks = [], []
    for _ in range(args.reps):
        be.bootstrap(x, 1 << 20, out=out); torch.cuda.synchronize()
        a, b = be.last_kernel_ms(); br.append(a); ks.append(b)
    print("B %d mode %s form %s: blind_rotate median %.3f best %.3f ms, keyswitch median %.3f ms" %
          (B, args.mode, be.last_launch()["form"], float(np.median(br)), min(br), float(np.median(ks))), flush=True)
be.close()
