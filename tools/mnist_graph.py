#!/usr/bin/env python3
"""Encrypted sign1024x1 (one image, device-resident) replayed as ONE HIP graph: the layer chain's ~20 launches and memsets
are captured once on a side stream (torch.cuda.graph drives hipStreamBeginCapture) and replayed. Prints eager vs graph
latency and checks that the replayed logits equal the eager ones word for word.

  python tools/mnist_graph.py [net]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import redsec_amd
from redsec_amd import client, nets
import plain_model as pm

sk = client.SecretKeySet("redsec_small_v2", seed=7)
be = redsec_amd.Backend(redsec_amd.params("redsec_small_v2"), 0)
be.load_keys(sk.bk, sk.ksk)
net = pm.load_net(sys.argv[1] if len(sys.argv) > 1 else "sign1024x1")
enc = nets.EncryptedMnist(be, net)
labels, pixels = pm.load_images()
ct = torch.from_numpy(sk.encrypt_image(pixels[1], seed=5)).cuda()

def timed(fn, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3

ref = enc.run(ct).clone()
eager = timed(lambda: enc.run(ct))
side = torch.cuda.Stream()
with torch.cuda.stream(side):          # the context keeps per-stream state: create it (and size its workspace) before capturing
    for _ in range(2):
        enc.run(ct)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = enc.run(ct)
g.replay(); torch.cuda.synchronize()
same = bool(torch.equal(out, ref))
graph = timed(g.replay)
print("sign1024x1 one image: eager %.2f ms, one-graph replay %.2f ms, logits equal: %s, label %d -> class %d"
      % (eager, graph, same, labels[1], sk.classify(out.cpu().numpy())))
