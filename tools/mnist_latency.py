"""Per-stage latency of encrypted sign1024x1 (one image, device-resident)."""
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
for _ in range(2): enc.run(ct)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); enc.run(ct); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("whole image ms:", [round(t * 1e3, 2) for t in ts])
for K in (8, 32):      # K images per call: every bootstrap stage one launch over all of them
    cts = torch.stack([torch.from_numpy(sk.encrypt_image(pixels[k % 100], seed=50 + k)).cuda() for k in range(K)])
    enc.run_many(cts); torch.cuda.synchronize()
    t0 = time.perf_counter(); many = enc.run_many(cts); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = all(torch.equal(many[k], enc.run(cts[k])) for k in (0, K - 1))
    print("%d images per call: %.2f ms = %.2f ms per image (equal to run() word for word: %s)" % (K, dt * 1e3, dt * 1e3 / K, same))
be.set_timing(True)
x = torch.randint(-2**31, 2**31 - 1, (196, be.W), dtype=torch.int64).to(torch.int32).cuda()
for B in (196, 600, 1024, 2048):
    x = torch.randint(-2**31, 2**31 - 1, (B, be.W), dtype=torch.int64).to(torch.int32).cuda()
    be.bootstrap(x, 1 << 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); be.bootstrap(x, 1 << 20); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print("B", B, "blind_rotate/keyswitch ms", [round(v, 3) for v in be.last_kernel_ms()], "wall", round(wall * 1e3, 3))
be.set_timing(False)
s, z, b = enc.fc[0]
bits = torch.randint(-2**31, 2**31 - 1, (196, be.W), dtype=torch.int64).to(torch.int32).cuda()
torch.cuda.synchronize(); t0 = time.perf_counter(); be.linear_fc(bits, s, z, bias_b=b); torch.cuda.synchronize(); print("fc 196->1024 ms", round((time.perf_counter() - t0) * 1e3, 3))
s, z, b = enc.final
bits = torch.randint(-2**31, 2**31 - 1, (1024, be.W), dtype=torch.int64).to(torch.int32).cuda()
torch.cuda.synchronize(); t0 = time.perf_counter(); be.linear_fc(bits, s, z, bias_b=b); torch.cuda.synchronize(); print("fc 1024->10 ms", round((time.perf_counter() - t0) * 1e3, 3))
