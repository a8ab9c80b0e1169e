import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, redsec_amd
from redsec_amd import client
sk = client.SecretKeySet("default128", seed=3)
be = redsec_amd.Backend(redsec_amd.params("default128"), 0); be.load_keys(sk.bk, sk.ksk); be.set_timing(True)
rng = np.random.default_rng(0)
for B in (600, 1024, 1536, 2047, 2048):
    a = torch.from_numpy(rng.integers(-2**31, 2**31, (B, be.W), dtype=np.int32)).cuda()
    b = torch.from_numpy(rng.integers(-2**31, 2**31, (B, be.W), dtype=np.int32)).cuda()
    be.gate("NAND", a, b); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = be.gate("NAND", a, b); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("RS_NO_WG4=%s" % os.environ.get("RS_NO_WG4", "0"), "B", B, be.last_launch()["form"], be.last_launch()["waves_per_block"], "br/ks ms", [round(v, 3) for v in be.last_kernel_ms()], "wall %.3f" % (dt * 1e3))
