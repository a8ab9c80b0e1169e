#!/usr/bin/env python3
"""debug: outputs of redsec_large (synthetic key) on 2 ciphertexts, saved per variant for comparison"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, redsec_amd
tag = sys.argv[1]; n_override = int(sys.argv[2]) if len(sys.argv) > 2 else None
p = redsec_amd.params("redsec_large")
be = redsec_amd.Backend(p, device=0)
be.load_synthetic_keys(77)
be.set_mode("split")
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.integers(-2**31, 2**31, (2, p.n + 1), dtype=np.int64).astype(np.int32)).cuda()
for rep in range(2):
    out = be.empty(2, p.n + 1)
    try:
        be.bootstrap(x, 1 << 29, out=out)
        torch.cuda.synchronize()
        cert = be.rounding_certificate()
    except Exception as e:
        cert = "ERR " + str(e)[:50]
    np.save("gpurun_out/r03_gen5/out_%s_%d.npy" % (tag, rep), out.cpu().numpy())
    print(tag, rep, cert, flush=True)
    try:
        be.certify(reset=True)
    except Exception:
        pass
