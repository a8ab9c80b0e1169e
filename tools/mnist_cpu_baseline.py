"""BASELINE configs[0]: encrypted nets/mnist/sign1024x1, one image, on the host CPU -- the plumbing
baseline that stands in for `make cpu-encrypt` (TFHE itself is not available): the oracle's layer chain
(tests/oracle_net.py) on its double-precision FFT product path, OpenMP over the gates of a layer.

  python tools/mnist_cpu_baseline.py [image_index]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol, oracle_net, plain_model as pm
from redsec_amd import client

i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sk = client.SecretKeySet("redsec_small_v2", seed=7)
net = pm.load_net("sign1024x1")
labels, pixels = pm.load_images()
ct = sk.encrypt_image(pixels[i], seed=5)


class K:
    pass
k = K(); k.p = ol.params("redsec_small_v2"); k.bk = sk.bk.ravel(); k.ksk = sk.ksk.ravel()
sys.path.insert(0, ROOT)
import bench
ol.lib().ro_set_threads(bench.host_cpu_share())   # the CPU share actually granted (cgroup quota), not the host's thread count
ctx = ol.Ctx(k)
ctx.set_fft(True)
t0 = time.perf_counter()
out = oracle_net.run(ctx, net, ct)
dt = time.perf_counter() - t0
logits = sk.decrypt_ints(out)
print("sign1024x1 on the CPU oracle (FFT path): %.2f s on %d OpenMP threads, 1220 bootstraps -> %.1f bootstraps/s"
      % (dt, ol.lib().ro_max_threads(), 1220 / dt))
print("label %d, encrypted argmax %d, plaintext argmax %d" % (labels[i], int(np.argmax(logits)), int(np.argmax(pm.forward(net, pixels[i])))))
