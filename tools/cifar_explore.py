"""Exploratory: reference's unmodified CIFAR driver on the GPU backend, logits vs plaintext checker."""
import os, sys, time, tempfile, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import plain_model as pm, refdrivers as rd
import shutil
net_name = sys.argv[1] if len(sys.argv) > 1 else "binarynet_small"
tmp = tempfile.mkdtemp()
client = os.path.join(tmp, "client"); netdir = os.path.join(tmp, "nets", "cifar", net_name)
os.makedirs(client); os.makedirs(netdir)
shutil.copyfile(os.path.join(rd.GOLD, "cifar_%s_var_prep.dat" % net_name), os.path.join(netdir, "var_prep.dat"))
print(rd.run("client_gen_secure_keyset.out", client).stdout[-60:])
labels, pix = pm.load_cifar_images()
net = pm.CifarNet(net_name)
params, lwe_key = rd.read_secret_key(os.path.join(client, "secret.key"))
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    with open(os.path.join(client, "img.csv"), "w") as f:
        f.write(",".join(str(int(v)) for v in [labels[i], 32, 32, 3] + list(pix[i])) + ",\n")
    assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
    t0 = time.time(); r = rd.run("cifar_%s_enc.out" % net_name, netdir); dt = time.time() - t0
    print("driver rc", r.returncode, "wall %.1f s" % dt, r.stdout[-200:].replace("\n", " | "), r.stderr[-300:])
    ct = rd.read_ciphertexts(os.path.join(client, "network_output.ctxt"), 350, 10)
    phase = (ct[:, 350].astype(np.int64) - (ct[:, :350].astype(np.int64) * lwe_key).sum(axis=1)) & 0xFFFFFFFF
    dec = ((phase + (1 << 19)) >> 20) & 0xFFF; dec = np.where(dec > 2048, dec - 4096, dec)
    plain = pm.cifar_forward(net, pix[i])
    print("img", i, "label", labels[i], "enc", dec.tolist(), "argmax", int(np.argmax(dec)))
    print("          plain", plain.tolist(), "argmax", int(np.argmax(plain)))
    print(rd.run("client_decrypt_image.out", client, "CIFAR-10").stdout.strip())
