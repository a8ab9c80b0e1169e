"""Host overhead of the C++ layer mirror on BASELINE configs[3]: the reference's unmodified nets/cifar/binarynet driver with
REDSEC_TRACE=1 (per layer: staging incl. freeing the caller's arrays, device stages, publication), eager and with
REDSEC_LAZY_HOST=1 (intermediate host arrays left unfilled), and with REDSEC_DEVICES=0,0 (two contexts on the one device).

  python tools/cifar_trace.py [binarynet|binarynet_small]
"""
import os, re, sys, time, tempfile, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import plain_model as pm, refdrivers as rd

net_name = sys.argv[1] if len(sys.argv) > 1 else "binarynet"
tmp = tempfile.mkdtemp()
client = os.path.join(tmp, "client"); netdir = os.path.join(tmp, "nets", "cifar", net_name)
os.makedirs(client); os.makedirs(netdir)
shutil.copyfile(os.path.join(rd.GOLD, "cifar_%s_var_prep.dat" % net_name), os.path.join(netdir, "var_prep.dat"))
assert rd.run("client_gen_secure_keyset.out", client).returncode == 0
labels, pix = pm.load_cifar_images()
with open(os.path.join(client, "img.csv"), "w") as f:
    f.write(",".join(str(int(v)) for v in [labels[1], 32, 32, 3] + list(pix[1])) + ",\n")
assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
exe = os.path.join(rd.REFNETS, "cifar_%s_enc.out" % net_name)
ref = None
for label, extra in (("eager", {}), ("eager again", {}), ("lazy host arrays", {"REDSEC_LAZY_HOST": "1"}),
                     ("two contexts on one device, lazy", {"REDSEC_LAZY_HOST": "1", "REDSEC_DEVICES": "0,0"})):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "redsec_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    env["REDSEC_TRACE"] = "1"
    env.update(extra)
    t0 = time.time()
    r = subprocess.run([exe], cwd=netdir, env=env, capture_output=True, text=True, timeout=900)
    wall = time.time() - t0
    lines = [l for l in r.stderr.splitlines() if "redsec trace" in l]
    tot = {"stage": 0.0, "device": 0.0, "publish": 0.0}
    for l in lines:
        for k in tot:
            m = re.search(k + r" ([0-9.]+) ms", l)
            if m:
                tot[k] += float(m.group(1))
    out = open(os.path.join(client, "network_output.ctxt"), "rb").read()
    same = ref is None or out == ref
    ref = ref or out
    print("%-36s rc %d wall %.2f s | layers: stage %.1f ms, device %.1f ms, publish %.1f ms | output equal to eager: %s"
          % (label, r.returncode, wall, tot["stage"], tot["device"], tot["publish"], same), flush=True)
    if label == "lazy host arrays":
        print("\n".join(lines))
print(rd.run("client_decrypt_image.out", client, "CIFAR-10").stdout.strip(), "(label %d)" % labels[1])
