"""Kernel-level profile of BASELINE configs[3]: the reference's unmodified nets/cifar/binarynet driver
(693,248 bootstraps) on one MI355X. Prepares keys + one encrypted image with the reference's client
tools, then runs the driver binary itself under `rocprofv3 --kernel-trace --stats`.

  python tools/cifar_profile.py [binarynet|binarynet_small] [out_dir]
"""
import os, sys, time, tempfile, shutil, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import plain_model as pm, refdrivers as rd

net_name = sys.argv[1] if len(sys.argv) > 1 else "binarynet"
out_dir = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "cifar_profile"))
tmp = tempfile.mkdtemp()
client = os.path.join(tmp, "client"); netdir = os.path.join(tmp, "nets", "cifar", net_name)
os.makedirs(client); os.makedirs(netdir); os.makedirs(out_dir, exist_ok=True)
shutil.copyfile(os.path.join(rd.GOLD, "cifar_%s_var_prep.dat" % net_name), os.path.join(netdir, "var_prep.dat"))
assert rd.run("client_gen_secure_keyset.out", client).returncode == 0
labels, pix = pm.load_cifar_images()
with open(os.path.join(client, "img.csv"), "w") as f:
    f.write(",".join(str(int(v)) for v in [labels[1], 32, 32, 3] + list(pix[1])) + ",\n")
assert rd.run("client_encrypt_image.out", client, "img.csv").returncode == 0
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "redsec_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
exe = os.path.join(rd.REFNETS, "cifar_%s_enc.out" % net_name)
t0 = time.time()
r = subprocess.run([exe], cwd=netdir, env=env, capture_output=True, text=True, timeout=900)
print("plain run: rc", r.returncode, "wall %.2f s" % (time.time() - t0))
env_t = dict(env); env_t["REDSEC_TRACE"] = "1"
t0 = time.time()
r = subprocess.run([exe], cwd=netdir, env=env_t, capture_output=True, text=True, timeout=900)
print("traced run: rc", r.returncode, "wall %.2f s" % (time.time() - t0))
print("".join(l + "\n" for l in r.stderr.splitlines() if "redsec trace" in l), end="")
t0 = time.time()
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "--", exe],
                   cwd=netdir, env=env, capture_output=True, text=True, timeout=1200)
print("profiled run: rc", r.returncode, "wall %.2f s" % (time.time() - t0), r.stderr[-300:] if r.returncode else "")
for f in glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
print(rd.run("client_decrypt_image.out", client, "CIFAR-10").stdout.strip(), "(label %d)" % labels[1])
