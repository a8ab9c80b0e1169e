"""Helpers to run the reference's own UNMODIFIED drivers (built by redsec_amd.build.build_reference_drivers
into build/refnets/) inside a scratch tree that reproduces the relative paths they hard-code
(nets/mnist/sign1024x1/net.cpp:53 "../../../client/eval.key", main.cpp:55,73)."""
import os
import shutil
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFNETS = os.path.join(ROOT, "build", "refnets")
GOLD = os.path.join(ROOT, "tests", "golden")
HEADER = struct.Struct("<I7i4d")   # redsec_amd/host/tfhe_shim.cpp ParamHeader


def available():
    return os.path.exists(os.path.join(REFNETS, "client_gen_secure_keyset.out"))


def make_tree(tmp, net="sign1024x1"):
    client = os.path.join(tmp, "client")
    netdir = os.path.join(tmp, "nets", "mnist", net)
    os.makedirs(client, exist_ok=True)
    os.makedirs(netdir, exist_ok=True)
    shutil.copyfile(os.path.join(GOLD, "mnist_%s_var_prep.dat" % net), os.path.join(netdir, "var_prep.dat"))
    return client, netdir


def run(exe, cwd, *args):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "redsec_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([os.path.join(REFNETS, exe)] + list(args), cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def write_image_csv(path, label, pixels):
    """client/image_converter.py output: label,H,W,C,pixels...  (every field comma-terminated:
    encrypt_image.cpp:37-79 only consumes tokens that are followed by a delimiter)."""
    with open(path, "w") as f:
        f.write(",".join(str(int(v)) for v in [label, 28, 28, 1] + list(pixels)) + ",\n")


def read_secret_key(path):
    """secret.key as the shim writes it: TFHE v1.1's layout (default) or the private RSS1 header + arrays
    (REDSEC_KEY_FORMAT=rs). The TFHE layout is parsed by redsec_amd/client.py's own reader, an independent
    restatement of the format in the other language."""
    with open(path, "rb") as f:
        if f.peek(1)[:1] == b"-":
            from redsec_amd import client
            k = client.read_tfhe_keyset(f, secret=True)
            return dict(n=k["n"], N=k["N"], k=k["k"], l=k["l"], Bgbit=k["Bgbit"], t=k["t"], basebit=k["basebit"]), k["lwe_key"]
        magic, n, N, k, l, bg, t, bb, *_ = HEADER.unpack(f.read(HEADER.size))
        assert magic == 0x31535352
        lwe = np.frombuffer(f.read(4 * n), np.int32)
    return dict(n=n, N=N, k=k, l=l, Bgbit=bg, t=t, basebit=bb), lwe


def read_ciphertexts(path, n, count):
    """TFHE v1.1 LweSample records: int32 type uid (42), int32 a[n], int32 b, double variance."""
    rec = 4 + 4 * n + 4 + 8
    raw = open(path, "rb").read()
    assert len(raw) == rec * count, (len(raw), rec, count)
    out = np.zeros((count, n + 1), np.int32)
    for i in range(count):
        assert np.frombuffer(raw[i * rec:i * rec + 4], np.int32)[0] == 42
        out[i] = np.frombuffer(raw[i * rec + 4:i * rec + 4 + 4 * (n + 1)], np.int32)
    return out
