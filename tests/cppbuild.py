"""Compiles the small C++ test programs under tests/cpp/ against the layer mirror (libredsec_layers.so), as a
REDsec translation unit would be: -I redsec_amd/host, the shim's <tfhe/tfhe.h>, the mirrored lib/*.h."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "tests")


def build(name):
    from redsec_amd import build as b
    b.build_layers()
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    exe = os.path.join(OUT, name + ".out")
    os.makedirs(OUT, exist_ok=True)
    deps = [src, b.LAYERS_LIB] + [os.path.join(ROOT, "redsec_amd", "host", "lib", h) for h in ("Layer.h", "BinLayer.h", "IntLayer.h", "BinFunc.h", "IntFunc.h")] + \
        [os.path.join(ROOT, "redsec_amd", "host", "tfhe", "tfhe.h")]
    if not b.is_stale(exe, deps):      # content fingerprints, not mtimes: a pushed snapshot never recompiles (redsec_amd/build.py)
        return exe
    cxx = shutil.which("g++")
    if cxx is None:
        return exe if os.path.exists(exe) else None
    lib = os.path.join(ROOT, "redsec_amd")
    subprocess.check_call([cxx, "-O1", "-w", "-std=c++17", "-I" + os.path.join(lib, "host"), src, "-L" + lib, "-lredsec_layers",
                           "-lredsec_hip", "-Wl,-rpath," + lib, "-o", exe])
    b.write_stamp(exe, deps)
    return exe


def run(exe, *args, cwd=None, env=None, timeout=900):
    e = dict(os.environ)
    e["LD_LIBRARY_PATH"] = os.path.join(ROOT, "redsec_amd") + ":" + e.get("LD_LIBRARY_PATH", "")
    e.update(env or {})
    return subprocess.run([exe] + list(args), cwd=cwd, env=e, capture_output=True, text=True, timeout=timeout)
